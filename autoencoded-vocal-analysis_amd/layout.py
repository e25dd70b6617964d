"""Static description of the VAE network and of the flat parameter arena.

Mirrors the layer table of the reference (``ava/models/vae.py:125-168``) and
its ``named_parameters()`` order (which is the Adam parameter index order used
in checkpoints, ``ava/models/vae.py:119,439``).  Nothing here touches a device.
"""
from collections import OrderedDict, namedtuple

X_SHAPE = (128, 128)          # ava/models/vae.py:33
X_DIM = 128 * 128             # ava/models/vae.py:35

# (name, Cin, Cout, stride) ; 3x3, padding 1 ; ava/models/vae.py:128-134
ENC_CONVS = [
    ("conv1", 1, 8, 1), ("conv2", 8, 8, 2), ("conv3", 8, 16, 1), ("conv4", 16, 16, 2),
    ("conv5", 16, 24, 1), ("conv6", 24, 24, 2), ("conv7", 24, 32, 1),
]
# ConvTranspose2d 3x3 padding 1 (stride 2 => output_padding 1) ; ava/models/vae.py:155-161
DEC_CONVTS = [
    ("convt1", 32, 24, 1), ("convt2", 24, 24, 2), ("convt3", 24, 16, 1), ("convt4", 16, 16, 2),
    ("convt5", 16, 8, 1), ("convt6", 8, 8, 2), ("convt7", 8, 1, 1),
]
ENC_BN = [("bn%d" % (i + 1), c[1]) for i, c in enumerate(ENC_CONVS)]        # vae.py:135-141
DEC_BN = [("bn%d" % (i + 8), c[1]) for i, c in enumerate(DEC_CONVTS)]       # vae.py:162-168


def bottleneck_features(x_shape=X_SHAPE):
    """fc1.in = fc8.out: 32 channels x (H/8) x (W/8) -- the literal 8192 of ava/models/vae.py:142,153,224,262 at the
    reference's 128 x 128."""
    return 32 * (x_shape[0] // 8) * (x_shape[1] // 8)


def check_x_shape(x_shape):
    """Spectrogram sizes the kernels are validated for (csrc/model.hip: size_ok): H and W each 128 or 256."""
    h, w = int(x_shape[0]), int(x_shape[1])
    if w not in (128, 256) or h not in (128, 256):
        raise ValueError("unsupported spectrogram size %dx%d: height and width must each be 128 or 256" % (h, w))
    return (h, w)


def fc_layers(z_dim, x_shape=X_SHAPE):
    """(name, in_features, out_features) ; ava/models/vae.py:142-154"""
    f = bottleneck_features(x_shape)
    return [
        ("fc1", f, 1024), ("fc2", 1024, 256),
        ("fc31", 256, 64), ("fc32", 256, 64), ("fc33", 256, 64),
        ("fc41", 64, z_dim), ("fc42", 64, z_dim), ("fc43", 64, z_dim),
        ("fc5", z_dim, 64), ("fc6", 64, 256), ("fc7", 256, 1024), ("fc8", 1024, f),
    ]


# Order of the top-level keys in a checkpoint (``_get_layers``, ava/models/vae.py:171-186)
def checkpoint_layer_order():
    names = ["fc1", "fc2", "fc31", "fc32", "fc33", "fc41", "fc42", "fc43", "fc5", "fc6", "fc7", "fc8"]
    names += ["bn%d" % i for i in range(1, 15)]
    names += ["conv%d" % i for i in range(1, 8)]
    names += ["convt%d" % i for i in range(1, 8)]
    return names


ParamSpec = namedtuple("ParamSpec", "name layer kind shape numel index")


def param_specs(z_dim, x_shape=X_SHAPE):
    """Parameters in the reference's ``named_parameters()`` order (= module
    registration order in ``_build_network``): conv1..7, bn1..7, fc*, convt1..7,
    bn8..14.  ``index`` is the Adam param index in ``optimizer_state``."""
    specs = []

    def add(layer, kind, shape):
        n = 1
        for s in shape:
            n *= s
        specs.append(ParamSpec("%s.%s" % (layer, kind), layer, kind, tuple(shape), n, len(specs)))

    for name, cin, cout, _ in ENC_CONVS:
        add(name, "weight", (cout, cin, 3, 3))
        add(name, "bias", (cout,))
    for name, c in ENC_BN:
        add(name, "weight", (c,))
        add(name, "bias", (c,))
    for name, fin, fout in fc_layers(z_dim, x_shape):
        add(name, "weight", (fout, fin))
        add(name, "bias", (fout,))
    for name, cin, cout, _ in DEC_CONVTS:
        add(name, "weight", (cin, cout, 3, 3))      # ConvTranspose2d layout [Cin,Cout,kH,kW]
        add(name, "bias", (cout,))
    for name, c in DEC_BN:
        add(name, "weight", (c,))
        add(name, "bias", (c,))
    return specs


def arena_offsets(z_dim, align=64, x_shape=X_SHAPE):
    """Offsets (in floats) of every parameter inside the flat fp32 arena.

    Mirror of ``build_param_table`` in ``csrc/model.hip`` (the native library is the source
    of truth; ``tests/test_layout.py`` checks the two agree).  The arena keeps
    ``named_parameters()`` order except that the three 256->64 head layers are grouped as
    ``fc31.w fc32.w fc33.w | fc31.b fc32.b fc33.b`` so they form one [192,256] matrix and one
    [192] bias.  Every tensor starts on a 256-byte boundary (``align`` floats); the same
    offsets index the gradient, exp_avg and exp_avg_sq arenas.
    Returns (OrderedDict name -> offset, total_floats)."""
    specs = param_specs(z_dim, x_shape)
    order = list(range(32)) + [32, 34, 36, 33, 35, 37] + list(range(38, len(specs)))
    offs = {}
    cur = 0
    for i in order:
        s = specs[i]
        offs[s.name] = cur
        cur += (s.numel + align - 1) // align * align
    return OrderedDict((s.name, offs[s.name]) for s in specs), cur


def num_params(z_dim, x_shape=X_SHAPE):
    return sum(s.numel for s in param_specs(z_dim, x_shape))
