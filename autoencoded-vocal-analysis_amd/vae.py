"""MI355X-native drop-in for ``ava.models.vae`` (reference: ``ava/models/vae.py``).

Same class surface (``VAE(save_dir, lr, z_dim, model_precision, device_name)``,
``encode/decode/forward``, ``train_epoch/test_epoch/train_loop``,
``save_state/load_state``, ``visualize``, ``get_latent``; constants ``X_SHAPE``,
``X_DIM``) and the same checkpoint dict, but every arithmetic op of the training
step runs in hand-written HIP kernels (``csrc/``) reached through the C ABI of
``libava_hip.so``.  PyTorch is used for device memory, streams, serialisation and
``torch.distributed`` only.  There is no CPU compute path: on a machine without
the GPU library/GPU the model can be constructed, saved and loaded, but any
forward raises.

Memory model: all 80 parameters are views into ONE flat fp32 arena on the
device (offsets from ``ava_param_offset``); three more arenas of the same shape
hold the gradient and Adam's exp_avg / exp_avg_sq, so that Adam and the
data-parallel gradient all-reduce are single flat operations.  The 40 layer
attributes (``conv1`` ... ``bn14``) are stock ``torch.nn`` modules used purely
as parameter containers so that ``state_dict()`` keys/shapes are the
reference's.
"""
import ctypes
import os

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import dist as _dist
from .layout import X_SHAPE, X_DIM, param_specs, checkpoint_layer_order, bottleneck_features, check_x_shape
from .optim import FlatAdam
from .feed import DeviceFeeder

__all__ = ["VAE", "X_SHAPE", "X_DIM"]

X_DIM = int(np.prod(X_SHAPE))

_BN_CH = [1, 8, 8, 16, 16, 24, 24, 32, 24, 24, 16, 16, 8, 8]


class _ElboFn(torch.autograd.Function):
    """Makes ``loss = model(x); loss.backward()`` work: forward ran the HIP forward,
    backward runs the HIP backward into the gradient arena (vae.py:350-352).

    The activations a backward needs live in the model's ONE workspace (not in per-call autograd
    storage), so a backward is only valid while no later ``forward`` / ``encode`` / ``decode`` /
    ``get_latent`` / ``visualize`` has overwritten them, and only once; both are checked (the
    reference's autograd would handle e.g. ``(model(x1) + model(x2)).backward()``; here it raises)."""

    @staticmethod
    def forward(ctx, anchor, model, x):
        ctx.model = model
        ctx.x = x
        loss = model._forward_device(x, need_grad=True).clone()
        ctx.generation = model._generation
        ctx.done = False
        return loss

    @staticmethod
    def backward(ctx, grad_out):
        model = ctx.model
        if ctx.done:
            raise RuntimeError("backward was already run for this forward (the HIP path keeps no per-call graph)")
        if ctx.generation != model._generation:
            raise RuntimeError("the saved activations of this forward were overwritten by a later forward / encode / "
                               "decode / get_latent / visualize call on the same model; run backward right after "
                               "the forward it belongs to")
        ctx.done = True
        # grad_out is 1 when the loss is the root of the graph (the reference's use); any other value scales the two
        # roots of the hand-written backward on the device (no host sync, no pass over the 70 MB gradient arena)
        scale = grad_out.detach().to(device=model.device, dtype=torch.float32).reshape(1).contiguous()
        _lib.check(_lib.load().ava_set_backward_scale(model._handle, scale.data_ptr()), "ava_set_backward_scale")
        model._scale_keepalive = scale
        model._backward_device(ctx.x)
        model._attach_grads()
        return None, None, None


class VAE(nn.Module):
    """Variational Autoencoder for single-channel spectrograms, 128x128 unless ``x_shape`` says otherwise
    (reference: ``ava/models/vae.py:40-122``)."""

    def __init__(self, save_dir='', lr=1e-3, z_dim=32, model_precision=10.0, device_name="auto", *, x_shape=X_SHAPE,
                 act_dtype="float32"):
        """Reference signature (vae.py:80-81) plus one keyword-only extension: ``x_shape`` -- the reference fixes the
        spectrogram size in the module constant ``X_SHAPE = (128, 128)`` and the literal 8192 (vae.py:33,142,153);
        here the layers scale with ``(H, W)`` (width 128 or 256, height a multiple of 128; BASELINE config 5 is
        256 x 256) and ``fc1.in = fc8.out = 32 * H/8 * W/8``; ``act_dtype`` -- ``"float32"`` (the reference) or
        ``"bfloat16"`` (BASELINE config 5: "bf16 conv + fp32 ELBO"): the activations between the 14 conv layers are stored
        as bf16 (``ava_model_create_ex``) and the twelve convolutions with >= 8 channels on both sides compute in bf16
        arithmetic -- weights and BatchNorm outputs rounded to bfloat16, products exact, fp32 accumulation, the backward
        the exact derivative of that function with fp32 gradients; conv1 / convt7, BatchNorm statistics, the fully
        connected layers, the ELBO and Adam stay fp32 (``oracle/vae_oracle.py: BF16_MATH_LAYERS``)."""
        super(VAE, self).__init__()
        if act_dtype not in ("float32", "bfloat16"):
            raise ValueError("act_dtype must be 'float32' or 'bfloat16'")
        self.act_dtype = act_dtype
        self.x_shape = check_x_shape(x_shape)
        self.x_dim = self.x_shape[0] * self.x_shape[1]
        self.save_dir = save_dir
        self.lr = lr
        self.z_dim = z_dim
        self.model_precision = model_precision
        assert device_name != "cuda" or torch.cuda.is_available()          # vae.py:112
        if device_name == "auto":
            device_name = "cuda" if torch.cuda.is_available() else "cpu"
        self.device = torch.device(device_name)
        if self.save_dir != '' and not os.path.exists(self.save_dir):
            os.makedirs(self.save_dir)
        self._build_network()
        self._flatten_parameters()
        self.optimizer = FlatAdam(self, lr=self.lr)
        self.epoch = 0
        self.loss = {'train': {}, 'test': {}}
        # noise for rsample: None -> device counter RNG; or callable(B, z_dim) -> (eps_W [B,1], eps_D [B,z])
        self.noise_source = None
        self.prefetch = True          # prefetch the next batch to the device on a copy stream in the epoch loops (feed.py)
        self._rng_seed = int(torch.randint(0, 2 ** 31 - 1, (1,)).item())
        self._rng_offset = 0
        self._handle = None
        self._max_batch = 0
        self._workspace = None
        self._last_x = None
        self._generation = 0          # bumped by every call that overwrites the workspace (see _ElboFn)
        self._grad_state = "none"     # "none" (zero_grad) | "filled": what the next backward / step must do (optim.py)

    # ------------------------------------------------------------------ network definition
    def _build_network(self):
        """Same modules, same registration order as the reference (vae.py:125-168); they only
        hold parameters/buffers -- their forward() is never called."""
        self.conv1 = nn.Conv2d(1, 8, 3, 1, padding=1)
        self.conv2 = nn.Conv2d(8, 8, 3, 2, padding=1)
        self.conv3 = nn.Conv2d(8, 16, 3, 1, padding=1)
        self.conv4 = nn.Conv2d(16, 16, 3, 2, padding=1)
        self.conv5 = nn.Conv2d(16, 24, 3, 1, padding=1)
        self.conv6 = nn.Conv2d(24, 24, 3, 2, padding=1)
        self.conv7 = nn.Conv2d(24, 32, 3, 1, padding=1)
        self.bn1 = nn.BatchNorm2d(1)
        self.bn2 = nn.BatchNorm2d(8)
        self.bn3 = nn.BatchNorm2d(8)
        self.bn4 = nn.BatchNorm2d(16)
        self.bn5 = nn.BatchNorm2d(16)
        self.bn6 = nn.BatchNorm2d(24)
        self.bn7 = nn.BatchNorm2d(24)
        self.fc1 = nn.Linear(bottleneck_features(self.x_shape), 1024)
        self.fc2 = nn.Linear(1024, 256)
        self.fc31 = nn.Linear(256, 64)
        self.fc32 = nn.Linear(256, 64)
        self.fc33 = nn.Linear(256, 64)
        self.fc41 = nn.Linear(64, self.z_dim)
        self.fc42 = nn.Linear(64, self.z_dim)
        self.fc43 = nn.Linear(64, self.z_dim)
        self.fc5 = nn.Linear(self.z_dim, 64)
        self.fc6 = nn.Linear(64, 256)
        self.fc7 = nn.Linear(256, 1024)
        self.fc8 = nn.Linear(1024, bottleneck_features(self.x_shape))
        self.convt1 = nn.ConvTranspose2d(32, 24, 3, 1, padding=1)
        self.convt2 = nn.ConvTranspose2d(24, 24, 3, 2, padding=1, output_padding=1)
        self.convt3 = nn.ConvTranspose2d(24, 16, 3, 1, padding=1)
        self.convt4 = nn.ConvTranspose2d(16, 16, 3, 2, padding=1, output_padding=1)
        self.convt5 = nn.ConvTranspose2d(16, 8, 3, 1, padding=1)
        self.convt6 = nn.ConvTranspose2d(8, 8, 3, 2, padding=1, output_padding=1)
        self.convt7 = nn.ConvTranspose2d(8, 1, 3, 1, padding=1)
        self.bn8 = nn.BatchNorm2d(32)
        self.bn9 = nn.BatchNorm2d(24)
        self.bn10 = nn.BatchNorm2d(24)
        self.bn11 = nn.BatchNorm2d(16)
        self.bn12 = nn.BatchNorm2d(16)
        self.bn13 = nn.BatchNorm2d(8)
        self.bn14 = nn.BatchNorm2d(8)

    def _get_layers(self):
        """name -> layer, in the reference's checkpoint key order (vae.py:171-186)."""
        return {name: getattr(self, name) for name in checkpoint_layer_order()}

    # ------------------------------------------------------------------ flat arenas
    def _arena_layout(self):
        """(offsets by parameter name, total floats).  The native library is the source of truth;
        ``layout.arena_offsets`` mirrors it (tests/test_layout.py) for machines without it."""
        specs = param_specs(self.z_dim, self.x_shape)
        H, W = self.x_shape
        try:
            lib = _lib.load()
            offs = {s.name: int(lib.ava_param_offset_hw(self.z_dim, H, W, s.index, None)) for s in specs}
            total = int(lib.ava_arena_floats_hw(self.z_dim, H, W))
        except (_lib.AvaHipError, OSError):
            from .layout import arena_offsets
            offs, total = arena_offsets(self.z_dim, x_shape=self.x_shape)
        return specs, offs, total

    def _flatten_parameters(self):
        """Move every parameter / BatchNorm buffer into the flat device arenas and re-point the
        container modules at views of them."""
        specs, offs, total = self._arena_layout()
        dev = self.device
        old = dict(self.named_parameters())
        self._params = torch.zeros(total, dtype=torch.float32, device=dev)
        self._grads = torch.zeros(total, dtype=torch.float32, device=dev)
        self._exp_avg = torch.zeros(total, dtype=torch.float32, device=dev)
        self._exp_avg_sq = torch.zeros(total, dtype=torch.float32, device=dev)
        self._arena_views = {}
        for s in specs:
            o = offs[s.name]
            view = self._params[o:o + s.numel].view(s.shape)
            view.copy_(old[s.name].detach().to(dev))
            p = nn.Parameter(view)
            setattr(getattr(self, s.layer), s.kind, p)
            self._arena_views[s.name] = (o, s.numel, s.shape)
        self._bn_running = torch.zeros(2, 14, 32, dtype=torch.float32, device=dev)
        self._bn_running[1].fill_(1.0)
        self._bn_batches = torch.zeros(14, dtype=torch.int64, device=dev)
        for l, c in enumerate(_BN_CH):
            bn = getattr(self, "bn%d" % (l + 1))
            self._bn_running[0, l, :c].copy_(bn.running_mean.detach().to(dev))
            self._bn_running[1, l, :c].copy_(bn.running_var.detach().to(dev))
            self._bn_batches[l].copy_(bn.num_batches_tracked.detach().to(dev))
            bn.running_mean = self._bn_running[0, l, :c]
            bn.running_var = self._bn_running[1, l, :c]
            bn.num_batches_tracked = self._bn_batches[l]
        self._attach_grads()
        self._anchor = torch.zeros((), device=dev, requires_grad=True)
        self._loss_buf = torch.zeros(4, dtype=torch.float32, device=dev)
        self._loss_acc = torch.zeros((), dtype=torch.float64, device=dev)
        self._status = torch.zeros(2, dtype=torch.int32, device=dev)   # [0] d <= 0 flag, [1] Adam steps skipped because of it

    def _grad_view(self, name):
        o, n, shape = self._arena_views[name]
        return self._grads[o:o + n].view(shape)

    def _attach_grads(self):
        for name, p in self.named_parameters():
            p.grad = self._grad_view(name)

    def _linked(self):
        p = self.conv1.weight
        return p.device == self._params.device and p.data_ptr() == self._params.data_ptr() + 4 * self._arena_views["conv1.weight"][0]

    def _apply(self, fn, *args, **kwargs):
        # .to()/.cuda()/.float() re-create parameters; fold them back into fresh arenas afterwards
        out = super(VAE, self)._apply(fn, *args, **kwargs)
        if hasattr(self, "_params") and not self._linked():
            self.device = self.conv1.weight.device
            m, v = self._exp_avg, self._exp_avg_sq
            self._destroy_handle()
            self._flatten_parameters()
            self._exp_avg.copy_(m.to(self.device))
            self._exp_avg_sq.copy_(v.to(self.device))
            if hasattr(self, "optimizer"):
                self.optimizer.rebind(self)
        return out

    # ------------------------------------------------------------------ native model handle
    def _destroy_handle(self):
        if getattr(self, "_handle", None):
            _lib.load().ava_model_destroy(self._handle)
        self._handle = None
        self._max_batch = 0
        self._workspace = None

    def __del__(self):
        try:
            self._destroy_handle()
        except Exception:
            pass

    def _ensure(self, batch):
        """(Re)create the native model with a workspace large enough for ``batch`` samples."""
        if self.device.type != "cuda":
            raise _lib.AvaHipError("the VAE hot path only runs on an MI355X (device %s): there is no CPU "
                                   "fallback in this package" % self.device)
        if self._handle is not None and batch <= self._max_batch:
            return
        lib = _lib.load()
        self._destroy_handle()
        cap = max(batch, 8)
        H, W = self.x_shape
        nbytes = lib.ava_workspace_bytes_hw(self.z_dim, H, W, cap)
        self._workspace = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        h = ctypes.c_void_p()
        rc = lib.ava_model_create_ex(ctypes.byref(h), self.z_dim, H, W, 1 if self.act_dtype == "bfloat16" else 0, cap,
                                     float(self.model_precision),
                                  self._params.data_ptr(), self._grads.data_ptr(), self._exp_avg.data_ptr(),
                                  self._exp_avg_sq.data_ptr(), self._bn_running.data_ptr(),
                                  self._bn_batches.data_ptr(), self._workspace.data_ptr(), nbytes)
        _lib.check(rc, "ava_model_create")
        self._handle = h
        self._max_batch = cap
        self._eps = torch.empty(cap * (self.z_dim + 1), dtype=torch.float32, device=self.device)

    def _feed(self, loader):
        """Batches of ``loader`` on the device, one batch ahead on a copy stream (``feed.DeviceFeeder``), unless
        ``self.prefetch`` is False, in which case every batch is moved synchronously like vae.py:349 does."""
        if getattr(loader, "device_resident", False):
            return loader                 # batches are produced on the device (spec.DeviceWindowLoader): nothing to copy
        if getattr(self, "prefetch", True) and self.device.type == "cuda":
            return DeviceFeeder(loader, self.device)
        return loader

    def _prep_x(self, x):
        x = x.to(device=self.device, dtype=torch.float32)
        assert x.dim() == 3 and tuple(x.shape[1:]) == self.x_shape, "expected [batch,%d,%d] spectrograms" % self.x_shape
        return x.contiguous()

    def _noise(self, B):
        """eps_W [B] and eps_D [B,z] on the device (reference draw order: eps_W first)."""
        if self.noise_source is not None:
            ew, ed = self.noise_source(B, self.z_dim)
            ew = torch.as_tensor(ew, dtype=torch.float32).reshape(B).to(self.device).contiguous()
            ed = torch.as_tensor(ed, dtype=torch.float32).reshape(B, self.z_dim).to(self.device).contiguous()
            return ew, ed
        n = B * (self.z_dim + 1)
        _lib.check(_lib.load().ava_fill_normal(self._eps.data_ptr(), n, self._rng_seed, self._rng_offset,
                                               _lib.stream()), "ava_fill_normal")
        self._rng_offset += n
        return self._eps[:B], self._eps[B:n].view(B, self.z_dim)

    # ------------------------------------------------------------------ device-side step pieces
    def _forward_device(self, x, need_grad, accumulate=False):
        """Runs ava_forward; returns the 0-dim loss view (device).  BatchNorm mode follows
        ``self.training`` exactly like nn.BatchNorm2d does in the reference.  ``accumulate`` adds the
        loss to ``self._loss_acc`` on the device (the epoch loops' running sum)."""
        B = x.shape[0]
        self._ensure(B)
        self._generation += 1
        self._finish_comm()                               # (a deferred all-reduce nobody consumed: never leave one in flight)
        _dist.apply_cu_reserve(_lib.load(), self._handle, False)  # no collective is in flight during a forward: grids for the whole chip
        if self.noise_source is None:
            # device counter RNG: the noise is drawn inside the forward's first launch (same stream as ava_fill_normal)
            n = B * (self.z_dim + 1)
            rc = _lib.load().ava_forward_noise(self._handle, x.data_ptr(), B, self._eps.data_ptr(), self._rng_seed,
                                               self._rng_offset, 1 if self.training else 0, self._loss_buf.data_ptr(),
                                               self._loss_acc.data_ptr() if accumulate else None,
                                               self._status.data_ptr(), _lib.stream())
            self._rng_offset += n
            ew, ed = self._eps[:B], self._eps[B:n].view(B, self.z_dim)
        else:
            ew, ed = self._noise(B)
            rc = _lib.load().ava_forward(self._handle, x.data_ptr(), B, ew.data_ptr(), ed.data_ptr(),
                                         1 if self.training else 0, self._loss_buf.data_ptr(),
                                         self._loss_acc.data_ptr() if accumulate else None,
                                         self._status.data_ptr(), _lib.stream())
        _lib.check(rc, "ava_forward")
        self._last_x = x
        self._last_noise = (ew, ed)
        return self._loss_buf[0]

    def _backward_device(self, x, defer_comm=False):
        """ava_backward into the gradient arena with torch's accumulation rule: after ``zero_grad()`` the arena is
        overwritten; a second backward without ``zero_grad()`` in between ADDS to it (gradient accumulation over
        micro-batches), at the price of one extra pass over the arena for that call only.

        ``defer_comm`` (data parallel only; the epoch loop and ``bench.py`` set it): the gradient buckets' all-reduces are
        left in flight and ``optimizer.step()`` consumes them bucket by bucket -- bucket 0's parameters are updated while
        bucket 1 is still on the wire.  ``p.grad`` must not be read in between (it is complete only after the step); every
        other caller gets the gradients complete on return."""
        lib = _lib.load()
        held = self._grads.clone() if self._grad_state == "filled" else None
        self._backward_kernels(lib, x, defer_comm and held is None)
        if held is not None:
            self._grads.add_(held)
        self._grad_state = "filled"

    def _backward_kernels(self, lib, x, defer_comm=False):
        if not _dist.active():
            _lib.check(lib.ava_backward(self._handle, x.data_ptr(), x.shape[0], _lib.stream()), "ava_backward")
            return
        # data parallel: backward runs in parts; the gradient bucket a part completes (fc8 + decoder, fc1's weight, the
        # other fully connected layers, the encoder) is all-reduced asynchronously while the next part is still running.
        # The "d not positive" word goes first: MAX over ranks, so that the Adam kernel's device-side guard and the epoch
        # loop's poll see the SAME word on every rank (one rank's NaN gradients reach every rank through the all-reduce).
        # CU reserve (dist.cu_reserve, default 0): part 0 runs with no collective in flight -- the status word goes out
        # behind it, with the first bucket -- so only the later parts launch on grids that leave room for a collective.
        self._finish_comm()
        pending = []
        for part, (off, cnt) in enumerate(self._buckets()):
            _dist.apply_cu_reserve(lib, self._handle, part > 0)
            _lib.check(lib.ava_backward_part(self._handle, x.data_ptr(), x.shape[0], part, _lib.stream()),
                       "ava_backward_part")
            if part == 0:
                pending.append((None, _dist.allreduce_max_async(self._status)))
            if self._sharded_adam():
                pending.append(((off, cnt), _dist.reduce_scatter_bucket_async(self._grads, off, cnt)))
            else:
                pending.append(((off, cnt), _dist.allreduce_gradients_async(self._grads[off:off + cnt])))
        _dist.apply_cu_reserve(lib, self._handle, False)
        self._pending_comm = pending
        if not defer_comm:
            self._finish_comm()

    def _finish_comm(self):
        """Make the compute stream wait for every collective ``_backward_kernels`` left in flight (gradients complete
        afterwards).  bench.py sets ``_comm_events`` to a list: the time the compute stream then spends blocked on the
        collectives (end of the last backward kernel -> all buckets reduced) is the exposed communication of the step."""
        pending = getattr(self, "_pending_comm", None)
        if not pending:
            return
        self._pending_comm = None
        evs = getattr(self, "_comm_events", None)
        if evs is not None:
            e0 = torch.cuda.Event(enable_timing=True)
            e0.record()
        _dist.wait_all([h for _, h in pending])
        if evs is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            evs.append((e0, e1))

    def _take_pending_comm(self):
        """The collectives ``_backward_kernels(defer_comm=True)`` left in flight, as [(bucket or None, handle)] in issue
        order, handed to the optimizer (FlatAdam.step waits for and updates one bucket at a time)."""
        pending = getattr(self, "_pending_comm", None)
        self._pending_comm = None
        return pending or []

    def _buckets(self):
        """(offset, count) of the gradient buckets, in the order the backward parts complete them"""
        out = getattr(self, "_bucket_cache", None)
        if out is not None and out[0] is self._handle:
            return out[1]
        lib = _lib.load()
        out = []
        for part in range(lib.ava_backward_num_parts()):
            off, cnt = ctypes.c_int64(), ctypes.c_int64()
            _lib.check(lib.ava_grad_bucket(self._handle, part, ctypes.byref(off), ctypes.byref(cnt)), "ava_grad_bucket")
            out.append((off.value, cnt.value))
        self._bucket_cache = (self._handle, out)
        return out

    def _sharded_adam(self):
        """Data parallel with the optimizer sharded over the ranks (dist.sharded_adam): only when every bucket splits."""
        if not _dist.sharded_adam() or self._handle is None:
            return False
        ok = getattr(self, "_sharded_ok", None)
        if ok is None:
            ok = self._sharded_ok = all(_dist.shard_of(o, c) is not None for o, c in self._buckets())
        return ok

    def gather_adam_state(self):
        """Sharded optimizer: make exp_avg / exp_avg_sq complete on every rank (collective; before a checkpoint)."""
        if not self._sharded_adam():
            return
        hs = []
        for o, c in self._buckets():
            hs.append(_dist.all_gather_bucket_async(self._exp_avg, o, c))
            hs.append(_dist.all_gather_bucket_async(self._exp_avg_sq, o, c))
        _dist.wait_all(hs)
        self._adam_state_stale = False

    def _check_status(self):
        """Blocking check of the device status word (d not positive in some forward so far).  Under data parallelism
        the word is MAX-ed over the ranks first, so that every rank raises together."""
        if _dist.active():
            _dist.allreduce_max_(self._status)
        if int(self._status[0].item()) != 0:
            self._raise_invalid_posterior()

    def _raise_invalid_posterior(self):
        # Adam launches issued between the offending forward and this point were skipped on the device (the kernel
        # counts them in word 1): take them back out of the host-side step count, so that ``state['step']`` and the
        # bias correction equal the number of updates really applied (the reference stops inside the offending forward,
        # vae.py:312, and never reaches ``optimizer.step()``).  What is NOT rolled back: the forwards that ran in the
        # meantime (at most the polling lag, two steps) have moved the BatchNorm running statistics.
        skipped = int(self._status[1].item())
        if self._handle is not None and _dist.active():
            skipped //= len(self._buckets())         # data parallel: one guarded launch per bucket (slice) and step
        opt = getattr(self, "optimizer", None)
        if opt is not None and skipped > 0:
            opt._step_count_flat = max(0, opt._step_count_flat - skipped)
        self._status.zero_()
        self._status_polls = []
        # LowRankMultivariateNormal's argument validation (vae.py:312) raises the same type
        raise ValueError("Expected parameter cov_diag to be positive (d = exp(.) under/overflowed)")

    _POLL_LAG = 2       # steps between a status copy and the step that looks at it

    def _poll_status(self):
        """Non-blocking form for the epoch loops.  The reference raises inside the offending ``forward``
        (vae.py:312); the loops here never wait for the device, so after every step the status word is copied to
        page-locked host memory asynchronously and looked at ``_POLL_LAG`` steps later (by then the copy has long
        landed: the wait is free), at the latest at the end of the epoch.  The lag is a fixed number of steps rather
        than "as soon as the copy has landed" so that under data parallelism -- where the word every rank copies is
        the MAX over ranks (``_backward_kernels``) -- all ranks raise at the SAME step and none is left waiting in a
        collective.  Parameters stay those of before the offending step in the meantime: ``ava_adam_step`` skips
        the update on the device while the word is set."""
        polls = getattr(self, "_status_polls", None)
        if polls is None:
            polls = self._status_polls = []
        if len(polls) >= self._POLL_LAG:
            host, ev = polls.pop(0)
            ev.synchronize()
            if int(host[0]) != 0:
                self._raise_invalid_posterior()
        ring = getattr(self, "_status_hosts", None)
        if ring is None:
            ring = self._status_hosts = [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(self._POLL_LAG + 1)]
            self._status_slot = 0
        host = ring[self._status_slot]
        self._status_slot = (self._status_slot + 1) % len(ring)
        host.copy_(self._status, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        polls.append((host, ev))

    # intermediates that are ACTIVATIONS between the convolutions: bfloat16 when act_dtype says so
    _ACT_BUFFERS = frozenset(["y%d" % i for i in range(1, 7)] + ["d%d" % i for i in range(1, 7)] + ["f8t"])

    # intermediates the step never stores (their consumers recompute them): written on demand by ava_debug_materialize
    _RECOMPUTED = frozenset(["y1"])

    def _workspace_tensor(self, name, shape):
        if name in self._RECOMPUTED:
            x = self._last_x
            _lib.check(_lib.load().ava_debug_materialize(self._handle, x.data_ptr(), x.shape[0], _lib.stream()),
                       "ava_debug_materialize")
        n = ctypes.c_int64()
        p = _lib.load().ava_debug_buffer(self._handle, name.encode(), ctypes.byref(n))
        if not p:
            raise KeyError(name)
        off = p - self._workspace.data_ptr()
        numel = int(np.prod(shape))
        if name in self._ACT_BUFFERS and self.act_dtype == "bfloat16":
            return self._workspace[off:off + 2 * numel].view(torch.bfloat16).view(shape)
        return self._workspace[off:off + 4 * numel].view(torch.float32).view(shape)

    # ------------------------------------------------------------------ reference API
    def encode(self, x):
        """q(z|x): returns ``mu [B,z], u [B,z,1], d [B,z]`` (vae.py:189-233)."""
        x = self._prep_x(x)
        B = x.shape[0]
        self._ensure(B)
        self._generation += 1
        mu = torch.empty(B, self.z_dim, device=self.device)
        u = torch.empty(B, self.z_dim, device=self.device)
        d = torch.empty(B, self.z_dim, device=self.device)
        rc = _lib.load().ava_encode(self._handle, x.data_ptr(), B, 1 if self.training else 0, mu.data_ptr(),
                                    u.data_ptr(), d.data_ptr(), _lib.stream())
        _lib.check(rc, "ava_encode")
        return mu, u.unsqueeze(-1), d

    def decode(self, z):
        """p(x|z) mean: ``[B,z] -> [B,16384]`` (vae.py:236-270)."""
        z = z.to(device=self.device, dtype=torch.float32).contiguous()
        B = z.shape[0]
        self._ensure(B)
        self._generation += 1
        out = torch.empty(B, self.x_dim, device=self.device)
        rc = _lib.load().ava_decode(self._handle, z.data_ptr(), B, 1 if self.training else 0, out.data_ptr(),
                                    _lib.stream())
        _lib.check(rc, "ava_decode")
        return out

    def forward(self, x, return_latent_rec=False):
        """-ELBO summed over the batch (vae.py:273-327).  The returned tensor supports
        ``.backward()`` (fills ``param.grad`` from the HIP backward) and ``.item()``."""
        x = self._prep_x(x)
        if torch.is_grad_enabled():
            loss = _ElboFn.apply(self._anchor, self, x)
        else:
            loss = self._forward_device(x, need_grad=False).clone()
        self._check_status()
        if return_latent_rec:
            B = x.shape[0]
            z = self._workspace_tensor("z", (B, self.z_dim)).detach().cpu().numpy()
            rec = self._workspace_tensor("xrec", (B, self.x_shape[0], self.x_shape[1])).detach().cpu().numpy()
            return loss, z, rec
        return loss

    def train_epoch(self, train_loader):
        """One epoch of Adam steps; returns the mean -ELBO per sample (vae.py:330-358).

        Same per-batch sequence as the reference (zero_grad, H2D, forward, backward, step) with
        the loss accumulated on the device and read back once per epoch instead of ``.item()``
        every step."""
        self.train()
        self._loss_acc.zero_()
        batch_idx = -1
        self._check_equal_shards(train_loader, "train_epoch")
        for batch_idx, data in enumerate(self._feed(train_loader)):
            self.optimizer.zero_grad()
            data = self._prep_x(data)
            self._forward_device(data, need_grad=True, accumulate=True)
            self._backward_device(data, defer_comm=True)     # data parallel: the step below consumes the buckets one by one
            self.optimizer.step()
            self._poll_status()
        self._check_status()
        train_loss = _dist.global_loss(self._loss_acc, self.z_dim, self.model_precision, batch_idx + 1, self.x_dim)
        train_loss /= self._global_len(train_loader)
        if _dist.rank() == 0:
            print('Epoch: {} Average loss: {:.4f}'.format(self.epoch, train_loss))
        self.epoch += 1
        return train_loss

    def _global_len(self, loader):
        """len(loader.dataset) over all ranks (single process: the local length, as the reference divides by,
        vae.py:355,383).  A collective under data parallelism, so it is computed once per dataset and remembered."""
        n = len(loader.dataset)
        if not _dist.active():
            return n
        cache = self.__dict__.setdefault("_global_len_cache", {})
        key = (id(loader.dataset), n)
        if key not in cache:
            cache[key] = _dist.global_dataset_len(n)
        return cache[key]

    def _check_equal_shards(self, loader, what):
        """Data parallel: every rank must run the same number of steps (dist.check_equal_batches: one 16-byte MAX
        all-reduce at the start of every epoch, entered by every rank)."""
        if not _dist.active():
            return
        try:
            nb = len(loader)
        except TypeError:
            nb = -1                                     # an iterable without a length: still enter the collective
        _dist.check_equal_batches(nb, what)

    def test_epoch(self, test_loader):
        """Mean -ELBO per sample with BatchNorm on running statistics (vae.py:361-385)."""
        self.eval()
        self._loss_acc.zero_()
        i = -1
        self._check_equal_shards(test_loader, "test_epoch")
        with torch.no_grad():
            for i, data in enumerate(self._feed(test_loader)):
                data = self._prep_x(data)
                self._forward_device(data, need_grad=False, accumulate=True)
        self._check_status()
        # data parallel: every rank iterates its own shard; the value returned is that of the GLOBAL test set on every rank,
        # like train_epoch's (eval mode has no per-rank BatchNorm statistics, so it equals the single-process value on the
        # concatenated data up to summation order).  Single process: the reference's expression unchanged.
        test_loss = _dist.global_loss(self._loss_acc, self.z_dim, self.model_precision, i + 1, self.x_dim)
        test_loss /= self._global_len(test_loader)
        if _dist.rank() == 0:
            print('Test loss: {:.4f}'.format(test_loss))
        return test_loss

    def train_loop(self, loaders, epochs=100, test_freq=2, save_freq=10, vis_freq=1):
        """Epoch scheduler (vae.py:388-430)."""
        if _dist.rank() == 0:               # data parallel: one banner, one PDF (rank 0's shard); single process: as the reference
            print("=" * 40)
            print("Training: epochs", self.epoch, "to", self.epoch + epochs - 1)
            print("Training set:", self._global_len(loaders['train']))
            print("Test set:", self._global_len(loaders['test']))
            print("=" * 40)
        elif _dist.active():                # (the two dataset lengths are collectives: every rank enters them, once per loader)
            self._global_len(loaders['train'])
            self._global_len(loaders['test'])
        for epoch in range(self.epoch, self.epoch + epochs):
            loss = self.train_epoch(loaders['train'])
            self.loss['train'][epoch] = loss
            if (test_freq is not None) and (epoch % test_freq == 0):
                loss = self.test_epoch(loaders['test'])
                self.loss['test'][epoch] = loss
            if (save_freq is not None) and (epoch % save_freq == 0) and (epoch > 0):
                filename = "checkpoint_" + str(epoch).zfill(3) + '.tar'
                self.gather_adam_state()                # sharded optimizer: collective, every rank
                if _dist.rank() == 0:
                    self.save_state(filename)
            if (vis_freq is not None) and (epoch % vis_freq == 0):
                self.visualize(loaders['test'])         # data parallel: every rank runs the forward, rank 0 writes the PDF

    def save_state(self, filename):
        """Checkpoint with the reference's dict layout (vae.py:433-446)."""
        layers = self._get_layers()
        state = {}
        for layer_name in layers:
            state[layer_name] = {k: v.detach().clone() for k, v in layers[layer_name].state_dict().items()}
        state['optimizer_state'] = self.optimizer.state_dict()
        state['loss'] = self.loss
        state['z_dim'] = self.z_dim
        state['epoch'] = self.epoch
        state['lr'] = self.lr
        state['save_dir'] = self.save_dir
        filename = os.path.join(self.save_dir, filename)
        torch.save(state, filename)

    def load_state(self, filename):
        """Load a checkpoint written by this class or by the reference (vae.py:449-472);
        ``lr``, ``save_dir`` and ``z_dim`` are not restored, like the reference."""
        checkpoint = torch.load(filename, map_location=self.device)
        assert checkpoint['z_dim'] == self.z_dim
        layers = self._get_layers()
        for layer_name in layers:
            layers[layer_name].load_state_dict(checkpoint[layer_name])
        self.optimizer.load_state_dict(checkpoint['optimizer_state'])
        self.loss = checkpoint['loss']
        self.epoch = checkpoint['epoch']

    def visualize(self, loader, num_specs=5, gap=(2, 6), save_filename='reconstruction.pdf'):
        """Plot random spectrograms and their reconstructions (vae.py:475-516)."""
        from .plotting import grid_plot
        assert num_specs <= len(loader.dataset) and num_specs >= 1
        indices = np.random.choice(np.arange(len(loader.dataset)), size=num_specs, replace=False)
        specs = torch.stack(loader.dataset[indices]).to(self.device)
        with torch.no_grad():
            _, _, rec_specs = self.forward(specs, return_latent_rec=True)
        specs = specs.detach().cpu().numpy()
        all_specs = np.stack([specs, rec_specs])
        save_filename = os.path.join(self.save_dir, save_filename)
        if _dist.rank() == 0:       # data parallel: forward() above holds a collective (the status word's MAX), so every rank
            grid_plot(all_specs, gap=gap, filename=save_filename)     # calls visualize; only rank 0 writes the one PDF
        return specs, rec_specs

    def get_latent(self, loader, bn_mode=None):
        """Latent means of everything in ``loader`` as float64 ``[N,z]`` (vae.py:519-547).

        ``bn_mode=None`` is the reference's behaviour: the module's current mode is used as is -- straight after
        training that is TRAIN mode, so every batch is normalised with its own statistics and the running
        statistics keep moving (the reference never calls ``eval()`` here).  ``bn_mode='eval'`` / ``'train'`` select
        the mode explicitly for this call and restore the previous one afterwards;
        with ``'eval'`` the result no longer depends on how the loader batches the data."""
        if bn_mode not in (None, 'eval', 'train'):
            raise ValueError("bn_mode must be None, 'eval' or 'train'")
        was_training = self.training
        if bn_mode is not None:
            self.train(bn_mode == 'train')
        try:
            n_total = len(loader.dataset)
            latent_dev = torch.zeros(n_total, self.z_dim, device=self.device)   # one D2H copy at the end, no per-batch sync
            i = 0
            for data in self._feed(loader):
                with torch.no_grad():
                    mu, _, _ = self.encode(data)
                latent_dev[i:i + len(mu)] = mu
                i += len(mu)
            latent = latent_dev.cpu().numpy().astype(np.float64)
        finally:
            if bn_mode is not None:
                self.train(was_training)
        return latent
