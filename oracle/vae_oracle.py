"""CPU oracle for the VAE training hot path -- TEST INFRASTRUCTURE ONLY.

This file is a CPU restatement (stock PyTorch-CPU tensor ops, the same ATen
kernels the reference dispatches to) of the algorithm in the reference's
``ava/models/vae.py``.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product path
(``autoencoded-vocal-analysis_amd/``) never does and fails loudly when the HIP
library is missing.

Parity pin: this restatement is checked against golden vectors produced by
importing the real reference in the build container
(``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``;
``tests/test_oracle_golden.py``).  The reference ships no tests or golden
vectors of its own for this path.

Each function cites the reference lines it restates.  ``torch/...`` citations
are into the torch 2.10 wheel, where the reference's arithmetic lives.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

X_DIM = 128 * 128
BN_EPS = 1e-5          # torch/nn/modules/batchnorm.py default, used by vae.py:135-141,162-168
BN_MOMENTUM = 0.1

ENC = [("conv1", "bn1", 1), ("conv2", "bn2", 2), ("conv3", "bn3", 1), ("conv4", "bn4", 2),
       ("conv5", "bn5", 1), ("conv6", "bn6", 2), ("conv7", "bn7", 1)]
DEC = [("convt1", "bn8", 1), ("convt2", "bn9", 2), ("convt3", "bn10", 1), ("convt4", "bn11", 2),
       ("convt5", "bn12", 1), ("convt6", "bn13", 2), ("convt7", "bn14", 1)]
BN_CHANNELS = {"bn1": 1, "bn2": 8, "bn3": 8, "bn4": 16, "bn5": 16, "bn6": 24, "bn7": 24,
               "bn8": 32, "bn9": 24, "bn10": 24, "bn11": 16, "bn12": 16, "bn13": 8, "bn14": 8}


def fresh_running_stats(dtype=torch.float32):
    """BatchNorm2d buffers at construction: mean 0, var 1, counter 0."""
    rs = {}
    for name, c in BN_CHANNELS.items():
        rs[name + ".running_mean"] = torch.zeros(c, dtype=dtype)
        rs[name + ".running_var"] = torch.ones(c, dtype=dtype)
        rs[name + ".num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return rs


class _BNTrain(torch.autograd.Function):
    """Train-mode BatchNorm2d with explicit formulas.

    The per-channel sums are accumulated in float64 and rounded to the tensor
    dtype afterwards, which is what ATen's CPU kernels do for float32 tensors
    (``at::acc_type<float, /*is_cuda=*/false>`` is ``double``); the reference's
    conv1/bn1 gradients are only reproducible to ~1e-6 with that accumulator
    (plain float32 sums give ~5e-3 on those four tensors)."""

    @staticmethod
    def forward(ctx, x, w, b):
        xd = x.double()
        mean = xd.mean(dim=(0, 2, 3))
        var = ((xd - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))      # biased
        mean, var = mean.to(x.dtype), var.to(x.dtype)
        invstd = torch.rsqrt(var + BN_EPS)
        ctx.save_for_backward(x, w, mean, invstd)
        ctx.mark_non_differentiable(mean, var)
        y = (x - mean[None, :, None, None]) * (invstd * w)[None, :, None, None] + b[None, :, None, None]
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        x, w, mean, invstd = ctx.saved_tensors
        n = x.numel() // x.shape[1]
        xc = x - mean[None, :, None, None]
        sum_dy = dy.double().sum(dim=(0, 2, 3))
        dotp = (dy.double() * xc.double()).sum(dim=(0, 2, 3))
        dgamma = (dotp * invstd.double()).to(x.dtype)
        dbeta = sum_dy.to(x.dtype)
        gmean = (sum_dy / n).to(x.dtype)
        k = (dotp * invstd.double() ** 2 / n).to(x.dtype)
        dx = (dy - gmean[None, :, None, None] - xc * k[None, :, None, None]) * (invstd * w)[None, :, None, None]
        return dx, dgamma, dbeta


def batchnorm(x, name, P, running, train, record):
    """BatchNorm2d on NCHW ``x`` (vae.py:217-223,263-269).

    Train: per-channel mean / *biased* variance over (N,H,W); running stats
    updated with momentum 0.1 and the *unbiased* variance.  Eval: running stats."""
    w, b = P[name + ".weight"], P[name + ".bias"]
    if train:
        y, mean, var = _BNTrain.apply(x, w, b)
        n = x.numel() // x.shape[1]
        if running is not None:
            with torch.no_grad():
                rm, rv = running[name + ".running_mean"], running[name + ".running_var"]
                running[name + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.to(rm.dtype)
                unb = var * (n / max(n - 1, 1))
                running[name + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unb.to(rv.dtype)
                running[name + ".num_batches_tracked"] = running[name + ".num_batches_tracked"] + 1
    else:
        mean = running[name + ".running_mean"].to(x.dtype)
        var = running[name + ".running_var"].to(x.dtype)
        inv = torch.rsqrt(var + BN_EPS)
        y = (x - mean[None, :, None, None]) * (inv * w)[None, :, None, None] + b[None, :, None, None]
    if record is not None:
        record[name + ".mean"] = mean.detach().clone()
        record[name + ".var"] = var.detach().clone()
    return y


def _relu(u, name, masks):
    """ReLU (vae.py:217-231,258-268).  ``masks`` (tests only): a dict name -> 0/1 tensor that REPLACES the sign
    decision, ``relu(u) -> u * mask``: with the masks of another evaluation imposed, the network is a smooth function
    of its parameters and two correct evaluations agree to rounding -- no ReLU-flip noise (tests/test_gpu_step.py)."""
    if masks is not None and name in masks:
        return u * masks[name].to(u.dtype)
    return F.relu(u)


def _store(h, act_dtype, name=None, stored=None):
    """Activation storage of this build's bf16 mode (BASELINE configs[4]; not in the reference): the tensors between the
    convolutions are STORED as bfloat16, rounded to nearest even by the producing kernel, and every consumer -- forward
    and backward -- reads the rounded values.  The rounding has no derivative of its own (straight-through): the
    gradient with respect to the stored value is used as the gradient with respect to the unrounded one.

    ``stored`` (tests only): a dict name -> the tensor another evaluation actually stored; it REPLACES the rounding
    (value = that tensor, gradient straight through).  Needed for a tight comparison: two evaluations of a
    bf16-rounded network decorrelate within a few layers -- a perturbation eps flips the rounding of a fraction
    eps/2^-8 of the elements by one bf16 ulp, which is a perturbation sqrt(eps * 2^-8) for the next layer, with fixed
    point 2^-8 -- and then disagree on ~0.4 % of the ReLU masks, i.e. by 5-10 % on gradients (measured even between
    an fp32 and an fp64 evaluation of the same rounded oracle, tools/bf16_probe.py)."""
    if stored is not None and name in stored:
        return h + (stored[name].to(h.dtype) - h).detach()
    if act_dtype is None:
        return h
    return h + (h.detach().to(act_dtype).to(h.dtype) - h.detach())


# Convolutions that compute in bf16 ARITHMETIC when this build's activations are bfloat16 (VAE(act_dtype='bfloat16'), BASELINE
# configs[4] "bf16 conv + fp32 ELBO"; not in the reference): the twelve layers with >= 8 channels on both sides -- the matrix-core
# layers.  conv1 and convt7 (one channel on one side: packed-FMA kernels bound by HBM, not by arithmetic) keep fp32 products.
BF16_MATH_LAYERS = frozenset(["conv%d" % i for i in range(2, 8)] + ["convt%d" % i for i in range(1, 7)])


def _q_bf16(t):
    """Round to bfloat16 (nearest even) with a straight-through gradient: the operand rounding of the bf16-arithmetic mode."""
    return t + (t.detach().to(torch.bfloat16).to(t.dtype) - t.detach())


def _conv_operands(h, w, layer, bf16_math):
    """The two operands of a layer's products: as they are (reference arithmetic), or -- bf16 arithmetic -- the BatchNorm output
    and the weights each rounded to bfloat16; the products of bfloat16 values are exact in fp32 and are accumulated in fp32
    (here: in the oracle's dtype).  Backward: the rounding has no derivative of its own, so the data gradient is taken against
    the ROUNDED weights and the weight gradient against the ROUNDED input -- the derivative of the function evaluated."""
    if bf16_math and layer in BF16_MATH_LAYERS:
        return _q_bf16(h), _q_bf16(w)
    return h, w


def encode(P, x, running=None, train=True, record=None, masks=None, act_dtype=None, stored=None, bf16_math=False):
    """vae.py:216-233.  ``x`` is ``[B,128,128]``; returns mu [B,z], u [B,z], d [B,z]."""
    h = x.unsqueeze(1)
    for conv, bn, stride in ENC:
        h = batchnorm(h, bn, P, running, train, record)
        h, w = _conv_operands(h, P[conv + ".weight"], conv, bf16_math)
        h = _relu(F.conv2d(h, w, P[conv + ".bias"], stride=stride, padding=1), conv, masks)
        if conv != "conv7":
            h = _store(h, act_dtype, conv, stored)               # conv7's output feeds the fp32 fully connected layers
        if record is not None:
            record[conv + ".out"] = h
    h = h.reshape(h.shape[0], -1)                               # vae.py:224 (NCHW flatten; 8192 features at 128 x 128)
    h = _relu(F.linear(h, P["fc1.weight"], P["fc1.bias"]), "fc1", masks)
    h = _relu(F.linear(h, P["fc2.weight"], P["fc2.bias"]), "fc2", masks)
    if record is not None:
        record["fc2.out"] = h
    mu = F.linear(_relu(F.linear(h, P["fc31.weight"], P["fc31.bias"]), "fc31", masks), P["fc41.weight"], P["fc41.bias"])
    u = F.linear(_relu(F.linear(h, P["fc32.weight"], P["fc32.bias"]), "fc32", masks), P["fc42.weight"], P["fc42.bias"])
    a = F.linear(_relu(F.linear(h, P["fc33.weight"], P["fc33.bias"]), "fc33", masks), P["fc43.weight"], P["fc43.bias"])
    if record is not None:
        record["logd"] = a
    return mu, u, torch.exp(a)                                   # vae.py:232


def decode(P, z, running=None, train=True, record=None, masks=None, x_shape=(128, 128), act_dtype=None, stored=None,
           bf16_math=False):
    """vae.py:258-270.  Returns x_rec ``[B, H*W]`` (``[B,16384]`` at the reference's X_SHAPE; ``x_shape`` other than
    (128, 128) is this build's size extension: fc8.out = 32 * H/8 * W/8)."""
    h = _relu(F.linear(z, P["fc5.weight"], P["fc5.bias"]), "fc5", masks)
    h = _relu(F.linear(h, P["fc6.weight"], P["fc6.bias"]), "fc6", masks)
    h = _relu(F.linear(h, P["fc7.weight"], P["fc7.bias"]), "fc7", masks)
    h = _relu(F.linear(h, P["fc8.weight"], P["fc8.bias"]), "fc8", masks)
    if record is not None:
        record["fc8.out"] = h
    h = h.reshape(-1, 32, x_shape[0] // 8, x_shape[1] // 8)     # vae.py:262 (32 x 16 x 16 at 128 x 128)
    h = _store(h, act_dtype, "fc8", stored)                      # the NHWC copy convt1 reads
    for i, (convt, bn, stride) in enumerate(DEC):
        h = batchnorm(h, bn, P, running, train, record)
        h, w = _conv_operands(h, P[convt + ".weight"], convt, bf16_math)
        h = F.conv_transpose2d(h, w, P[convt + ".bias"], stride=stride, padding=1, output_padding=stride - 1)
        if i < 6:
            h = _store(_relu(h, convt, masks), act_dtype, convt, stored)   # no ReLU after convt7 (vae.py:269)
        if record is not None:
            record[convt + ".out"] = h
    return h.reshape(-1, x_shape[0] * x_shape[1])


def rsample(mu, u, d, eps_w, eps_d):
    """``LowRankMultivariateNormal.rsample`` for rank 1
    (torch/distributions/lowrank_multivariate_normal.py:214-223):
    ``z = mu + u*eps_W + sqrt(d)*eps_D`` with eps_W [B,1] drawn before eps_D [B,z]."""
    return mu + u * eps_w + torch.sqrt(d) * eps_d


def entropy(u, d):
    """Per-sample entropy of N(mu, u u^T + diag(d))
    (torch/distributions/lowrank_multivariate_normal.py:17-38,242-252):
    ``H = 0.5*(z*(1+ln 2pi) + ln K + sum ln d)``, ``K = 1 + sum u^2/d``."""
    zdim = u.shape[1]
    K = 1.0 + (u * u / d).sum(dim=1)
    return 0.5 * (zdim * (1.0 + math.log(2 * math.pi)) + torch.log(K) + torch.log(d).sum(dim=1))


def loss_terms(x, x_rec, z, u, d, model_precision=10.0):
    """The three batch sums of vae.py:316-323 plus the assembled -ELBO.

    ``-elbo = 0.5*(sum z^2 + zdim*ln2pi) + 0.5*X_DIM*ln(2pi/prec)
              + 0.5*prec*SSE - sum_b H_b``; the two constants are added once
    per call, not per sample."""
    zdim = z.shape[1]
    sum_z2 = (z * z).sum()
    sse = ((x.reshape(x.shape[0], -1) - x_rec) ** 2).sum(dim=1).sum()
    sum_h = entropy(u, d).sum()
    c1 = 0.5 * zdim * math.log(2 * math.pi)
    c2 = 0.5 * x[0].numel() * math.log(2 * math.pi / model_precision)      # X_DIM (vae.py:35,318)
    loss = 0.5 * sum_z2 + c1 + c2 + 0.5 * model_precision * sse - sum_h
    return loss, sum_z2, sse, sum_h


def forward(P, x, eps_w, eps_d, running=None, train=True, model_precision=10.0, record=None, masks=None,
            act_dtype=None, stored=None, bf16_math=False):
    """vae.py:311-327 with the two normal draws injected.  Raises ValueError
    like the reference's argument validation when ``d`` is not positive."""
    mu, u, d = encode(P, x, running, train, record, masks, act_dtype, stored, bf16_math)
    if not bool((d > 0).all()):
        raise ValueError("cov_diag must be positive")
    z = rsample(mu, u, d, eps_w, eps_d)
    x_rec = decode(P, z, running, train, record, masks, tuple(x.shape[1:]), act_dtype, stored, bf16_math)
    loss, sum_z2, sse, sum_h = loss_terms(x, x_rec, z, u, d, model_precision)
    out = dict(loss=loss, sum_z2=sum_z2, sse=sse, sum_h=sum_h, mu=mu, u=u, d=d, z=z, x_rec=x_rec)
    return out


def latent_backward(g, u, d, eps_w, eps_d):
    """Closed-form gradients of the latent block, given
    ``g = z + dL_dec/dz``: returns (dmu, du, da) with ``a = log d``."""
    K = 1.0 + (u * u / d).sum(dim=1, keepdim=True)
    dmu = g
    du = g * eps_w - (u / d) / K
    da = 0.5 * g * eps_d * torch.sqrt(d) - 0.5 * (1.0 - u * u / (d * K))
    return dmu, du, da


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """One torch.optim.Adam update, single-tensor path (torch/optim/adam.py:414-547),
    defaults of vae.py:119.  ``step`` is the 1-based step count after increment.
    Operates in the dtype of the inputs; returns (p, m, v)."""
    m = m + (g - m) * (1 - b1)                                   # exp_avg.lerp_(grad, 1-beta1)
    v = v * b2 + (1 - b2) * g * g                                # exp_avg_sq.mul_().addcmul_()
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def to_params(np_params, dtype=torch.float32, requires_grad=False):
    out = {}
    for k, v in np_params.items():
        t = torch.tensor(np.asarray(v), dtype=dtype)
        t.requires_grad_(requires_grad)
        out[k] = t
    return out


def train_step(P, x, eps_w, eps_d, running, opt_state, lr=1e-3, model_precision=10.0):
    """zero_grad -> forward -> backward -> Adam.step (vae.py:348-353) on a dict of
    leaf tensors ``P`` (requires_grad).  ``opt_state`` = {'step': int, 'm': {}, 'v': {}}.
    Returns (loss float, grads dict).  Gradients come from torch autograd on the
    restated forward (the reference does the same, vae.py:352)."""
    for t in P.values():
        t.grad = None
    out = forward(P, x, eps_w, eps_d, running, True, model_precision)
    out["loss"].backward()
    grads = {k: t.grad.detach().clone() for k, t in P.items()}
    opt_state["step"] += 1
    with torch.no_grad():
        for k, t in P.items():
            m = opt_state["m"].get(k, torch.zeros_like(t))
            v = opt_state["v"].get(k, torch.zeros_like(t))
            p_new, m, v = adam_step(t.detach(), grads[k], m, v, opt_state["step"], lr)
            t.copy_(p_new)
            opt_state["m"][k], opt_state["v"][k] = m, v
    return float(out["loss"].detach()), grads, out
