"""Whole-path parity of the HIP VAE (through ava_amd.vae.VAE -> C ABI) against
(1) golden vectors captured from the real reference (tests/golden/*.npz) and
(2) the CPU oracle on the same seeded inputs; plus size-independent properties at the
benchmark's full batch of 256.  Tolerances: ELBO 1e-5 relative (north-star asks 1e-4);
gradients as in tests/test_oracle_golden.py (fp32 noise floor of the reference itself)."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden, sample_idx, GOLDEN
from gpu_util import build_model, rel
from ava_amd import synthetic as syn
from ava_amd.layout import param_specs
from oracle import vae_oracle as O

# Gradient tolerances against the reference goldens.  Two correct fp32 evaluations of this network do not agree on
# every ReLU mask: a batch of 8 has ~3e7 convolution pre-activations of O(0.1), so a handful lie within one fp32
# rounding of zero, and ANY change of summation order upstream (another GEMM tiling, another reduction tree) flips a
# few of them; each flip moves the gradient norms of the tensors upstream of it by up to a few 1e-4 (observed 3.8e-4
# on conv2.weight when the small fully connected products changed kernels; the reference's own fp32 result is 2e-2
# away from an fp64 evaluation on conv1/bn1 at this fixture, tools/flip_hunt.py).  Everything
# that does not pass through a mask decision is held to 1e-5 (loss terms, BatchNorm statistics) and the kernels are
# held to 2e-5 with the masks given as inputs (test_gpu_kernels.py).
# Measured against an fp64 evaluation (test_gradients_against_fp64_noise_floor, profiles/r02/grad_fp64.json): the HIP
# path is 4e-7 (B=8) / 1.6e-4 (B=64) / 5e-5 (B=256) away globally, the reference-equivalent fp32 CPU oracle 4e-6 / 8e-5 /
# 5e-5: against the fp32 goldens the bound is therefore the goldens' own distance from fp64 -- 1e-4 at B=8 outside
# conv1/bn1 (whose B=8 golden carries one ReLU flip: 9e-3), 1e-2 at B=64 (same values as tests/test_oracle_golden.py).
FLIP_TOL = 2e-2
GTOL = {8: 1e-4, 64: 1e-2}   # B = 64: the golden's own distance from fp64 (its gradients carry the reference's ReLU flips)


def _oracle_grads(fp, x, ew, ed, dtype):
    """gradients (float64 numpy, by name) and loss of the CPU oracle evaluated in `dtype`"""
    P = O.to_params(fp, dtype=dtype, requires_grad=True)
    out = O.forward(P, torch.as_tensor(x, dtype=dtype), torch.as_tensor(ew, dtype=dtype), torch.as_tensor(ed, dtype=dtype),
                    None, True)
    out["loss"].backward()
    return {k: v.grad.double().numpy().ravel() for k, v in P.items()}, float(out["loss"].detach())


def _errors_vs_fp64(g, g64, z):
    """per-tensor relative L2 error against the fp64 evaluation, and the global one.  conv1 / bn1 gradients are sums
    of terms that cancel to ~1e-6 of their magnitude, so "relative" there is taken against the conv1.bias gradient norm
    (same summands) when that is larger -- as in tests/test_oracle_golden.py."""
    cb = np.linalg.norm(g64["conv1.bias"])
    per, num, den = {}, 0.0, 0.0
    for s in param_specs(z):
        r = g64[s.name]
        scale = max(np.linalg.norm(r), cb if s.layer in ("conv1", "bn1") else 0.0, 1e-300)
        d = np.linalg.norm(g[s.name] - r)
        per[s.name] = d / scale
        num += d * d
        den += np.linalg.norm(r) ** 2
    return per, (num / den) ** 0.5


def _assert_within_noise_floor(got, g32, g64, z, report=None):
    """Distance from an fp64 evaluation, HIP next to the fp32 CPU oracle (= the reference's own arithmetic) on the same
    inputs.  Both carry rounding plus ReLU-mask flips (a pre-activation within one rounding of zero); which units flip
    is a different random draw in each implementation and a tensor's error is set by the one to three flips upstream of
    it, so tensor-by-tensor ratios scatter (0.4 ... 6 observed, profiles/r02/grad_fp64.log) and only the GLOBAL relative
    L2 error is asserted here (HIP within 4x the oracle's; observed 0.1x ... 1.9x).  The per-tensor bound without flip
    noise is _masked_fp64_grad_errors below (1e-4 on every tensor at every batch size)."""
    eh, gh = _errors_vs_fp64(got, g64, z)
    eo, go = _errors_vs_fp64(g32, g64, z)
    if report is not None:
        report.update({"hip_global": gh, "o32_global": go, "hip_max": max(eh.values()), "o32_max": max(eo.values())})
    assert gh <= 4 * go + 1e-6, (gh, go)


def _hip_masks(model, B):
    """0/1 ReLU masks of the forward that just ran on the device, by oracle layer name (NCHW for conv layers)."""
    m = {}
    H, W = model.x_shape
    # (channels, divisor of the spectrogram size) of every ReLU-ed conv output
    ch = {"conv1": (8, 1), "conv2": (8, 2), "conv3": (16, 2), "conv4": (16, 4), "conv5": (24, 4), "conv6": (24, 8),
          "convt1": (24, 8), "convt2": (24, 4), "convt3": (16, 4), "convt4": (16, 2), "convt5": (8, 2), "convt6": (8, 1)}
    for i in range(1, 7):
        c, d = ch["conv%d" % i]
        m["conv%d" % i] = (model._workspace_tensor("y%d" % i, (B, H // d, W // d, c)) > 0).permute(0, 3, 1, 2).cpu()
        c, d = ch["convt%d" % i]
        m["convt%d" % i] = (model._workspace_tensor("d%d" % i, (B, H // d, W // d, c)) > 0).permute(0, 3, 1, 2).cpu()
    m["conv7"] = (model._workspace_tensor("y7", (B, H // 8, W // 8, 32)) > 0).permute(0, 3, 1, 2).cpu()
    for name, n in (("h1", 1024), ("h2", 256), ("h5", 64), ("h6", 256), ("h7", 1024), ("f8", 32 * (H // 8) * (W // 8))):
        key = {"h1": "fc1", "h2": "fc2", "h5": "fc5", "h6": "fc6", "h7": "fc7", "f8": "fc8"}[name]
        m[key] = (model._workspace_tensor(name, (B, n)) > 0).cpu()
    h3 = (model._workspace_tensor("h3", (B, 192)) > 0).cpu()
    m["fc31"], m["fc32"], m["fc33"] = h3[:, :64], h3[:, 64:128], h3[:, 128:]
    return m


def _masked_fp64_grad_errors(model, fp, x, ew, ed, z):
    """Whole-path gradient error WITHOUT ReLU-flip noise: the fp64 oracle is evaluated with the HIP forward's own
    ReLU masks imposed (relu(u) -> u * mask), which makes it the exact derivative of the function the device
    evaluated; what remains is fp32 rounding.  Returns (per-tensor relative L2 errors, relative loss difference)."""
    B = x.shape[0]
    named = dict(model.named_parameters())
    got = {n: p.grad.detach().cpu().double().numpy().ravel() for n, p in named.items()}
    masks = _hip_masks(model, B)
    P = O.to_params(fp, dtype=torch.float64, requires_grad=True)
    out = O.forward(P, torch.as_tensor(x, dtype=torch.float64), torch.as_tensor(ew, dtype=torch.float64),
                    torch.as_tensor(ed, dtype=torch.float64), None, True, masks=masks)
    out["loss"].backward()
    g64 = {k: v.grad.numpy().ravel() for k, v in P.items()}
    per, _ = _errors_vs_fp64(got, g64, z)
    return per, float(out["loss"].detach())


def fixed_noise(model, B, z, sw=2002, sd=3003):
    ew, ed = syn.noise(B, z, sw, sd)
    model.noise_source = lambda b, zz: (ew, ed)
    return ew, ed


@pytest.mark.parametrize("B,z", [(8, 32), (8, 64), (64, 32)])
def test_train_step_matches_reference_golden(B, z):
    G = load_golden("step_B%d_z%d.npz" % (B, z))
    model = build_model(z)
    fixed_noise(model, B, z)
    x = torch.from_numpy(syn.spectrograms(B))
    loss = model.forward(x)
    lb = model._loss_buf.cpu().numpy()
    assert rel(float(loss.item()), G["s1.loss"]) < 1e-5
    assert rel(lb[1], G["s1.sum_z2"]) < 1e-5 and rel(lb[2], G["s1.sse"]) < 1e-5 and rel(lb[3], G["s1.sum_h"]) < 1e-5
    assert rel(model._workspace_tensor("mu", (B, z))[:2].cpu(), G["s1.mu"]) < 1e-4
    assert rel(model._workspace_tensor("u", (B, z))[:2].cpu(), G["s1.u"]) < 1e-4
    assert rel(model._workspace_tensor("d", (B, z))[:2].cpu(), G["s1.d"]) < 1e-4
    assert rel(model._workspace_tensor("z", (B, z))[:2].cpu(), G["s1.z"]) < 1e-4
    xr = model._workspace_tensor("xrec", (B * 16384,)).cpu().numpy()
    assert rel(xr[G["s1.xrec_idx"]], G["s1.xrec"]) < 1e-4
    bn_save = model._workspace_tensor("bn_save", (14, 4, 32)).cpu().numpy()
    for i in range(1, 15):
        c = len(G["s1.bn%d.mean" % i])
        assert rel(bn_save[i - 1, 0, :c], G["s1.bn%d.mean" % i]) < 1e-4
        assert rel(1.0 / bn_save[i - 1, 1, :c] ** 2 - 1e-5, G["s1.bn%d.var" % i]) < 1e-4
    loss.backward()
    named = dict(model.named_parameters())
    # Gradients.  B = 8: 1e-4 per tensor (FLIP_TOL on conv1 / bn1, whose golden carries one flip).  B = 64: this golden's gradients
    # carry the ReLU flips of the reference's own fp32 evaluation (it is 1e-2 away from fp64), so 1e-2 is all it can assert
    # (kept as a coarse check rather than dropped); the tight B = 64 evidence is
    # test_flip_free_reference_golden[64] (the REAL reference, 1e-4 per tensor, nothing masked) and the mask-imposed fp64 suite.
    check_grads = True
    for s in (param_specs(z) if check_grads else ()):
        g = named[s.name].grad.cpu().numpy().ravel()
        sens = s.layer in ("conv1", "bn1")
        tol = FLIP_TOL if sens else GTOL[B]
        ref_scale = max(float(G["s1.gradnorm." + s.name]), float(G["s1.gradnorm.conv1.bias"]) if sens else 0.0)
        gn = np.sqrt((g.astype(np.float64) ** 2).sum())
        assert abs(gn - float(G["s1.gradnorm." + s.name])) < tol * ref_scale, s.name
        scale = max(np.abs(g).max(), ref_scale if sens else 0.0)
        np.testing.assert_allclose(g[sample_idx(g.size, s.index)], G["s1.grad." + s.name], rtol=10 * tol,
                                   atol=tol * scale, err_msg=s.name)
    for i in range(1, 15):
        bn = getattr(model, "bn%d" % i)
        assert rel(bn.running_mean.cpu(), G["s1.bn%d.running_mean" % i]) < 1e-5
        assert rel(bn.running_var.cpu(), G["s1.bn%d.running_var" % i]) < 1e-5
        assert int(bn.num_batches_tracked) == int(G["s1.bn%d.num_batches_tracked" % i])
    model.optimizer.step()
    st = model.optimizer.state_dict()["state"]
    for s in param_specs(z):
        idx = sample_idx(s.numel, s.index)
        sens = s.layer in ("conv1", "bn1")
        tol = FLIP_TOL if sens else GTOL.get(B, 0.0)
        gs = float(G["s1.gradnorm.conv1.bias"]) if sens else 0.0
        m = st[s.index]["exp_avg"].cpu().numpy().ravel()
        v = st[s.index]["exp_avg_sq"].cpu().numpy().ravel()
        if check_grads:
            np.testing.assert_allclose(m[idx], G["s1.exp_avg." + s.name], rtol=20 * tol, atol=tol * max(np.abs(m).max(), 0.1 * gs))
            np.testing.assert_allclose(v[idx], G["s1.exp_avg_sq." + s.name], rtol=40 * tol,
                                       atol=tol * max(np.abs(v).max(), 1e-3 * gs * gs))
        pv = named[s.name].detach().cpu().numpy().ravel()
        # Adam's first step moves every entry by ~lr = 1e-3: most sampled entries must land on the reference's value
        # (a model Adam never touched fails this), the rest (gradient entries near zero whose sign a ReLU-mask flip
        # changes) within the step size.  The update itself is pinned entry by entry in
        # test_adam_delta_matches_reference_golden.
        dv = np.abs(pv[idx] - G["s1.val." + s.name])
        assert np.mean(dv < 2e-5) >= (0.5 if sens else 0.8), s.name
        assert dv.max() < 2.2e-3, s.name
        assert float(st[s.index]["step"]) == float(G["s1.adam_step"])


def test_three_steps_then_eval_matches_golden():
    """Loss trajectory over three Adam steps and the eval-mode (running statistics) forward."""
    B, z = 8, 32
    G = load_golden("step_B8_z32.npz")
    model = build_model(z)
    fixed_noise(model, B, z)
    x = torch.from_numpy(syn.spectrograms(B))
    for step in (1, 2, 3):
        model.optimizer.zero_grad()
        loss = model.forward(x)
        assert rel(float(loss.item()), G["s%d.loss" % step]) < (1e-5 if step == 1 else 1e-4)
        loss.backward()
        model.optimizer.step()
    for i in range(1, 15):
        assert rel(getattr(model, "bn%d" % i).running_var.cpu(), G["final.bn%d.running_var" % i]) < 1e-4
    model.eval()
    with torch.no_grad():
        # behind three Adam steps: the first steps are sign-like (g / sqrt(g^2)), so rounding noise in near-zero gradient
        # entries is an O(lr) parameter difference and two fp32 trajectories separate (tools/traj.py: 2e-5 at step 3 on the
        # training loss, more on the eval loss, which also sees the running statistics of all three steps); the freshly
        # built model below pins the eval-mode forward itself to 1e-5
        assert rel(float(model.forward(x).item()), G["eval.loss"]) < 1e-3
    fresh = build_model(z, train=False)
    fixed_noise(fresh, B, z)
    with torch.no_grad():
        assert rel(float(fresh.forward(x).item()), G["eval_fresh.loss"]) < 1e-5


def test_six_step_trajectory_matches_oracle():
    """Training dynamics: 6 Adam steps on three alternating batches (B = 16, z = 32), the loss of every step against
    the CPU oracle run on the same inputs and noise.  Adam's first steps are sign-like (g / sqrt(g^2)), so rounding
    noise in near-zero gradient entries becomes O(lr) parameter differences and the two fp32 trajectories separate
    exponentially (measured: 1e-7, 2e-7, 2e-5, 5e-5, 2e-4, 6e-5 ... 1.5e-3 at step 12, tools/traj.py): steps 1-2 are
    held to 1e-5, the rest to 1e-3."""
    B, z, steps = 16, 32, 6
    xs = [torch.from_numpy(syn.spectrograms(B, salt=77 + i)) for i in range(3)]
    model = build_model(z)
    ew, ed = fixed_noise(model, B, z)
    P = O.to_params(syn.fixture_parameters(z), requires_grad=True)
    running = O.fresh_running_stats()
    opt = {"step": 0, "m": {}, "v": {}}
    for step in range(steps):
        x = xs[step % 3]
        model.optimizer.zero_grad()
        loss = model.forward(x)
        loss.backward()
        model.optimizer.step()
        want, _, _ = O.train_step(P, x, torch.from_numpy(ew), torch.from_numpy(ed), running, opt)
        assert rel(float(loss.item()), want) < (1e-5 if step < 2 else 1e-3), step
    for i in range(1, 15):
        assert rel(getattr(model, "bn%d" % i).running_mean.cpu(), running["bn%d.running_mean" % i]) < 5e-3


@pytest.mark.parametrize("B,z", [(1, 32), (5, 32), (37, 32), (5, 8), (3, 30), (6, 100), (2, 128)],
                         ids=["B1", "B5", "B37", "z8", "z30", "z100", "z128"])
def test_forward_backward_matches_oracle_odd_batches(B, z):
    """Ragged batch sizes (visualize uses 5, shotgun_movie batch 1: vae.py:503-510, shotgun_movie.py:116-120) and
    latent sizes other than 32 / 64 (z_dim is a constructor argument, vae.py:80; 30 is not a multiple of 4, so the
    fully connected products around the latent take the scalar-load GEMM path)."""
    model = build_model(z)
    ew, ed = fixed_noise(model, B, z, 11, 12)
    x = torch.from_numpy(syn.spectrograms(B, salt=77))
    loss = model.forward(x)
    loss.backward()
    per, l64 = _masked_fp64_grad_errors(model, syn.fixture_parameters(z), x.numpy(), ew, ed, z)
    assert rel(float(loss.item()), l64) < 1e-6
    bad = {k: v for k, v in per.items() if v > 1e-4}       # SURVEY Appendix B's bound, on every tensor
    assert not bad, bad


def test_encode_decode_and_get_latent_golden():
    """get_latent on a fresh module runs BatchNorm in TRAIN mode and keeps updating the running
    statistics (vae.py:538-547 never calls eval())."""
    G = load_golden("get_latent.npz")
    model = build_model(32)
    loader = syn.get_synthetic_data_loaders(16, batch_size=8, shuffle=(False, False))["train"]
    lat = model.get_latent(loader)
    assert lat.dtype == np.float64 and lat.shape == (16, 32)
    assert rel(lat, G["latent"]) < 1e-4
    for i in range(1, 8):
        assert rel(getattr(model, "bn%d" % i).running_mean.cpu(), G["after.bn%d.running_mean" % i]) < 1e-5
        assert int(getattr(model, "bn%d" % i).num_batches_tracked) == 2
    for i in range(8, 15):
        assert int(getattr(model, "bn%d" % i).num_batches_tracked) == 0
    # encode -> (mu,u,d), decode(z) consistent with forward's x_rec
    B = 6
    model2 = build_model(32)
    ew, ed = fixed_noise(model2, B, 32)
    x = torch.from_numpy(syn.spectrograms(B))
    with torch.no_grad():
        _, zs, rec = model2.forward(x, return_latent_rec=True)
        mu, u, d = model2.encode(x)
        assert u.shape == (B, 32, 1) and d.min() > 0
        z2 = mu + u[..., 0] * torch.from_numpy(ew).cuda() + torch.sqrt(d) * torch.from_numpy(ed).cuda()
        assert rel(z2.cpu(), zs) < 1e-5
        rec2 = model2.decode(torch.from_numpy(zs))
        assert rec2.shape == (B, 16384) and rel(rec2.cpu().numpy().reshape(B, 128, 128), rec) < 1e-5


def test_two_shard_data_parallel_equivalence():
    """an N-rank step == each shard stepped from identical weights with its own noise,
    gradients summed.  Emulated on one GPU; pinned by the reference-generated ddp2 golden."""
    G = load_golden("ddp2.npz")
    x = syn.spectrograms(16)
    ew, ed = syn.noise(16, 32)
    total = None
    for r in range(2):
        model = build_model(32)
        sl = slice(8 * r, 8 * r + 8)
        model.noise_source = lambda b, zz, sl=sl: (ew[sl], ed[sl])
        loss = model.forward(torch.from_numpy(x[sl]))
        assert rel(float(loss.item()), G["shard%d.loss" % r]) < 1e-5
        loss.backward()
        g = model._grads.double().clone()
        total = g if total is None else total + g
    for s in param_specs(32):
        o, n, _ = model._arena_views[s.name]
        gn = float(total[o:o + n].norm())
        sens = s.layer in ("conv1", "bn1")
        ref_scale = max(float(G["gradnorm." + s.name]), float(G["gradnorm.conv1.bias"]) if sens else 0.0)
        assert abs(gn - float(G["gradnorm." + s.name])) < (FLIP_TOL if sens else 3e-3) * ref_scale, s.name   # shard 1 has a ReLU flip upstream of conv2 (4e-4 in the oracle)


def test_backward_in_parts_equals_whole():
    """ava_backward_part(0..n-1) (the data-parallel overlap path) == ava_backward, bit for bit, and the gradient
    buckets tile the arena in the order tail (fc8 + decoder), fc1.weight, the rest of the middle (fc1.bias..fc7), head
    (encoder)."""
    import ctypes
    from ava_amd import _lib
    lib = _lib.load()
    B, z = 8, 32
    x = torch.from_numpy(syn.spectrograms(B)).cuda()
    nparts = lib.ava_backward_num_parts()
    assert nparts == 4
    grads = []
    for split in (False, True):
        model = build_model(z)
        fixed_noise(model, B, z)
        model._forward_device(x, need_grad=True)
        if split:
            for part in range(nparts):
                _lib.check(lib.ava_backward_part(model._handle, x.data_ptr(), B, part, _lib.stream()), "part")
        else:
            _lib.check(lib.ava_backward(model._handle, x.data_ptr(), B, _lib.stream()), "whole")
        grads.append(model._grads.clone())
    assert torch.equal(grads[0], grads[1])
    rng = []
    for b in range(nparts):
        o, c = ctypes.c_int64(), ctypes.c_int64()
        assert lib.ava_grad_bucket(model._handle, b, ctypes.byref(o), ctypes.byref(c)) == 0
        rng.append((o.value, c.value))
    assert rng[3][0] == 0 and rng[3][1] == rng[1][0] and rng[1][0] + rng[1][1] == rng[2][0] and rng[2][0] + rng[2][1] == rng[0][0]
    assert rng[0][0] + rng[0][1] == model._grads.numel()
    assert rng[0][0] == model._arena_views["fc8.weight"][0] and rng[1][0] == model._arena_views["fc1.weight"][0]
    assert rng[2][0] == model._arena_views["fc1.bias"][0] and rng[1][1] >= model._arena_views["fc1.weight"][1]
    assert lib.ava_grad_bucket(model._handle, nparts, ctypes.byref(o), ctypes.byref(c)) != 0


def test_forward_noise_equals_fill_normal_then_forward():
    """ava_forward_noise draws the rsample noise inside the forward's first launch: same counter stream as
    ava_fill_normal (bit for bit) and therefore the same loss as ava_forward fed with that noise."""
    from ava_amd import _lib
    lib = _lib.load()
    B, z = 5, 32
    x = torch.from_numpy(syn.spectrograms(B)).cuda()
    n = B * (z + 1)
    model = build_model(z)
    model.train()
    model.noise_source = None
    model._rng_seed, model._rng_offset = 123, 1000
    loss_a = float(model._forward_device(x, need_grad=True).item())
    assert model._rng_offset == 1000 + n
    eps_a = model._eps[:n].clone()
    buf = torch.empty(n, device="cuda")
    _lib.check(lib.ava_fill_normal(buf.data_ptr(), n, 123, 1000, _lib.stream()), "fill")
    assert torch.equal(buf, eps_a)
    assert abs(float(buf.mean())) < 0.5 and 0.5 < float(buf.std()) < 1.5
    ref = build_model(z)
    ref.train()
    ref.noise_source = lambda b, zz: (buf[:B], buf[B:].view(B, z))
    assert float(ref._forward_device(x, need_grad=True).item()) == loss_a
    # eval mode (no statistics to fuse): the noise falls back to its own launch, same values
    model.eval(); ref.eval()
    model._rng_offset = 1000
    assert float(model._forward_device(x, need_grad=False).item()) == float(ref._forward_device(x, need_grad=False).item())


def test_profile_passes_agree():
    """ava_profile_enable(1) brackets every launch group with HIP events, ava_profile_enable(2) only the runs of
    same-family kernels (what bench.py's roofline uses): the coarse pass must record far fewer events, leave the results
    untouched and attribute about the same time to the conv family (the fine pass is stretched by its own events)."""
    import ctypes
    from ava_amd import _lib
    lib = _lib.load()
    B, z = 64, 32
    x = torch.from_numpy(syn.spectrograms(B)).cuda()
    model = build_model(z)
    fixed_noise(model, B, z)
    model.train()

    def step():
        model.optimizer.zero_grad()
        model._forward_device(x, need_grad=True)
        model._backward_device(x)
        model.optimizer.step()

    step()
    fam = (0, 1, 2, 3, 8)                                 # conv fwd / bwd-data / wgrad, BatchNorm, pack
    out = {}
    for mode in (1, 2):
        ms = (ctypes.c_float * 16)()
        cnt = (ctypes.c_int * 16)()
        assert lib.ava_profile_enable(model._handle, mode) == 0
        for _ in range(5):
            step()
            assert lib.ava_profile_read(model._handle, ms, cnt) == 9
        lib.ava_profile_enable(model._handle, 0)
        out[mode] = (sum(ms[i] for i in fam) / 5, sum(ms[i] for i in range(9)) / 5, sum(cnt[i] for i in range(9)) // 5)
    (conv1, all1, n1), (conv2, all2, n2) = out[1], out[2]
    assert n1 >= 50 and 4 <= n2 <= 30, (n1, n2)                 # fine pass: one bracket per launch group (60 at this build)
    assert 0 < conv2 <= conv1 * 1.05 and conv2 > 0.6 * conv1, (conv1, conv2)
    assert 0 < all2 <= all1 * 1.05, (all1, all2)
    loss_profiled = float(model._loss_buf[0].item())
    ref = build_model(z)
    fixed_noise(ref, B, z)
    ref.train()
    for _ in range(11):
        ref.optimizer.zero_grad()
        ref._forward_device(x, need_grad=True)
        ref._backward_device(x)
        ref.optimizer.step()
    assert float(ref._loss_buf[0].item()) == loss_profiled   # timing never changes results


def test_harness_train_loop_checkpoint_golden(tmp_path):
    """train_epoch / test_epoch / train_loop side effects and the checkpoint layout (vae.py:330-472)."""
    G = load_golden("harness.npz")
    manifest = json.load(open(os.path.join(GOLDEN, "checkpoint_manifest.json")))
    B, nb, z = 8, 2, 32
    from ava_amd.vae import VAE
    model = build_model(z)
    model.save_dir = str(tmp_path)
    loaders = syn.get_synthetic_data_loaders(B * nb, batch_size=B, shuffle=(False, False))
    loaders["test"] = loaders["train"]
    queue = []
    for tag in (0, 100, 1, 101):
        for k in range(nb):
            queue.append(syn.noise(B, z, 2002 + 10 * k + tag, 3003 + 10 * k + tag))
    model.noise_source = lambda b, zz: queue.pop(0)
    model.train_loop(loaders, epochs=2, test_freq=1, save_freq=1, vis_freq=None)
    assert not queue
    assert model.epoch == int(G["epoch"]) == 2
    assert rel(model.loss["train"][0], G["train_loss"][0]) < 1e-5
    assert rel(model.loss["test"][0], G["test_loss"][0]) < 1e-3          # behind the first Adam step
    assert rel(model.loss["train"][1], G["train_loss"][1]) < 1e-3
    assert sorted(os.listdir(tmp_path)) == manifest["files"] == ["checkpoint_001.tar"]
    ck = torch.load(os.path.join(tmp_path, "checkpoint_001.tar"), weights_only=True)
    assert list(ck.keys()) == manifest["keys"]
    assert ck["epoch"] == manifest["epoch"] and ck["z_dim"] == manifest["z_dim"] and ck["lr"] == manifest["lr"]
    for name, entries in manifest["layers"].items():
        assert list(ck[name].keys()) == manifest["layer_key_order"][name], name
        for k, (shape, dtype) in entries.items():
            assert list(ck[name][k].shape) == shape and str(ck[name][k].dtype) == dtype, (name, k)
    pg = ck["optimizer_state"]["param_groups"]
    assert len(pg) == 1
    for k, v in manifest["param_groups"][0].items():
        got = pg[0][k]
        assert (list(got) if isinstance(got, (tuple, list)) else got) == v, k
    st = ck["optimizer_state"]["state"]
    assert sorted(st.keys()) == list(range(80))
    for i, entries in manifest["opt_state"].items():
        for k, (shape, dtype) in entries.items():
            assert list(st[int(i)][k].shape) == shape and str(st[int(i)][k].dtype) == dtype
    assert {k: sorted(v.keys()) for k, v in ck["loss"].items()} == manifest["loss_keys"]
    # resume: construct + load_state reproduces the eval loss of the trained model bit for bit
    model.eval()
    x = torch.from_numpy(syn.spectrograms(B))
    ew, ed = syn.noise(B, z)
    model.noise_source = lambda b, zz: (ew, ed)
    with torch.no_grad():
        want = float(model.forward(x).item())
    m2 = VAE(save_dir=str(tmp_path), z_dim=z, device_name="cuda")
    m2.load_state(os.path.join(tmp_path, "checkpoint_001.tar"))   # load_state takes the path as given (vae.py:464)
    assert m2.epoch == 2 and m2.loss["train"].keys() == model.loss["train"].keys()
    m2.eval()
    m2.noise_source = model.noise_source
    with torch.no_grad():
        assert float(m2.forward(x).item()) == want
    # a resumed step equals a continued step (Adam state restored)
    model.train(); m2.train()
    for mm in (model, m2):
        mm.optimizer.zero_grad()
        mm.forward(x).backward()
        mm.optimizer.step()
    assert torch.equal(model._params, m2._params)
    with pytest.raises(AssertionError):
        VAE(z_dim=64, device_name="cuda").load_state(os.path.join(tmp_path, "checkpoint_001.tar"))


def test_visualize_writes_pdf(tmp_path):
    model = build_model(32)
    model.save_dir = str(tmp_path)
    loaders = syn.get_synthetic_data_loaders(12, batch_size=4)
    specs, rec = model.visualize(loaders["train"])
    assert specs.shape == rec.shape == (5, 128, 128)
    assert os.path.exists(os.path.join(tmp_path, "reconstruction.pdf"))
    with pytest.raises(AssertionError):
        model.visualize(loaders["train"], num_specs=13)


def test_invalid_posterior_raises_value_error():
    """d = exp(.) overflowing to inf/NaN -> the reference's distribution validation raises ValueError."""
    model = build_model(32)
    with torch.no_grad():
        model.fc43.bias.fill_(float("nan"))
    with pytest.raises(ValueError):
        model.forward(torch.from_numpy(syn.spectrograms(4)))


def test_full_batch_gradients_vs_oracle():
    """B = 256, z = 32 (the benchmark's configuration): the whole gradient arena against the CPU oracle's autograd
    backward on the same inputs.  fp32 ReLU-mask flips bound what two correct evaluations can agree on (the
    reference itself is 1e-3 away from an fp64 evaluation at B = 64): global relative L2 error
    < 5e-4 (measured 4.8e-5), every tensor < 2e-2 (measured <= 4.9e-3)."""
    B, z = 256, 32
    from ava_amd import layout
    x = torch.from_numpy(syn.spectrograms(B, salt=4242))
    ew, ed = syn.noise(B, z, 5, 6)
    model = build_model(z)
    model.noise_source = lambda b, zz: (ew, ed)
    loss = model.forward(x.cuda())
    loss.backward()
    g = model._grads.cpu().double()
    P = O.to_params(syn.fixture_parameters(z), requires_grad=True)
    out = O.forward(P, x, torch.from_numpy(ew), torch.from_numpy(ed), None, True)
    out["loss"].backward()
    offs, total = layout.arena_offsets(z)
    ref = torch.zeros(total, dtype=torch.float64)
    for s in param_specs(z):
        r = P[s.name].grad.reshape(-1).double()
        ref[offs[s.name]:offs[s.name] + s.numel] = r
        got = g[offs[s.name]:offs[s.name] + s.numel]
        assert float((got - r).norm() / max(float(r.norm()), 1e-30)) < 2e-2, s.name
    assert float((g - ref).norm() / ref.norm()) < 5e-4
    assert rel(float(loss.item()), float(out["loss"])) < 1e-5


@pytest.mark.parametrize("B,z", [(256, 32), (256, 64), (128, 32)], ids=["config2_B256_z32", "config3_B256_z64", "config4_B128_z32"])
def test_full_batch_properties(B, z):
    """BASELINE.json's full sizes (configs[1], [2] and the per-GPU batch of the strong-scaling reading of [3]):
    run-to-run bit determinism (no float atomics anywhere), the ELBO against the CPU oracle's forward on the same
    inputs (1e-5 relative; north-star tolerance 1e-4), the fused SSE / sum z^2 reductions against torch reductions
    of the kernels' own outputs, gradient finiteness, and step-to-step loss decrease under Adam."""
    x = torch.from_numpy(syn.spectrograms(B, salt=4242)).cuda()
    ew, ed = syn.noise(B, z, 5, 6)
    runs = []
    for _ in range(2):
        model = build_model(z)
        model.noise_source = lambda b, zz: (ew, ed)
        loss = model.forward(x)
        loss.backward()
        runs.append((float(loss.item()), model._grads.clone()))
    assert runs[0][0] == runs[1][0]
    assert torch.equal(runs[0][1], runs[1][1])
    assert bool(torch.isfinite(runs[0][1]).all())
    with torch.no_grad():
        P = O.to_params(syn.fixture_parameters(z))
        want = O.forward(P, x.cpu(), torch.from_numpy(ew), torch.from_numpy(ed), None, True)
    assert rel(runs[0][0], float(want["loss"])) < 1e-5
    xr = model._workspace_tensor("xrec", (B, 16384))
    zs = model._workspace_tensor("z", (B, z))
    lb = model._loss_buf.cpu().double().numpy()
    assert rel(lb[2], float(((x.view(B, -1).double() - xr.double()) ** 2).sum())) < 1e-6
    assert rel(lb[1], float((zs.double() ** 2).sum())) < 1e-6
    seed = model._workspace_tensor("seed", (B, 16384))
    assert rel(seed.cpu(), (10.0 * (xr - x.view(B, -1))).cpu()) < 1e-6
    losses = []
    for _ in range(5):
        model.optimizer.zero_grad()
        l = model.forward(x)
        l.backward()
        model.optimizer.step()
        losses.append(float(l.item()))
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("act", [None, "bfloat16"], ids=["fp32", "bf16"])
def test_config5_full_shard_properties(act):
    """BASELINE configs[4] at its REAL per-GPU size: 256 x 256 spectrograms, z = 128, 64 samples
    (batch 512 over 8 GPUs), fp32 and `act_dtype='bfloat16'` (bf16 activation storage + bf16 conv arithmetic).  The
    oracle comparisons of this configuration run at batch 4 (test_256x256_*, test_bf16_activation_storage); here the
    size-independent properties at full size: finite loss and gradients, run-to-run bit identity, the fused SSE
    reduction against a torch reduction of the kernel's own output, loss decrease under Adam, and the two modes' ELBO
    within 5e-3 of each other (the cost of bf16 arithmetic, stated in test_bf16_activation_storage)."""
    from ava_amd.vae import VAE
    shape, z, B = (256, 256), 128, 64
    fp = syn.fixture_parameters(z, shape)
    x = torch.from_numpy(syn.spectrograms(B, salt=909, shape=shape)).cuda()
    ew, ed = syn.noise(B, z, 7, 8)
    runs = []
    for _ in range(2):
        kw = {} if act is None else {"act_dtype": act}
        model = VAE(z_dim=z, device_name="cuda", x_shape=shape, **kw)
        with torch.no_grad():
            for name, prm in model.named_parameters():
                prm.copy_(torch.from_numpy(fp[name]))
        model.noise_source = lambda b, zz: (ew, ed)
        model.train()
        loss = model.forward(x)
        loss.backward()
        runs.append((float(loss.item()), model._grads.clone()))
    assert np.isfinite(runs[0][0]) and bool(torch.isfinite(runs[0][1]).all())
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1])
    xr = model._workspace_tensor("xrec", (B, shape[0] * shape[1]))
    lb = model._loss_buf.cpu().double().numpy()
    assert rel(lb[2], float(((x.view(B, -1).double() - xr.double()) ** 2).sum())) < 1e-6
    if act is not None:
        ref = VAE(z_dim=z, device_name="cuda", x_shape=shape)
        with torch.no_grad():
            for name, prm in ref.named_parameters():
                prm.copy_(torch.from_numpy(fp[name]))
        ref.noise_source = lambda b, zz: (ew, ed)
        ref.train()
        with torch.no_grad():
            assert rel(runs[0][0], float(ref.forward(x).item())) < 5e-3
        del ref
    losses = []
    for _ in range(4):
        model.optimizer.zero_grad()
        l = model.forward(x)
        l.backward()
        model.optimizer.step()
        losses.append(float(l.item()))
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("B", [8, 64, 256])
def test_gradients_against_fp64_noise_floor(B):
    """the gradient tolerance is derived from an fp64 evaluation of the oracle instead of being
    asserted -- err(HIP, fp64) must stay within the distance err(fp32 oracle, fp64) on the same fixture
    (_assert_within_noise_floor).  Measured (tools/grad_fp64.py, profiles/r02/grad_fp64.log): global relative L2 error
    HIP / fp32 oracle = 4.2e-7 / 4.1e-6 at B=8, 1.6e-4 / 8.0e-5 at B=64, 5.0e-5 / 5.0e-5 at B=256."""
    z = 32
    x = syn.spectrograms(B)
    ew, ed = syn.noise(B, z)
    model = build_model(z)
    model.noise_source = lambda b, zz: (ew, ed)
    loss = model.forward(torch.from_numpy(x))
    loss.backward()
    named = dict(model.named_parameters())
    got = {s.name: named[s.name].grad.cpu().double().numpy().ravel() for s in param_specs(z)}
    fp = syn.fixture_parameters(z)
    g64, l64 = _oracle_grads(fp, x, ew, ed, torch.float64)
    g32, _ = _oracle_grads(fp, x, ew, ed, torch.float32)
    assert rel(float(loss.item()), l64) < 1e-6           # fp32 ELBO vs fp64: measured 2e-8
    rep = {}
    _assert_within_noise_floor(got, g32, g64, z, rep)
    per, _ = _masked_fp64_grad_errors(model, fp, x, ew, ed, z)
    print("B=%d  HIP vs fp64: global %.2e max %.2e ; fp32 oracle vs fp64: global %.2e max %.2e ; HIP vs fp64 with the "
          "device's ReLU masks imposed: max %.2e" % (B, rep["hip_global"], rep["hip_max"], rep["o32_global"],
                                                      rep["o32_max"], max(per.values())))
    bad = {k: v for k, v in per.items() if v > 1e-4}
    assert not bad, bad


def test_flip_free_fixture_meets_appendix_b():
    """A fixture with NO ReLU pre-activation within 2e-5 (relative to its channel's rms) of zero (tests/flipfree.py:
    biases nudged by <= 2 % of the rms): no mask can flip, so whole-path gradients must meet SURVEY Appendix B's 1e-4
    against an fp64 evaluation on EVERY tensor (measured <= 2.2e-6) -- a real 1e-3 bug anywhere in the hand-derived
    backward (e.g. the BatchNorm-backward coefficient hand-off between layers) cannot hide behind flip noise here."""
    from flipfree import flipfree_parameters
    B, z = 8, 32
    x = syn.spectrograms(B)
    ew, ed = syn.noise(B, z)
    fp, min_rel = flipfree_parameters(syn.fixture_parameters(z), x, ew, ed)
    assert min_rel >= 1.9e-5
    model = build_model(z, fixture=False)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            prm.copy_(torch.from_numpy(fp[name]))
    model.noise_source = lambda b, zz: (ew, ed)
    loss = model.forward(torch.from_numpy(x))
    loss.backward()
    named = dict(model.named_parameters())
    got = {s.name: named[s.name].grad.cpu().double().numpy().ravel() for s in param_specs(z)}
    g64, l64 = _oracle_grads(fp, x, ew, ed, torch.float64)
    assert rel(float(loss.item()), l64) < 1e-6
    eh, gh = _errors_vs_fp64(got, g64, z)
    bad = {k: v for k, v in eh.items() if v > 1e-4}
    assert not bad, bad
    assert gh < 1e-5, gh


@pytest.mark.parametrize("B", [8, 64])
def test_flip_free_reference_golden(B):
    """the HIP path against gradients the REAL reference produced on the flip-free fixture
    (tests/golden/flipfree_B*_z32.npz, generated by tests/golden/make_golden.py from /root/reference): all 80 tensors within
    1e-4, the ELBO and its three sums within 1e-5, with NO masks imposed and NO flip allowance -- on this fixture no ReLU mask
    can differ between two correct fp32 evaluations, so this is what decides whether an arithmetic (limb set, tiling,
    reduction order) is admissible; the flip-sensitive goldens above no longer select kernels."""
    from test_oracle_golden import flipfree_fixture, assert_flipfree_gradients
    z = 32
    G, fp, x, ew, ed = flipfree_fixture(B, z)
    model = build_model(z, fixture=False)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            prm.copy_(torch.from_numpy(fp[name]))
    model.noise_source = lambda b, zz: (ew, ed)
    loss = model.forward(torch.from_numpy(x))
    lb = model._loss_buf.cpu().numpy()
    loss.backward()
    assert rel(float(loss.item()), G["loss"]) < 1e-5
    assert rel(lb[1], G["sum_z2"]) < 1e-5 and rel(lb[2], G["sse"]) < 1e-5 and rel(lb[3], G["sum_h"]) < 1e-5
    named = dict(model.named_parameters())
    got = {s.name: named[s.name].grad.cpu().numpy() for s in param_specs(z)}
    worst = assert_flipfree_gradients(got, G, z)
    print("B=%d: HIP vs reference on the flip-free fixture: loss %.1e, worst tensor %.2e" %
          (B, rel(float(loss.item()), G["loss"]), worst))


@pytest.mark.parametrize("shape,B,z", [((256, 256), 4, 128), ((128, 256), 3, 32), ((256, 128), 5, 64)],
                         ids=["config5_256x256_z128", "128x256", "256x128"])
def test_size_extension_forward_backward_vs_oracle(shape, B, z):
    """BASELINE configs[4]'s geometry (256 x 256 spectrograms, z = 128) and the two non-square sizes, in fp32: the
    reference has no 256 x 256 path (X_SHAPE and the literal 8192 are module constants, vae.py:33,142), so parity is
    against the oracle -- pinned to the reference at 128 x 128 -- with fc1.in = fc8.out = 32*H/8*W/8.  ELBO and its
    three sums 1e-5 against the fp32 oracle; every gradient tensor 1e-4 against the fp64 oracle with the device's ReLU
    masks imposed; one Adam step; decode / encode shapes."""
    from ava_amd.vae import VAE
    H, W = shape
    fp = syn.fixture_parameters(z, shape)
    model = VAE(z_dim=z, device_name="cuda", x_shape=shape)
    with torch.no_grad():
        for name, prm in model.named_parameters():
            prm.copy_(torch.from_numpy(fp[name]))
    ew, ed = syn.noise(B, z, 21, 22)
    model.noise_source = lambda b, zz: (ew, ed)
    x = torch.from_numpy(syn.spectrograms(B, salt=55, shape=shape))
    model.train()
    loss = model.forward(x)
    lb = model._loss_buf.cpu().numpy()
    with torch.no_grad():
        want = O.forward(O.to_params(fp), x, torch.from_numpy(ew), torch.from_numpy(ed), None, True)
    assert rel(float(loss.item()), float(want["loss"])) < 1e-5
    assert rel(lb[1], float(want["sum_z2"])) < 1e-5 and rel(lb[2], float(want["sse"])) < 1e-5 and rel(lb[3], float(want["sum_h"])) < 1e-5
    xr = model._workspace_tensor("xrec", (B, H * W)).cpu()
    assert rel(xr, want["x_rec"]) < 1e-4
    loss.backward()
    named = dict(model.named_parameters())
    got = {n: p.grad.detach().cpu().double().numpy().ravel() for n, p in named.items()}
    masks = _hip_masks(model, B)
    P = O.to_params(fp, dtype=torch.float64, requires_grad=True)
    out = O.forward(P, x.double(), torch.from_numpy(ew).double(), torch.from_numpy(ed).double(), None, True, masks=masks)
    out["loss"].backward()
    cb = float(P["conv1.bias"].grad.norm())
    bad = {}
    for n, p in P.items():
        r = p.grad.numpy().ravel()
        scale = max(np.linalg.norm(r), cb if n.split(".")[0] in ("conv1", "bn1") else 0.0, 1e-300)
        e = np.linalg.norm(got[n] - r) / scale
        if e > 1e-4:
            bad[n] = e
    assert not bad, bad
    before = model._params.clone()
    model.optimizer.step()
    assert float((model._params - before).abs().max()) > 5e-4           # Adam's first step ~ lr on every weight
    with torch.no_grad():
        mu, u, d = model.encode(x)
        rec = model.decode(torch.zeros(B, z))
    assert mu.shape == (B, z) and u.shape == (B, z, 1) and rec.shape == (B, H * W)
    with pytest.raises(AssertionError):
        model.forward(torch.zeros(B, 128, 128) if shape != (128, 128) else torch.zeros(B, 64, 64))


def _hip_stored(model, B):
    """the activations the device stored between the conv layers, by oracle layer name, NCHW float64"""
    H, W = model.x_shape
    ch = {"conv1": (8, 1), "conv2": (8, 2), "conv3": (16, 2), "conv4": (16, 4), "conv5": (24, 4), "conv6": (24, 8),
          "convt1": (24, 8), "convt2": (24, 4), "convt3": (16, 4), "convt4": (16, 2), "convt5": (8, 2), "convt6": (8, 1)}
    st = {}
    for i in range(1, 7):
        c, d = ch["conv%d" % i]
        st["conv%d" % i] = model._workspace_tensor("y%d" % i, (B, H // d, W // d, c)).permute(0, 3, 1, 2).double().cpu()
        c, d = ch["convt%d" % i]
        st["convt%d" % i] = model._workspace_tensor("d%d" % i, (B, H // d, W // d, c)).permute(0, 3, 1, 2).double().cpu()
    st["fc8"] = model._workspace_tensor("f8t", (B, H // 8, W // 8, 32)).permute(0, 3, 1, 2).double().cpu()
    return st


@pytest.mark.parametrize("shape,B,z", [((128, 128), 8, 32), ((256, 256), 4, 128)], ids=["128x128", "config5_256x256_z128"])
def test_bf16_activation_storage(shape, B, z):
    """BASELINE configs[4] "bf16 conv + fp32 ELBO": VAE(act_dtype='bfloat16') stores the thirteen activation tensors
    between the conv layers as bfloat16 (rounded to nearest even by the producing kernel) AND -- round 5 -- computes the
    twelve convolutions with >= 8 channels on both sides in bf16 ARITHMETIC: weights and BatchNorm outputs rounded to
    bfloat16, one-limb products on v_mfma_f32_16x16x32_bf16 (three-limb fp32 gradients in the backward), fp32 accumulation;
    conv1 / convt7, BatchNorm statistics, gradients, fully connected layers, ELBO and Adam stay fp32
    (oracle.BF16_MATH_LAYERS, oracle._conv_operands).

    Tolerances (stated):
    * tight: against the fp64 oracle evaluated ON THE TENSORS THE DEVICE STORED (oracle._store(stored=...): value =
      the device's bf16 tensor, straight-through gradient) with the device's ReLU masks and the SAME operand rounding
      (bf16_math=True: the rounded weights and rounded BatchNorm outputs enter the products, straight-through) -- the
      exact derivative of the function the device evaluated: -ELBO 1e-6 relative, every gradient tensor 1e-4 (what is
      left besides fp32 rounding: a BatchNorm output within ~1e-7 of a bfloat16 rounding boundary may round the other
      way in the oracle's fp64 arithmetic, one bf16 ulp on ~5e-5 of the elements);
    * the stored tensors themselves: bfloat16, equal to round-to-nearest-even of the fp32 oracle's first activation
      on > 99.8 % of the elements (the rest sit on a rounding boundary within fp32 noise);
    * what the mode costs against the plain fp32 oracle: -ELBO within 5e-3 relative (measured 2e-5 with storage alone,
      ~6e-4 with bf16 arithmetic); gradients are
      NOT comparable tensor by tensor -- any two evaluations of a bf16-rounded network decorrelate to one bf16 ulp
      (0.4 %) per element within a few layers and then disagree on ~0.4 % of the ReLU masks (5-25 % per gradient
      tensor, the same between an fp32 and an fp64 run of the rounded oracle: tools/bf16_probe.py) -- so only the
      direction is held (cosine > 0.9 on every weight tensor) plus loss decrease under Adam."""
    from ava_amd.vae import VAE
    H, W = shape
    fp = syn.fixture_parameters(z, shape)
    model = VAE(z_dim=z, device_name="cuda", x_shape=shape, act_dtype="bfloat16")
    with torch.no_grad():
        for name, prm in model.named_parameters():
            prm.copy_(torch.from_numpy(fp[name]))
    ew, ed = syn.noise(B, z, 21, 22)
    model.noise_source = lambda b, zz: (ew, ed)
    x = torch.from_numpy(syn.spectrograms(B, salt=55, shape=shape))
    model.train()
    model.optimizer.zero_grad()
    loss = model.forward(x)
    loss.backward()
    got = {n: p.grad.detach().cpu().double().numpy().ravel() for n, p in model.named_parameters()}
    y1 = model._workspace_tensor("y1", (B, H, W, 8))
    assert y1.dtype == torch.bfloat16
    # ---- tight: fp64 oracle on the device's stored tensors and masks ----
    P = O.to_params(fp, dtype=torch.float64, requires_grad=True)
    out = O.forward(P, x.double(), torch.from_numpy(ew).double(), torch.from_numpy(ed).double(), None, True,
                    masks=_hip_masks(model, B), stored=_hip_stored(model, B), bf16_math=True)
    out["loss"].backward()
    assert rel(float(loss.item()), float(out["loss"].detach())) < 1e-6
    cb = float(P["conv1.bias"].grad.norm())
    bad = {}
    for n, p in P.items():
        r = p.grad.numpy().ravel()
        scale = max(np.linalg.norm(r), cb if n.split(".")[0] in ("conv1", "bn1") else 0.0, 1e-300)
        e = np.linalg.norm(got[n] - r) / scale
        if e > 1e-4:
            bad[n] = e
    assert not bad, bad
    # ---- the stored tensor is the rounded fp32 activation ----
    P32 = O.to_params(fp, requires_grad=True)
    rec = {}
    o32 = O.forward(P32, x, torch.from_numpy(ew), torch.from_numpy(ed), None, True, record=rec)
    o32["loss"].backward()
    want_y1 = rec["conv1.out"].detach().permute(0, 2, 3, 1).bfloat16()
    assert float((y1.cpu() != want_y1).float().mean()) < 2e-3
    # ---- cost of the mode against plain fp32 ----
    assert rel(float(loss.item()), float(o32["loss"].detach())) < 5e-3
    for n, p in P32.items():
        if n.endswith(".weight") and not n.startswith("bn"):
            r = p.grad.double().numpy().ravel()
            cos = float(np.dot(got[n], r) / max(np.linalg.norm(got[n]) * np.linalg.norm(r), 1e-300))
            assert cos > 0.9, (n, cos)
    # ---- bitwise determinism, a few Adam steps, encode ----
    g1 = model._grads.clone()
    model.optimizer.zero_grad()
    model.forward(x).backward()
    assert torch.equal(model._grads, g1)
    losses = []
    for _ in range(4):
        model.optimizer.zero_grad()
        l = model.forward(x)
        l.backward()
        model.optimizer.step()
        losses.append(float(l.item()))
    assert losses[-1] < losses[0]
    with torch.no_grad():
        mu, _, _ = model.encode(x)
        assert mu.shape == (B, z) and bool(torch.isfinite(mu).all())
    with pytest.raises(ValueError):
        VAE(device_name="cuda", act_dtype="float16")
