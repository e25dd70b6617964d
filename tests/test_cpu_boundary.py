"""CPU-side tests (no GPU needed): the C-ABI library loads and exports everything include/ava_hip.h
declares, the arena layout mirror agrees with the library, the VAE class surface / checkpoint layout
matches the reference manifest, and the compute path refuses to run without the GPU."""
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT
from ava_amd import _lib, layout, synthetic as syn
from ava_amd.vae import VAE, X_SHAPE, X_DIM


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "ava_hip.h")).read()
    declared = set(re.findall(r"\b(ava_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 30
    lib = _lib.load()
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ava_version() >= 100


@pytest.mark.parametrize("z", [8, 32, 64, 100])
def test_arena_layout_mirror(z):
    lib = _lib.load()
    offs, total = layout.arena_offsets(z)
    assert total == lib.ava_arena_floats(z)
    import ctypes
    for s in layout.param_specs(z):
        n = ctypes.c_int64()
        assert offs[s.name] == lib.ava_param_offset(z, s.index, ctypes.byref(n)) and n.value == s.numel
        assert offs[s.name] % 64 == 0
    # the three 256->64 heads form one [192,256] matrix and one [192] bias
    assert offs["fc32.weight"] == offs["fc31.weight"] + 64 * 256 and offs["fc33.weight"] == offs["fc32.weight"] + 64 * 256
    assert offs["fc32.bias"] == offs["fc31.bias"] + 64 and offs["fc33.bias"] == offs["fc32.bias"] + 64
    assert lib.ava_workspace_bytes(z, 256) > lib.ava_workspace_bytes(z, 8) > 0


def test_constants_and_parameter_order():
    assert X_SHAPE == (128, 128) and X_DIM == 16384
    m = VAE(device_name="cpu")
    names = [n for n, _ in m.named_parameters()]
    assert names == [s.name for s in layout.param_specs(32)]
    assert sum(p.numel() for p in m.parameters()) == 17426323 == layout.num_params(32)
    assert sum(p.numel() for p in VAE(z_dim=64, device_name="cpu").parameters()) == 17434611
    for s in layout.param_specs(32):
        assert tuple(dict(m.named_parameters())[s.name].shape) == s.shape
    assert list(m._get_layers().keys()) == layout.checkpoint_layer_order()
    for attr in ("save_dir", "lr", "z_dim", "model_precision", "device", "optimizer", "epoch", "loss"):
        assert hasattr(m, attr)
    assert m.loss == {"train": {}, "test": {}} and m.epoch == 0 and m.training


def test_parameters_are_views_of_one_arena():
    m = VAE(device_name="cpu")
    base = m._params.data_ptr()
    for name, p in m.named_parameters():
        o, n, shape = m._arena_views[name]
        assert p.data_ptr() == base + 4 * o and p.is_contiguous()
        assert p.grad is not None and p.grad.data_ptr() == m._grads.data_ptr() + 4 * o
    with torch.no_grad():
        m.fc1.weight.fill_(3.0)
    o, n, _ = m._arena_views["fc1.weight"]
    assert float(m._params[o:o + n].min()) == 3.0
    # default init statistics are torch's (kaiming-uniform(a=sqrt 5): bound 1/sqrt(fan_in))
    m2 = VAE(device_name="cpu")
    assert float(m2.fc2.weight.abs().max()) <= 1 / np.sqrt(1024) + 1e-6
    assert float(m2.bn3.weight.min()) == 1.0 and float(m2.bn3.bias.abs().max()) == 0.0
    assert float(m2.bn3.running_var.min()) == 1.0 and int(m2.bn3.num_batches_tracked) == 0


def test_checkpoint_layout_matches_reference_manifest(tmp_path):
    manifest = json.load(open(os.path.join(GOLDEN, "checkpoint_manifest.json")))
    m = VAE(save_dir=str(tmp_path), device_name="cpu")
    m.save_state("ck.tar")
    ck = torch.load(os.path.join(tmp_path, "ck.tar"), weights_only=True)
    assert list(ck.keys()) == manifest["keys"]
    for name, entries in manifest["layers"].items():
        assert list(ck[name].keys()) == manifest["layer_key_order"][name]
        for k, (shape, dtype) in entries.items():
            assert list(ck[name][k].shape) == shape and str(ck[name][k].dtype) == dtype
    assert ck["optimizer_state"]["state"] == {}          # empty before the first step, like torch's Adam
    for k, v in manifest["param_groups"][0].items():
        got = ck["optimizer_state"]["param_groups"][0][k]
        assert (list(got) if isinstance(got, (tuple, list)) else got) == v, k
    # round trip
    m2 = VAE(save_dir=str(tmp_path), device_name="cpu")
    m2.load_state(os.path.join(tmp_path, "ck.tar"))     # load_state does not prepend save_dir (vae.py:464)
    assert torch.equal(m._params, m2._params)
    with pytest.raises(AssertionError):
        VAE(save_dir=str(tmp_path), z_dim=16, device_name="cpu").load_state(os.path.join(tmp_path, "ck.tar"))


def test_loads_optimizer_state_written_by_torch_adam(tmp_path):
    """A checkpoint whose optimizer_state comes from a stock torch.optim.Adam (what the reference writes)."""
    m = VAE(save_dir=str(tmp_path), device_name="cpu")
    ref_opt = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in m.parameters()], lr=1e-3)
    for p in ref_opt.param_groups[0]["params"]:
        p.grad = torch.full_like(p, 0.5)
    ref_opt.step(); ref_opt.step()
    m.optimizer.load_state_dict(ref_opt.state_dict())
    assert m.optimizer._step_count_flat == 2
    o, n, _ = m._arena_views["fc2.bias"]
    assert torch.allclose(m._exp_avg[o:o + n], ref_opt.state[ref_opt.param_groups[0]["params"][31]]["exp_avg"])
    sd = m.optimizer.state_dict()
    assert sorted(sd["state"].keys()) == list(range(80))
    assert sd["state"][0]["step"].dtype == torch.float32 and float(sd["state"][0]["step"]) == 2.0
    assert sd["state"][28]["exp_avg"].shape == (1024, 8192)


def test_no_cpu_fallback():
    m = VAE(device_name="cpu")
    x = torch.from_numpy(syn.spectrograms(2))
    with pytest.raises(_lib.AvaHipError):
        m.forward(x)
    with pytest.raises(_lib.AvaHipError):
        m.encode(x)
    with pytest.raises(AssertionError):
        if not torch.cuda.is_available():
            VAE(device_name="cuda")
        else:
            raise AssertionError


def test_synthetic_loader_contract():
    loaders = syn.get_synthetic_data_loaders(10, 4, batch_size=4, shuffle=(False, False))
    ds = loaders["train"].dataset
    assert len(ds) == 10 and len(loaders["test"].dataset) == 4
    item = ds[3]
    assert item.dtype == torch.float32 and tuple(item.shape) == X_SHAPE
    several = ds[np.array([1, 3])]
    assert isinstance(several, list) and torch.equal(several[1], item)
    batches = list(loaders["train"])
    assert [tuple(b.shape) for b in batches] == [(4, 128, 128), (4, 128, 128), (2, 128, 128)]
    assert float(batches[0].min()) == 0.0 and float(batches[0].max()) <= 1.0
    assert 0.2 < float((batches[0] == 0).float().mean()) < 0.4          # ~29 % exact zeros
    assert syn.get_synthetic_data_loaders(5)["test"] is None
    # the integer-hash recipe is platform independent: pin a few values
    np.testing.assert_allclose(syn.u01(3, 1001), [0.17569250689332405, 0.46630860756399706, 0.06287327795656672], rtol=0, atol=1e-15)
    assert torch.equal(torch.from_numpy(syn.spectrograms(3)[2]), ds[2])


@pytest.mark.parametrize("shape", [(256, 256), (128, 256), (256, 128)])
def test_size_extension_layout_mirror_and_surface(shape):
    """x_shape other than the reference's (128, 128) (BASELINE config 5: 256 x 256): fc1.in = fc8.out = 32*H/8*W/8, the
    arena mirror agrees with the library, everything else keeps the reference's shapes."""
    import ctypes
    lib = _lib.load()
    z = 128
    H, W = shape
    f = 32 * (H // 8) * (W // 8)
    assert layout.bottleneck_features(shape) == f
    offs, total = layout.arena_offsets(z, x_shape=shape)
    assert total == lib.ava_arena_floats_hw(z, H, W)
    for s in layout.param_specs(z, shape):
        n = ctypes.c_int64()
        assert offs[s.name] == lib.ava_param_offset_hw(z, H, W, s.index, ctypes.byref(n)) and n.value == s.numel
    assert lib.ava_workspace_bytes_hw(z, H, W, 8) > lib.ava_workspace_bytes(z, 8) > 0
    m = VAE(z_dim=z, device_name="cpu", x_shape=shape)
    assert tuple(m.fc1.weight.shape) == (1024, f) and tuple(m.fc8.weight.shape) == (f, 1024)
    assert m.x_shape == shape and m.x_dim == H * W
    assert [n for n, _ in m.named_parameters()] == [s.name for s in layout.param_specs(z, shape)]
    assert tuple(m.conv1.weight.shape) == (8, 1, 3, 3) and tuple(m.convt7.weight.shape) == (8, 1, 3, 3)


def test_unsupported_sizes_are_refused():
    lib = _lib.load()
    for bad in [(128, 192), (96, 128), (64, 128), (128, 512), (2048, 128), (384, 128), (512, 256)]:
        with pytest.raises(ValueError):
            VAE(device_name="cpu", x_shape=bad)
        assert lib.ava_arena_floats_hw(32, bad[0], bad[1]) == -1 and lib.ava_workspace_bytes_hw(32, bad[0], bad[1], 8) == 0
    assert lib.ava_arena_floats_hw(32, 128, 128) == lib.ava_arena_floats(32)


def test_no_packed_fma_selects_a_high_register_for_its_low_lane():
    """profiles/NOTES.md item 44: on gfx950 `v_pk_fma_f32 ... op_sel:[0,1,0]` (src1 read from the odd register of a pair by both
    lanes) returns a wrong low half in lanes 48..63 while a bf16-MFMA wave of another kernel shares the SIMD.  The two kernels
    in which hipcc produces that form pin their broadcast scalars (`conv_thin_kernels.h: ava_pin`); this scans the BUILT code
    objects so that a later edit -- or another operand order chosen by the compiler -- cannot silently bring the form back.  Skipped where the objects or the llvm tools are not at hand."""
    import glob
    import subprocess
    import sys
    import pytest
    objs = glob.glob(os.path.join(ROOT, "autoencoded-vocal-analysis_amd", "csrc", "*.o"))
    if not objs or not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"):
        pytest.skip("no built objects / llvm-objdump")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "lab", "op_sel_scan.py")], capture_output=True, text=True,
                         timeout=600)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "packed fp32 instructions scanned, 0 with" in res.stdout, res.stdout[-500:]


def test_op_sel_scanner_flags_exactly_the_form_the_probe_found_wrong():
    """the scanner's rule on disassembly lines of every form tools/lab/op_sel_forms.hip measured (profiles/NOTES.md item 44 h)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("op_sel_scan", os.path.join(ROOT, "tools", "lab", "op_sel_scan.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad = ["\tv_pk_fma_f32 v[86:87], v[142:143], v[46:47], v[86:87] op_sel:[0,1,0]// 0000: D3B0",
           "\tv_pk_mul_f32 v[2:3], v[4:5], v[6:7] op_sel:[0,1]",
           "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,1,0]",
           "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[0,1]"]                      # (add: not probed on src1, flagged conservatively)
    good = ["\tv_pk_fma_f32 v[4:5], v[24:25], v[24:25], v[4:5]",
            "\tv_pk_fma_f32 v[8:9], v[6:7], v[46:47], v[8:9] op_sel:[1,0,0]",
            "\tv_pk_fma_f32 v[8:9], v[6:7], v[46:47], v[8:9] op_sel:[0,0,1]",
            "\tv_pk_fma_f32 v[126:127], v[142:143], v[186:187], v[126:127] op_sel_hi:[1,0,1]",
            "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel_hi:[0,1,1]",
            "\tv_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[1,1,0] op_sel_hi:[0,0,1]",
            "\tv_pk_add_f32 v[0:1], v[2:3], v[4:5] op_sel:[1,0]",
            "\tv_pk_mul_lo_u16 v4, v4, 60 op_sel_hi:[1,0]",
            "\tv_fma_f32 v0, v1, v2, v3"]
    assert all(mod.is_bad(l) for l in bad) and not any(mod.is_bad(l) for l in good)
