"""Persistent grids beside another kernel's persistent workgroups (data parallelism: RCCL's channels).

Every conv / large-GEMM launch is one resident wave of workgroups over a STATIC tile partition, so what a step computes
cannot depend on who else holds wave slots meanwhile -- only its duration can.  `ava_set_cu_reserve(r)` sizes the grids
for 256 - r CUs; `ava_occupy_cus` stands in for the collective (workgroups that only hold their slots on a side stream).
The gpurun boxes have one GPU: this is the evidence the design offers for the RCCL case."""
import ctypes

import numpy as np
import pytest
import torch

from ava_amd import _lib, synthetic as syn
from gpu_util import build_model

pytestmark = pytest.mark.gpu


def _step(model, x, thief_wgs=0):
    lib = _lib.load()
    model.optimizer.zero_grad()
    side = torch.cuda.Stream()
    loss = model.forward(x)
    torch.cuda.synchronize()
    if thief_wgs:
        # 100 KB of LDS each: at most one per CU, and a CU that hosts one has room for one conv workgroup less
        _lib.check(lib.ava_occupy_cus(thief_wgs, 100 * 1024, 3000.0, ctypes.c_void_p(side.cuda_stream)), "occupy")
    loss.backward()
    torch.cuda.synchronize()
    return float(loss.item()), model._grads.clone()


def test_results_do_not_depend_on_a_co_resident_kernel_and_reserve_is_reproducible():
    lib = _lib.load()
    z, B = 32, 64
    x = torch.from_numpy(syn.spectrograms(B)).cuda()
    ew, ed = syn.noise(B, z)
    assert lib.ava_get_cu_reserve() == 0
    try:
        out = {}
        for reserve in (0, 32):
            _lib.check(lib.ava_set_cu_reserve(reserve), "reserve")
            model = build_model(z)
            model.noise_source = lambda b, zz: (ew, ed)
            l0, g0 = _step(model, x)
            l1, g1 = _step(model, x, thief_wgs=32)          # slots stolen during the backward
            l2, g2 = _step(model, x)
            assert l0 == l1 == l2
            assert torch.equal(g0, g1) and torch.equal(g0, g2)     # bit-identical with and without the thief
            out[reserve] = (l0, g0)
        # a different reserve is a different (equally valid) partition of the same sums: fp32 rounding apart
        (la, ga), (lb, gb) = out[0], out[32]
        assert abs(la - lb) / abs(la) < 1e-6
        # the same reserve set PER MODEL (ava_model_set_cu_reserve: what dist.py uses) with the process-wide value at 0: the
        # model's entry points snapshot it at their start -- bit-identical to the process-wide 32 above; a second model in the
        # same process keeps its own setting
        _lib.check(lib.ava_set_cu_reserve(0), "reserve")
        model = build_model(z)
        model.noise_source = lambda b, zz: (ew, ed)
        model._ensure(B)
        other = build_model(z)
        other.noise_source = lambda b, zz: (ew, ed)
        other._ensure(B)
        _lib.check(lib.ava_model_set_cu_reserve(model._handle, 32), "model reserve")
        assert lib.ava_model_get_cu_reserve(model._handle) == 32 and lib.ava_model_get_cu_reserve(other._handle) == -1
        lm, gm = _step(model, x)
        lo, go = _step(other, x)
        assert lm == lb and torch.equal(gm, gb)
        assert lo == la and torch.equal(go, ga)
        assert lib.ava_model_set_cu_reserve(model._handle, 129) != 0 and lib.ava_model_set_cu_reserve(model._handle, -2) != 0
        num = float((ga.double() - gb.double()).norm())
        assert num / float(ga.double().norm()) < 1e-4       # fp32 rounding of the partial sums + the odd ReLU-mask flip
    finally:
        lib.ava_set_cu_reserve(0)


def test_reserve_argument_checks():
    lib = _lib.load()
    assert lib.ava_set_cu_reserve(-1) != 0 and lib.ava_set_cu_reserve(129) != 0
    assert lib.ava_occupy_cus(0, 0, 10.0, None) != 0 and lib.ava_occupy_cus(8, 0, 1e9, None) != 0
    assert lib.ava_get_cu_reserve() == 0
