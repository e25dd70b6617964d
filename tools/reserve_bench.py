#!/usr/bin/env python3
"""Step time with wave slots stolen by a co-resident persistent kernel (stand-in for RCCL's channels), for grids sized
for the whole chip (reserve 0) and for 256 - r CUs.  One GPU; prints a table for DESIGN.md section 4."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ava_amd import _lib, synthetic as syn
from gpu_util import build_model

lib = _lib.load()
B, z = 256, 32
x = torch.from_numpy(syn.spectrograms(B)).cuda()
side = torch.cuda.Stream()


def run(model, thief, steps=30):
    def one():
        model.optimizer.zero_grad()
        model._forward_device(x, need_grad=True)
        model._backward_device(x)
        model.optimizer.step()
    for _ in range(5):
        one()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if thief:   # holds its slots for the whole timed region (the all-reduces of a step cover most of its backward)
        _lib.check(lib.ava_occupy_cus(thief, 100 * 1024, min(19000.0, steps * 2600.0), ctypes.c_void_p(side.cuda_stream)), "occupy")
    e0.record()
    for _ in range(steps):
        one()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


print("reserve  thief_wgs  ms/step")
for reserve in (0, 16, 32, 64):
    lib.ava_set_cu_reserve(reserve)
    model = build_model(z)
    model.train()
    for thief in (0, 16, 32, 64):
        print("%7d  %9d  %.4f" % (reserve, thief, run(model, thief, steps=6 if thief else 30)), flush=True)
lib.ava_set_cu_reserve(0)
