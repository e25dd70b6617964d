#!/usr/bin/env python3
"""Do a weight-gradient kernel and the next layer's data-gradient kernel overlap when issued on two streams?"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import LAYERS, MODE_S1, MODE_DOWN, MODE_UP, PRO_BWD, EPI_BWD, p, out_size
from ava_amd import _lib
lib = _lib.load()
B = 256
L = {l[0]: l for l in LAYERS}

def mk(name):
    _, cin, cout, mode, hi, tr = L[name]
    ho = out_size(hi, mode)
    d = dict(cin=cin, cout=cout, mode=mode, hi=hi, ho=ho)
    d["x"] = torch.rand(B, hi, hi, cin, device="cuda"); d["y"] = torch.rand(B, ho, ho, cout, device="cuda")
    d["g"] = torch.randn(B, ho, ho, cout, device="cuda"); d["dx"] = torch.empty(B, hi, hi, cin, device="cuda")
    d["coef"] = torch.rand(3, 32, device="cuda"); d["G"] = torch.randn(9 * cin * cout, device="cuda") * 0.1
    d["bnp"] = torch.zeros(1024, 64, device="cuda"); d["wp"] = torch.zeros(512, 9 * cin * cout + cout, device="cuda")
    d["bmode"] = MODE_S1 if mode == MODE_S1 else (MODE_UP if mode == MODE_DOWN else MODE_DOWN)
    return d

def bwd(d, s):
    c = d["coef"]
    return lib.ava_conv3x3(p(d["g"]), p(d["y"]), p(c[0]), p(c[1]), p(c[2]), p(d["G"]), None, p(d["dx"]), None, p(d["x"]), p(c[1]), p(c[2]),
                           p(d["bnp"]), B, d["ho"], d["ho"], d["cout"], d["cin"], d["bmode"], PRO_BWD, EPI_BWD, 0, 0.0, ctypes.c_void_p(s.cuda_stream))
def wg(d, s):
    c = d["coef"]
    return lib.ava_conv3x3_wgrad(p(d["x"]), p(c[0]), p(c[1]), p(d["g"]), p(d["y"]), p(c[0]), p(c[1]), p(c[2]), p(d["wp"]), B, d["hi"], d["hi"],
                                 d["cin"], d["cout"], d["mode"], PRO_BWD, ctypes.c_void_p(s.cuda_stream))

s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for a, b in (("conv6", "conv5"), ("conv7", "conv6"), ("convt1", "convt2"), ("convt2", "convt3")):
    da, db = mk(a), mk(b)
    for _ in range(3): assert wg(da, s1) == 0 and bwd(db, s1) == 0
    torch.cuda.synchronize()
    def run(conc, n=20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(s1)
        for _ in range(n):
            if conc:
                ev = torch.cuda.Event(); ev.record(s1); s2.wait_event(ev)
                wg(da, s2); bwd(db, s1)
                if conc == 2:
                    ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
            else:
                wg(da, s1); bwd(db, s1)
        if conc == 1:
            ev2 = torch.cuda.Event(); ev2.record(s2); s1.wait_event(ev2)
        e1.record(s1); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / n
    print("wgrad(%s) + bwd(%s): serial %.1f us, fork per pair + join at end %.1f us, fork+join per pair %.1f us" % (a, b, run(0), run(1), run(2)), flush=True)
