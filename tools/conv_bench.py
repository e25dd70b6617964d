#!/usr/bin/env python3
"""Time every conv-family kernel of one train step (B=256) through the C ABI; prints us, algorithmic GB/s
(SURVEY 8d accounting) and useful TFLOP/s per kernel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import LAYERS, MODE_S1, MODE_DOWN, MODE_UP, PRO_BN, PRO_BWD, PRO_ID, EPI_FWD, EPI_BWD, EPI_SSE, p, stream, out_size
from ava_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
only = sys.argv[2] if len(sys.argv) > 2 else None
lib = _lib.load()


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot = {"fwd": 0.0, "bwd": 0.0, "wgrad": 0.0}
for name, cin, cout, mode, hi, tr in LAYERS:
    if only and only != name: continue
    ho = out_size(hi, mode)
    x = torch.rand(B, hi, hi, cin, device="cuda")
    y = torch.rand(B, ho, ho, cout, device="cuda")
    g = torch.randn(B, ho, ho, cout, device="cuda")
    out = torch.empty(B, ho, ho, cout, device="cuda")
    dx = torch.empty(B, hi, hi, cin, device="cuda")
    seed = torch.empty(B, ho, ho, cout, device="cuda")
    coef = torch.rand(3, 32, device="cuda")
    G = torch.randn(9 * cin * cout, device="cuda") * 0.1
    bias = torch.randn(32, device="cuda")
    grid = lib.ava_conv_grid(B, ho, ho, mode)
    parts = torch.zeros(1024, 64, device="cuda")
    bmode = MODE_S1 if mode == MODE_S1 else (MODE_UP if mode == MODE_DOWN else MODE_DOWN)
    wgrid = lib.ava_conv_wgrad_grid(B, ho, ho, mode)
    wparts = torch.zeros(wgrid, 9 * cin * cout + cout, device="cuda")
    last = name == "convt7"
    def fwd():
        return lib.ava_conv3x3(p(x), None, p(coef[0]), p(coef[1]), None, p(G), p(bias), p(out), p(seed) if last else None,
                               p(y) if last else None, None, None, p(parts), B, hi, hi, cin, cout, mode, PRO_BN,
                               EPI_SSE if last else EPI_FWD, 1, 10.0, stream())
    def bwd():
        return lib.ava_conv3x3(p(g), p(y), p(coef[0]), p(coef[1]), p(coef[2]), p(G), None, p(dx), None, p(x), p(coef[1]),
                               p(coef[2]), p(parts), B, ho, ho, cout, cin, bmode, PRO_BWD, EPI_BWD, 0, 0.0, stream())
    def wg():
        return lib.ava_conv3x3_wgrad(p(x), p(coef[0]), p(coef[1]), p(g), p(y), p(coef[0]), p(coef[1]), p(coef[2]),
                                     p(wparts), B, hi, hi, cin, cout, mode, PRO_BWD, stream())
    assert fwd() == 0 and bwd() == 0 and wg() == 0
    nin, nout = B * hi * hi * cin * 4, B * ho * ho * cout * 4
    flops = 2.0 * B * 9 * cin * cout * (ho * ho if mode != MODE_UP else hi * hi)
    for kind, fn, bytes_ in (("fwd", fwd, nin + nout), ("bwd", bwd, 2 * nout + 2 * nin), ("wgrad", wg, nin + 2 * nout)):
        us = timeit(fn)
        tot[kind] += us
        print("%-7s %-5s %2d->%2d @%3d  %7.1f us  %6.0f GB/s(real bytes)  %5.1f TFLOP/s" % (name, kind, cin, cout, hi, us, bytes_ / us / 1e3, flops / us / 1e6))
print(tot, sum(tot.values()))
