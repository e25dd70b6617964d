#!/usr/bin/env python3
"""Time the fused backward kernel of every layer that has one (B=256) through the C ABI, next to the separate
backward-data + weight-gradient kernels.  AVA_FUSED_VAR=n selects tile variant n (conv_fused.hip)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import LAYERS, MODE_S1, MODE_DOWN, MODE_UP, PRO_BN, PRO_BWD, PRO_ID, EPI_FWD, EPI_BWD, p, stream, out_size
from ava_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
lib = _lib.load()
NBUF = 3          # rotate buffer sets so consecutive launches do not find their inputs in the 256 MB MALL


def timeit(fn, n=9):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


tot_f = tot_u = 0.0
for name, cin, cout, mode, hi, tr in LAYERS:
    fgrid = lib.ava_conv_fused_grid(B, hi, hi, cin, cout, mode)
    if fgrid <= 0: continue
    ho = out_size(hi, mode)
    xs = [torch.rand(B, hi, hi, cin, device="cuda") for _ in range(NBUF)]
    ys = [torch.rand(B, ho, ho, cout, device="cuda") for _ in range(NBUF)]
    gs = [torch.randn(B, ho, ho, cout, device="cuda") for _ in range(NBUF)]
    dxs = [torch.empty(B, hi, hi, cin, device="cuda") for _ in range(NBUF)]
    coef = torch.rand(3, 32, device="cuda")
    G = torch.randn(9 * cin * cout, device="cuda") * 0.1
    bnp = torch.zeros(1024, 64, device="cuda")
    wgrid = lib.ava_conv_wgrad_grid(B, ho, ho, mode)
    wparts = torch.zeros(max(wgrid, fgrid), 9 * cin * cout + cout, device="cuda")
    bmode = MODE_S1 if mode == MODE_S1 else (MODE_UP if mode == MODE_DOWN else MODE_DOWN)
    def fused(i):
        k = i % NBUF
        return lib.ava_conv3x3_bwd_fused(p(xs[k]), p(coef[0]), p(coef[1]), p(gs[k]), p(ys[k]), p(coef[0]), p(coef[1]), p(coef[2]),
                                         p(G), p(dxs[k]) if cin > 1 else None, p(coef[1]), p(coef[2]), p(bnp), p(wparts), B, hi, hi, cin, cout, mode,
                                         PRO_BWD, stream())
    def bwd(i):
        k = i % NBUF
        return lib.ava_conv3x3(p(gs[k]), p(ys[k]), p(coef[0]), p(coef[1]), p(coef[2]), p(G), None, p(dxs[k]), None, p(xs[k]), p(coef[1]),
                               p(coef[2]), p(bnp), B, ho, ho, cout, cin, bmode, PRO_BWD, EPI_BWD, 0, 0.0, stream())
    def wg(i):
        k = i % NBUF
        return lib.ava_conv3x3_wgrad(p(xs[k]), p(coef[0]), p(coef[1]), p(gs[k]), p(ys[k]), p(coef[0]), p(coef[1]), p(coef[2]),
                                     p(wparts), B, hi, hi, cin, cout, mode, PRO_BWD, stream())
    assert fused(0) == 0 and bwd(0) == 0 and wg(0) == 0
    tf, tb, tw = timeit(fused), timeit(bwd), timeit(wg)
    tot_f += tf; tot_u += tb + tw
    nbytes = 4 * B * (2 * hi * hi * cin + 2 * ho * ho * cout)
    print("%-7s %2d->%2d @%3d  fused %7.1f us (%5.0f GB/s)   separate %6.1f + %6.1f = %6.1f us" %
          (name, cin, cout, hi, tf, nbytes / tf / 1e3, tb, tw, tb + tw), flush=True)
print("total fused %.1f us, separate %.1f us" % (tot_f, tot_u))
