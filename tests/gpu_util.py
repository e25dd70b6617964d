"""Helpers for the -m gpu tests: thin wrappers that call the C ABI of libava_hip.so with torch tensors."""
import ctypes

import numpy as np
import torch

from ava_amd import _lib

MODE_S1, MODE_DOWN, MODE_UP = 0, 1, 2
PRO_BN, PRO_BWD, PRO_ID = 0, 1, 2
EPI_FWD, EPI_BWD, EPI_SSE = 0, 1, 2

# (name, cin, cout, mode, input size, transposed) -- ava/models/vae.py:128-134,155-161
LAYERS = [
    ("conv1", 1, 8, MODE_S1, 128, 0), ("conv2", 8, 8, MODE_DOWN, 128, 0), ("conv3", 8, 16, MODE_S1, 64, 0),
    ("conv4", 16, 16, MODE_DOWN, 64, 0), ("conv5", 16, 24, MODE_S1, 32, 0), ("conv6", 24, 24, MODE_DOWN, 32, 0),
    ("conv7", 24, 32, MODE_S1, 16, 0), ("convt1", 32, 24, MODE_S1, 16, 1), ("convt2", 24, 24, MODE_UP, 16, 1),
    ("convt3", 24, 16, MODE_S1, 32, 1), ("convt4", 16, 16, MODE_UP, 32, 1), ("convt5", 16, 8, MODE_S1, 64, 1),
    ("convt6", 8, 8, MODE_UP, 64, 1), ("convt7", 8, 1, MODE_S1, 128, 1),
]


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda().contiguous()


def p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def out_size(hi, mode):
    return hi if mode == MODE_S1 else (hi // 2 if mode == MODE_DOWN else hi * 2)


def pack(w, kind):
    lib = _lib.load()
    g = torch.empty(w.numel(), device="cuda")
    _lib.check(lib.ava_pack_conv_weight(p(w), p(g), w.shape[0], w.shape[1], kind, stream()), "pack")
    torch.cuda.synchronize()
    return g


def conv3x3(inp, G, cin, cout, mode, pro, epi, B, hi, in2=None, pa=None, pb=None, pc=None, bias=None, relu=0,
            epi_x=None, epi_mean=None, epi_invstd=None, prec=0.0, want_out=True):
    lib = _lib.load()
    ho = out_size(hi, mode)
    grid = lib.ava_conv_grid(B, ho, ho, mode)
    out = torch.empty(B, ho, ho, cout, device="cuda") if want_out else None
    out2 = torch.empty(B, ho, ho, cout, device="cuda") if epi == EPI_SSE else None
    partials = torch.zeros(grid, 2 * cout, device="cuda")
    rc = lib.ava_conv3x3(p(inp), p(in2), p(pa), p(pb), p(pc), p(G), p(bias), p(out), p(out2), p(epi_x), p(epi_mean),
                         p(epi_invstd), p(partials), B, hi, hi, cin, cout, mode, pro, epi, relu, prec, stream())
    _lib.check(rc, "ava_conv3x3")
    torch.cuda.synchronize()      # callers pass temporaries; keep them alive until the kernel is done
    return out, out2, partials


def wgrad(x, xa, xb, dy, cin, cout, mode, dy_pro, B, hi, dy2=None, da=None, db=None, dc=None, kind=0):
    lib = _lib.load()
    ho = out_size(hi, mode)
    grid = lib.ava_conv_wgrad_grid(B, ho, ho, mode)
    partials = torch.zeros(grid, 9 * cin * cout + cout, device="cuda")
    rc = lib.ava_conv3x3_wgrad(p(x), p(xa), p(xb), p(dy), p(dy2), p(da), p(db), p(dc), p(partials), B, hi, hi, cin,
                               cout, mode, dy_pro, stream())
    _lib.check(rc, "ava_conv3x3_wgrad")
    dw = torch.empty(9 * cin * cout, device="cuda")
    dbias = torch.empty(cout, device="cuda")
    rows = lib.ava_conv_wgrad_rows(B, hi, hi, cin, cout, mode, dy_pro)      # rows actually written (<= grid)
    assert 0 < rows <= grid
    assert float(partials[rows:].abs().sum()) == 0.0
    _lib.check(lib.ava_conv_wgrad_reduce(p(partials), rows, p(dw), p(dbias), cin, cout, kind, stream()), "reduce")
    torch.cuda.synchronize()
    return dw, dbias


def bwd_fused(x, xa, xb, dy, Gb, mean, invstd, cin, cout, mode, dy_pro, B, hi, dy2=None, da=None, db=None, dc=None,
              kind=0):
    """fused backward: (dx, bn partial sums [2*cin] as float64, dw, dbias) or None when the shape has no instantiation"""
    lib = _lib.load()
    grid = lib.ava_conv_fused_grid(B, hi, hi, cin, cout, mode)
    if grid <= 0:
        return None
    dx = torch.empty(B, hi, hi, cin, device="cuda") if cin > 1 else None     # the 1 -> 8 layer forms no data gradient
    bnp = torch.zeros(grid, 2 * cin, device="cuda")
    wgp = torch.zeros(grid, 9 * cin * cout + cout, device="cuda")
    rc = lib.ava_conv3x3_bwd_fused(p(x), p(xa), p(xb), p(dy), p(dy2), p(da), p(db), p(dc), p(Gb), p(dx), p(mean),
                                   p(invstd), p(bnp), p(wgp), B, hi, hi, cin, cout, mode, dy_pro, stream())
    _lib.check(rc, "ava_conv3x3_bwd_fused")
    dw = torch.empty(9 * cin * cout, device="cuda")
    dbias = torch.empty(cout, device="cuda")
    _lib.check(lib.ava_conv_wgrad_reduce(p(wgp), grid, p(dw), p(dbias), cin, cout, kind, stream()), "reduce")
    torch.cuda.synchronize()
    return dx, bnp.double().sum(dim=0).cpu(), dw, dbias


def gemm(A, B, M, N, K, a_k, b_k, bias=None, act=0, mask=None, colsum=False, lda=0, ldb=0, ldc=0, C=None):
    lib = _lib.load()
    nbytes = lib.ava_gemm_workspace_bytes(M, N, K)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
    if C is None:
        C = torch.zeros(M, ldc if ldc else N, device="cuda")
    cs = torch.zeros(M, device="cuda") if colsum else None
    rc = lib.ava_gemm(p(A), lda, p(B), ldb, p(bias), p(C), ldc, p(mask), p(cs), M, N, K, a_k, b_k, act, p(ws), nbytes,
                      stream())
    _lib.check(rc, "ava_gemm")
    torch.cuda.synchronize()
    return C, cs


def rel(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def nhwc(t):
    """torch NCHW -> contiguous NHWC"""
    return t.permute(0, 2, 3, 1).contiguous()


def build_model(z=32, fixture=True, train=True):
    from ava_amd import synthetic as syn
    from ava_amd.vae import VAE
    m = VAE(z_dim=z, device_name="cuda")
    if fixture:
        fp = syn.fixture_parameters(z)
        with torch.no_grad():
            for name, prm in m.named_parameters():
                prm.copy_(torch.from_numpy(fp[name]))
    m.train(train)
    return m
