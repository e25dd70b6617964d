#!/usr/bin/env python3
"""Asymptotic rate of the 128x128 GEMM main loop (large square products, no split-K) vs the short-K products of the model."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import p, stream
from ava_amd import _lib
lib = _lib.load()
for (M, N, K, ak, bk) in ((2048, 2048, 8192, 1, 1), (2048, 2048, 8192, 1, 0), (2048, 2048, 8192, 0, 0), (4096, 4096, 2048, 1, 1),
                          (2048, 2048, 512, 1, 1), (2048, 2048, 256, 1, 1), (1024, 8192, 256, 0, 0), (256, 1024, 8192, 1, 1)):
    A = torch.randn(M * K, device="cuda"); Bm = torch.randn(K * N, device="cuda"); C = torch.empty(M, N, device="cuda")
    nbytes = lib.ava_gemm_workspace_bytes(M, N, K); ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
    run = lambda: lib.ava_gemm(p(A), 0, p(Bm), 0, None, p(C), 0, None, None, M, N, K, ak, bk, 0, p(ws), nbytes, stream())
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): run()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 100
    print("M=%5d N=%5d K=%5d ak=%d bk=%d  %8.1f us  %6.1f TFLOP/s" % (M, N, K, ak, bk, us, 2.0 * M * N * K / us / 1e6), flush=True)
