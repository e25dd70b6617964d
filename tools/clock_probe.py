#!/usr/bin/env python3
"""Shader clock while the fp32 matrix cores are saturated (large ava_gemm in a loop) vs idle: the sustained
clock, not the 2.4 GHz boost clock, prices the fp32 MFMA ceiling of the GEMM rows in DESIGN.md."""
import os, sys, subprocess, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import p, stream
from ava_amd import _lib
lib = _lib.load()
def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=20).stdout
        return "\n".join(l for l in out.splitlines() if "sclk" in l or "Power" in l or "mclk" in l)
    except Exception as e:
        return "rocm-smi failed: %r" % (e,)
print("idle:\n" + smi())
M = N = 4096; K = 4096
A = torch.randn(M * K, device="cuda"); B = torch.randn(K * N, device="cuda"); C = torch.empty(M, N, device="cuda")
res = []
th = threading.Timer(1.5, lambda: res.append(smi()))
th.start()
t0 = time.perf_counter(); n = 0
while time.perf_counter() - t0 < 4.0:
    for _ in range(20):
        lib.ava_gemm(p(A), 0, p(B), 0, None, p(C), 0, None, None, M, N, K, 1, 1, 0, None, 0, stream())
    torch.cuda.synchronize(); n += 20
dt = time.perf_counter() - t0
th.join()
print("under fp32 MFMA load (%.1f TFLOP/s):\n%s" % (2.0 * M * N * K * n / dt / 1e12, res[0] if res else "?"))
