#!/usr/bin/env python3
"""Time every GEMM shape of one VAE train step through the C ABI (torch events on the current stream)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import gemm, p, stream
from ava_amd import _lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
z = 32
fc = [("fc1", 8192, 1024), ("fc2", 1024, 256), ("fc3x", 256, 192), ("fc4x", 64, z), ("fc5", z, 64), ("fc6", 64, 256),
      ("fc7", 256, 1024), ("fc8", 1024, 8192)]
lib = _lib.load()
tot = {"fwd": 0.0, "dX": 0.0, "dW": 0.0}
for name, fin, fout in fc:
    for kind, (M, N, K, ak, bk) in (("fwd", (B, fout, fin, 1, 1)), ("dX", (B, fin, fout, 1, 0)), ("dW", (fout, fin, B, 0, 0))):
        A = torch.randn(M * K, device="cuda"); Bm = torch.randn(K * N, device="cuda")
        C = torch.empty(M, N, device="cuda"); cs = torch.empty(M, device="cuda")
        nbytes = lib.ava_gemm_workspace_bytes(M, N, K)
        ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
        def run():
            return lib.ava_gemm(p(A), 0, p(Bm), 0, None, p(C), 0, None, p(cs) if kind == "dW" else None, M, N, K, ak, bk, 0, p(ws), nbytes, stream())
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n): run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        # headroom probe: the vendor library (torch.mm -> rocBLAS / hipBLASLt) on the same product
        At = A.view(M, K) if ak else A.view(K, M).t()
        Bt = Bm.view(N, K).t() if bk else Bm.view(K, N)
        for _ in range(3): torch.mm(At, Bt, out=C)
        e0.record()
        for _ in range(n): torch.mm(At, Bt, out=C)
        e1.record(); torch.cuda.synchronize()
        us_lib = e0.elapsed_time(e1) * 1e3 / n
        # accuracy against an fp64 product of the same fp32 operands (ours / the vendor library's fp32)
        ref = At.double() @ Bt.double()
        lib.ava_gemm(p(A), 0, p(Bm), 0, None, p(C), 0, None, p(cs) if kind == "dW" else None, M, N, K, ak, bk, 0, p(ws), nbytes, stream())
        torch.cuda.synchronize()
        den = float((At.double().abs() @ Bt.double().abs()).max())
        err = float((C.double() - ref).abs().max()) / den
        err_lib = float((torch.mm(At, Bt).double() - ref).abs().max()) / den
        mult = 3 if name in ("fc4x",) else 1
        tot[kind] += us * mult
        print("%-5s %-3s M=%5d N=%5d K=%5d  %7.1f us  %6.1f TFLOP/s  ws=%d MB   torch.mm %7.1f us   err/sum|ab| %.2e (torch.mm %.2e)" % (name, kind, M, N, K, us, 2.0 * M * N * K / us / 1e6, nbytes >> 20, us_lib, err, err_lib))
print(tot, sum(tot.values()))
