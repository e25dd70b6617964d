#!/usr/bin/env python3
"""Throughput of the MMD^2 kernels (csrc/mmd.hip) next to the numpy oracle and the reference-style Python double loop.
GPU box:  python tools/mmd_bench.py [--n 20000] [--z 32]"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ava_amd import mmd, synthetic as syn
from oracle import mmd_oracle as MO

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=20000)
ap.add_argument("--z", type=int, default=32)
a = ap.parse_args()
n, z = a.n, a.z
latent = syn.gauss(2 * n * z, 31).reshape(2 * n, z)
latent[n:] += 0.3
i1, i2 = np.arange(n), n + np.arange(n)
sigma = float(np.sqrt(z))
L = mmd._latent_dev(latent)
mmd._terms(L, i1[:128], i2[:128], sigma)                       # warm-up / attribute set-up
torch.cuda.synchronize()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    got = mmd._terms(L, i1, i2, sigma)
    ts.append(time.perf_counter() - t0)
pairs = n * (n - 1) + n * n                                      # kernel evaluations: two upper triangles + the cross term
gpu = min(ts)
# CPU references on a sample that finishes in seconds
m = min(n, 1500)
t0 = time.perf_counter(); want = MO.estimate_mmd2_terms(latent, i1[:m], i2[:m], sigma); cpu_np = time.perf_counter() - t0
mm = min(n, 120)
A = -0.5 / sigma ** 2
t0 = time.perf_counter()
t3 = 0.0
for i in range(mm):
    for j in range(mm):
        t3 += np.exp(A * np.sum(np.power(latent[i1[i]] - latent[i2[j]], 2)))
cpu_loop = time.perf_counter() - t0
sub = mmd._terms(L, i1[:m], i2[:m], sigma)
print(json.dumps({"n_per_condition": n, "z": z, "gpu_s": round(gpu, 5), "gpu_pairs_per_s": pairs / gpu,
                  "gpu_kernel_flops_per_s": pairs * (3 * z + 20) / gpu,
                  "numpy_oracle_pairs_per_s": (m * (m - 1) + m * m) / cpu_np,
                  "reference_python_loop_pairs_per_s": mm * mm / cpu_loop,
                  "max_rel_err_vs_oracle": float(max(abs(g - w) / max(abs(w), 1e-3) for g, w in zip(sub, want))),
                  "mmd2": float(got[3])}))
