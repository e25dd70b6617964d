#!/usr/bin/env python3
"""Per-step time of the first N training steps of a fresh process (HIP events around every step): does the step time drift after
the 5 warm-up steps the driver's bench run uses?  usage: python tools/lab/step_drift.py [N] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ava_amd import synthetic as syn
from gpu_util import build_model
N = int(sys.argv[1]) if len(sys.argv) > 1 else 120
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
model = build_model(32)
model.noise_source = None
x = torch.from_numpy(syn.spectrograms(B)).cuda()
evs = [torch.cuda.Event(enable_timing=True) for _ in range(N + 1)]
if os.environ.get("DRIFT_PREHEAT"):                      # unrelated GPU work first: is the drift the chip's (clocks) or this library's?
    a = torch.randn(8192, 8192, device="cuda"); b = torch.randn(8192, 8192, device="cuda")
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(int(os.environ["DRIFT_PREHEAT"])): c = a @ b
    t1.record(); torch.cuda.synchronize()
    print("preheat: %.1f ms of fp32 matmuls" % t0.elapsed_time(t1))
    del a, b, c
import time
host = []
evs[0].record()
for i in range(N):
    h0 = time.perf_counter()
    model.optimizer.zero_grad()
    model._forward_device(x, need_grad=True)
    model._backward_device(x)
    model.optimizer.step()
    evs[i + 1].record()
    host.append(1e3 * (time.perf_counter() - h0))
torch.cuda.synchronize()
ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(N)]
print("host ms to ENQUEUE a step: steps 1-5 %s; 6-25 mean %.3f; 26+ mean %.3f" % (" ".join("%.2f" % h for h in host[:5]), sum(host[5:25]) / 20, sum(host[25:]) / max(1, len(host) - 25)))
def mean(a): return sum(a) / len(a)
print("steps 1-5: %s" % " ".join("%.3f" % m for m in ms[:5]))
for lo in range(5, N, 20):
    seg = ms[lo:lo + 20]
    print("steps %3d-%3d: mean %.4f ms  min %.4f  max %.4f" % (lo + 1, lo + len(seg), mean(seg), min(seg), max(seg)))
