#!/usr/bin/env python3
"""Does replaying the train step as a HIP graph beat the stream launches?  (Captures one step; the captured noise
offset is frozen, which is fine for timing.)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ava_amd import synthetic as syn
from ava_amd.vae import VAE

B = 256
model = VAE(z_dim=32, device_name="cuda"); model.train()
x = torch.from_numpy(syn.spectrograms(B)).cuda()

def one_step():
    model.optimizer.zero_grad()
    model._forward_device(x, need_grad=True, accumulate=True)
    model._backward_device(x)
    model.optimizer.step()

for _ in range(5): one_step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): one_step()
torch.cuda.synchronize()
print("stream launches: %.4f ms/step" % ((time.perf_counter() - t0) * 1e3 / 50))

s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): one_step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    one_step()
for _ in range(5): g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): g.replay()
torch.cuda.synchronize()
print("graph replay:    %.4f ms/step" % ((time.perf_counter() - t0) * 1e3 / 50))
# sanity: the replayed steps trained the same model the eager steps would have (finite parameters, loss keeps falling)
model._loss_acc.zero_()
one_step()
torch.cuda.synchronize()
print("eager step after the replays: loss per sample %.3f, parameters finite: %s"
      % (float(model._loss_acc.item()) / B, bool(torch.isfinite(model._params).all().item()) if hasattr(model, "_params") else "n/a"))
