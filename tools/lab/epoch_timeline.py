#!/usr/bin/env python3
"""Host timeline of one host-fed epoch boundary (PinnedBatchLoader + DeviceFeeder): when the previous epoch's loss read-back returns,
when the first collation starts / ends, when the first step is enqueued.  Usage: python tools/lab/epoch_timeline.py"""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from ava_amd import synthetic as syn
from ava_amd.vae import VAE
from ava_amd import feed, dist as _dist
import importlib
vae_mod = importlib.import_module(VAE.__module__)

B, NB = 256, 8
model = VAE(z_dim=32, device_name="cuda")
base = syn.spectrograms(B * NB, salt=1001)
log = []
T = time.perf_counter
def mark(name): log.append((T(), name))

orig_fill = feed.PinnedBatchLoader._fill
def fill(self, slot, idx):
    mark("fill start"); r = orig_fill(self, slot, idx); mark("fill end"); return r
feed.PinnedBatchLoader._fill = fill
orig_fwd = VAE._forward_device
def fwd(self, *a, **k):
    mark("forward enqueue start"); r = orig_fwd(self, *a, **k); mark("forward enqueued"); return r
VAE._forward_device = fwd
orig_step = model.optimizer.step
def step(*a, **k):
    r = orig_step(*a, **k); mark("adam enqueued"); return r
model.optimizer.step = step
orig_gl = vae_mod._dist.global_loss
def gl(*a, **k):
    mark("loss read-back start"); r = orig_gl(*a, **k); mark("loss read-back end"); return r
vae_mod._dist.global_loss = gl
orig_rp = torch.randperm
def rp(*a, **k):
    mark("randperm start"); r = orig_rp(*a, **k); mark("randperm end"); return r
torch.randperm = rp

loader = feed.PinnedBatchLoader(base, batch_size=B, shuffle=True)
model.prefetch = True
with contextlib.redirect_stdout(io.StringIO()):
    for _ in range(3):
        model.train_epoch(loader)
    torch.cuda.synchronize()
    log.clear()
    mark("epoch A call")
    model.train_epoch(loader)
    mark("epoch B call")
    model.train_epoch(loader)
    mark("epoch B returned")
    torch.cuda.synchronize()
t0 = log[0][0]
for t, n in log:
    print("%9.3f ms  %s" % ((t - t0) * 1e3, n))
