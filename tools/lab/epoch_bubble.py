#!/usr/bin/env python3
"""Where the host-fed epoch loses against the device-resident step: epoch time of VAE.train_epoch over PinnedBatchLoader as a
function of batches per epoch (T = a + b * nb: a = what an epoch boundary costs, b = the steady-state step), plus the same for
device-resident batches.  Usage: python tools/lab/epoch_bubble.py [batch]"""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ava_amd import synthetic as syn
from ava_amd.vae import VAE
from ava_amd.feed import PinnedBatchLoader

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = VAE(z_dim=32, device_name="cuda")
base = syn.spectrograms(B * 64, salt=1001)


def timed(loader, epochs):
    with contextlib.redirect_stdout(io.StringIO()):
        model.train_epoch(loader); torch.cuda.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(epochs):
                model.train_epoch(loader)
            torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / epochs)
    return best * 1e3


class Resident:
    def __init__(self, nb):
        self.b = [torch.from_numpy(base[i * B:(i + 1) * B]).cuda() for i in range(nb)]
        self.dataset = range(nb * B)
    def __iter__(self): return iter(self.b)
    def __len__(self): return len(self.b)


for name, mk, pf in (("device-resident", lambda nb: Resident(nb), False),
                     ("pinned ring f32", lambda nb: PinnedBatchLoader(base[:nb * B], batch_size=B, shuffle=True), True)):
    model.prefetch = pf
    pts = []
    for nb in (4, 8, 16, 32, 64):
        ms = timed(mk(nb), max(2, 64 // nb))
        pts.append((nb, ms))
        print("%-16s nb %3d : %8.3f ms / epoch  %8.1f spectrograms/s" % (name, nb, ms, nb * B / ms * 1e3))
    x = np.array([p[0] for p in pts], float); y = np.array([p[1] for p in pts], float)
    b, a = np.polyfit(x, y, 1)
    print("%-16s fit: %.3f ms per epoch boundary + %.4f ms per batch (%.1f spectrograms/s steady)" % (name, a, b, B / b * 1e3))
