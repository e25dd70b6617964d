#!/usr/bin/env python3
"""PCIe-inclusive rate of VAE.train_epoch over a loader of pageable CPU batches (the reference's hand-over,
vae.py:349): synchronous ``.to(device)`` per step vs the prefetching DeviceFeeder (ava_amd/feed.py).
The device-resident rate of bench.py is printed beside it.  Usage: python tools/feed_bench.py [batch] [batches]"""
import os, sys, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ava_amd import synthetic as syn
from ava_amd.vae import VAE

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 40


class Loader:
    def __init__(self, batches):
        self.batches = batches
        self.dataset = range(sum(len(b) for b in batches))
    def __iter__(self):
        return iter(self.batches)
    def __len__(self):
        return len(self.batches)


pool = [torch.from_numpy(syn.spectrograms(B, salt=1001, start_item=i * B)) for i in range(8)]
loader = Loader([pool[i % 8] for i in range(NB)])
model = VAE(z_dim=32, device_name="cuda")


def epoch(prefetch):
    model.prefetch = prefetch
    with contextlib.redirect_stdout(io.StringIO()):
        model.train_epoch(loader)                      # warm-up epoch
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train_epoch(loader)
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return B * NB / dt, 1e3 * dt / NB


dev_pool = [p.cuda() for p in pool]
model.prefetch = False
resident = Loader([dev_pool[i % 8] for i in range(NB)])
with contextlib.redirect_stdout(io.StringIO()):
    model.train_epoch(resident); torch.cuda.synchronize()
    t0 = time.perf_counter(); model.train_epoch(resident); torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("device-resident batches      : %9.0f spectrograms/s  %.3f ms/step" % (B * NB / dt, 1e3 * dt / NB))
for name, pf in (("synchronous .to(device)", False), ("prefetching feeder, pageable", True)):
    r, ms = epoch(pf)
    print("%-29s: %9.0f spectrograms/s  %.3f ms/step" % (name, r, ms))
pinned = [p.pin_memory() for p in pool]
loader = Loader([pinned[i % 8] for i in range(NB)])       # what DataLoader(pin_memory=True) hands over
for name, pf in (("synchronous, pinned batches", False), ("prefetching feeder, pinned", True)):
    r, ms = epoch(pf)
    print("%-29s: %9.0f spectrograms/s  %.3f ms/step" % (name, r, ms))

# ---- build-owned page-locked ring (PinnedBatchLoader): collation alone, then the whole fed epoch ----
import numpy as np
from ava_amd.feed import PinnedBatchLoader
base = np.concatenate([p.numpy() for p in pool])                     # [8*B,128,128] float32 in host memory
for dt_name, data in (("float32", base), ("float64", base.astype(np.float64)), ("uint8", (base * 255).astype(np.uint8))):
    for workers in (1, 4, 8):
        L = PinnedBatchLoader(data, batch_size=B, shuffle=True, workers=workers, prefetch=False)
        list(L)
        t0 = time.perf_counter(); n = 0
        for _ in range(5):
            for b in L:
                n += 1
        print("collate only  %-8s workers=%d : %.3f ms/batch" % (dt_name, workers, 1e3 * (time.perf_counter() - t0) / n))
    for workers, pf in ((4, True), (8, True), (4, False)):
        class Rep:                                                     # NB batches per epoch out of the 8-batch array
            def __init__(self): self.l = PinnedBatchLoader(data, batch_size=B, shuffle=True, workers=workers, prefetch=pf)
            dataset = range(B * NB)
            def __len__(self): return NB
            def __iter__(self):
                for _ in range(NB // 8):
                    yield from self.l
        loader = Rep()
        r, ms = epoch(True)
        print("ring %-8s workers=%d producer-thread=%-5s: %9.0f spectrograms/s  %.3f ms/step" % (dt_name, workers, pf, r, ms))
