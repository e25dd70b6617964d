#!/usr/bin/env python3
"""Achievable HBM bandwidth of plain streaming kernels at the sizes of the 8-channel full-resolution tensors
(134 MB each at batch 256): the yardstick for the thin / 8-channel conv kernels in DESIGN.md."""
import torch
B = 256
g = torch.randn(B, 128, 128, 8, device="cuda"); y = torch.randn_like(g); o = torch.empty_like(g)
x1 = torch.randn(B, 128, 128, device="cuda")
bufs = [(torch.randn_like(g), torch.randn_like(g), torch.empty_like(g)) for _ in range(3)]   # rotate: defeat the 256 MB MALL
def t(fn, n=12):
    for i in range(3): fn(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fn(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
nb = g.numel() * 4
for name, fn, bytes_ in (
    ("read 1 tensor  (sum)", lambda i: bufs[i % 3][0].sum(), nb),
    ("copy           (1r+1w)", lambda i: bufs[i % 3][2].copy_(bufs[i % 3][0]), 2 * nb),
    ("add            (2r+1w)", lambda i: torch.add(bufs[i % 3][0], bufs[i % 3][1], out=bufs[i % 3][2]), 3 * nb),
    ("mul+sum        (2r)", lambda i: torch.dot(bufs[i % 3][0].view(-1), bufs[i % 3][1].view(-1)), 2 * nb),
    ("fill           (1w)", lambda i: bufs[i % 3][2].fill_(1.0), nb)):
    us = t(fn)
    print("%-24s %7.1f us  %6.0f GB/s" % (name, us, bytes_ / us / 1e3))
