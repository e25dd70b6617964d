#!/usr/bin/env python3
"""A/B of one lab switch on the gradients of a fixed step: run once per value, then diff the dumps.
    AVA_HIP_LIB_TAG=lab AVA_X=0 python tools/ab_grads.py dump /tmp/a.npz ; AVA_X=1 ... dump /tmp/b.npz ; python tools/ab_grads.py diff /tmp/a.npz /tmp/b.npz"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

if sys.argv[1] == "dump":
    import torch
    from ava_amd import synthetic as syn, layout
    from gpu_util import build_model
    B, z = int(os.environ.get("AB_B", "8")), int(os.environ.get("AB_Z", "64"))
    model = build_model(z)
    ew, ed = syn.noise(B, z)
    model.noise_source = lambda b, zz: (ew, ed)
    x = torch.from_numpy(syn.spectrograms(B)).cuda()
    model.optimizer.zero_grad()
    loss = model.forward(x)
    loss.backward()
    torch.cuda.synchronize()
    offs, total = layout.arena_offsets(z)
    g = model._grads.cpu().numpy()
    np.savez(sys.argv[2], loss=float(loss.item()), **{s.name: g[offs[s.name]:offs[s.name] + s.numel] for s in layout.param_specs(z)})
else:
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    print("loss", float(a["loss"]), float(b["loss"]))
    worst = []
    for k in a.files:
        if k == "loss":
            continue
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()
        n = np.abs(a[k]).max() + 1e-30
        worst.append((d / n, k))
    for r, k in sorted(worst)[::-1][:20]:
        print("%-18s max|diff|/max|a| = %.3e" % (k, r))
