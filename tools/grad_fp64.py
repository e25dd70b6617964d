#!/usr/bin/env python3
"""Per-tensor gradient error of the HIP path and of the fp32 CPU oracle, both against an fp64 evaluation of the
oracle, on the seeded fixtures (B = 8, 64, 256) and on the flip-free fixture (tests/flipfree.py).  Prints a table and
writes gpurun_out/grad_fp64.json.  GPU box:  python tools/grad_fp64.py [--batches 8,64,256]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch

from ava_amd import synthetic as syn
from ava_amd.layout import param_specs
from oracle import vae_oracle as O


def oracle_grads(fp, x, ew, ed, dtype):
    P = O.to_params(fp, dtype=dtype, requires_grad=True)
    out = O.forward(P, torch.as_tensor(x, dtype=dtype), torch.as_tensor(ew, dtype=dtype), torch.as_tensor(ed, dtype=dtype), None, True)
    out["loss"].backward()
    return {k: v.grad.double().numpy().ravel() for k, v in P.items()}, float(out["loss"])


def hip_grads(fp, x, ew, ed, z):
    from gpu_util import build_model
    m = build_model(z, fixture=False)
    with torch.no_grad():
        for name, p in m.named_parameters():
            p.copy_(torch.from_numpy(fp[name]))
    m.noise_source = lambda b, zz: (ew, ed)
    m.optimizer.zero_grad()
    loss = m.forward(torch.from_numpy(x))
    loss.backward()
    return {n: p.grad.detach().cpu().double().numpy().ravel() for n, p in m.named_parameters()}, float(loss.item())


def table(fp, x, ew, ed, z, tag):
    t0 = time.time()
    g64, l64 = oracle_grads(fp, x, ew, ed, torch.float64)
    g32, l32 = oracle_grads(fp, x, ew, ed, torch.float32)
    gh, lh = hip_grads(fp, x, ew, ed, z)
    rows = {}
    tot = {"hip": 0.0, "o32": 0.0, "ref": 0.0}
    for s in param_specs(z):
        r = g64[s.name]
        n = max(np.linalg.norm(r), 1e-300)
        rows[s.name] = (np.linalg.norm(gh[s.name] - r) / n, np.linalg.norm(g32[s.name] - r) / n, n)
        tot["hip"] += np.linalg.norm(gh[s.name] - r) ** 2
        tot["o32"] += np.linalg.norm(g32[s.name] - r) ** 2
        tot["ref"] += n ** 2
    print("== %s: loss hip %.9g o32 %.9g o64 %.12g  (%.0f s)" % (tag, lh, l32, l64, time.time() - t0))
    print("   global rel L2: hip %.2e  o32 %.2e" % ((tot["hip"] / tot["ref"]) ** 0.5, (tot["o32"] / tot["ref"]) ** 0.5))
    worst = sorted(rows.items(), key=lambda kv: -kv[1][0])[:12]
    for k, (eh, eo, n) in worst:
        print("   %-16s hip %.2e  o32 %.2e  ratio %6.2f  |g| %.3e" % (k, eh, eo, eh / max(eo, 1e-30), n))
    return {"loss": [lh, l32, l64], "global": [(tot["hip"] / tot["ref"]) ** 0.5, (tot["o32"] / tot["ref"]) ** 0.5],
            "tensors": {k: [float(a), float(b), float(c)] for k, (a, b, c) in rows.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="8,64,256")
    ap.add_argument("--z", type=int, default=32)
    a = ap.parse_args()
    torch.set_num_threads(min(os.cpu_count() or 8, 32))
    z = a.z
    res = {}
    for B in [int(b) for b in a.batches.split(",")]:
        x = syn.spectrograms(B)
        ew, ed = syn.noise(B, z)
        res["B%d" % B] = table(syn.fixture_parameters(z), x, ew, ed, z, "fixture B=%d" % B)
    from flipfree import flipfree_parameters
    B = 8
    x = syn.spectrograms(B)
    ew, ed = syn.noise(B, z)
    t0 = time.time()
    fp, mn = flipfree_parameters(syn.fixture_parameters(z), x, ew, ed)
    print("flip-free fixture: min |pre-activation| = %.3e (%.0f s)" % (mn, time.time() - t0))
    res["flipfree_B8"] = table(fp, x, ew, ed, z, "flip-free B=8")
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "grad_fp64.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
