#!/usr/bin/env python3
"""Which ReLU unit has a different sign on the GPU than in the fp64 oracle (B=8 fixture)?  Prints, per fully
connected activation, the units whose mask differs and how close to zero their pre-activation is."""
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import torch.nn.functional as F
from gpu_util import build_model
from ava_amd import synthetic as syn
B, z = 8, int(sys.argv[1]) if len(sys.argv) > 1 else 64
x = torch.from_numpy(syn.spectrograms(B)); ew, ed = syn.noise(B, z)
model = build_model(z); model.noise_source = lambda b, zz: (ew, ed)
loss = model.forward(x.cuda()); torch.cuda.synchronize()
fp = {k: torch.from_numpy(v).double() for k, v in syn.fixture_parameters(z).items()}
def lin(h, n): return F.linear(h, fp[n + ".weight"], fp[n + ".bias"])
h1g = model._workspace_tensor("h1", (B, 1024)).cpu().double()
y7t = model._workspace_tensor("y7t", (B, 8192)).cpu().double()
pre = {}
pre["h1"] = lin(y7t, "fc1"); h = F.relu(pre["h1"])
pre["h2"] = lin(model._workspace_tensor("h1", (B, 1024)).cpu().double(), "fc2")
h2 = model._workspace_tensor("h2", (B, 256)).cpu().double()
pre["h3"] = torch.cat([lin(h2, "fc31"), lin(h2, "fc32"), lin(h2, "fc33")], 1)
zs = model._workspace_tensor("z", (B, z)).cpu().double()
pre["h5"] = lin(zs, "fc5")
pre["h6"] = lin(model._workspace_tensor("h5", (B, 64)).cpu().double(), "fc6")
pre["h7"] = lin(model._workspace_tensor("h6", (B, 256)).cpu().double(), "fc7")
pre["f8"] = lin(model._workspace_tensor("h7", (B, 1024)).cpu().double(), "fc8")
for name, shape in (("h1", (B, 1024)), ("h2", (B, 256)), ("h3", (B, 192)), ("h5", (B, 64)), ("h6", (B, 256)), ("h7", (B, 1024)), ("f8", (B, 8192))):
    got = model._workspace_tensor(name, shape).cpu().double()
    p = pre[name]
    diff = ((got > 0) != (p > 0))
    scale = float(p.abs().mean())
    print("%-3s units with a different mask: %d ; |pre| of those: %s (mean |pre| %.3g); max |relu(pre) - got| %.3g"
          % (name, int(diff.sum()), ["%.2e" % float(v) for v in p[diff].abs()[:5]], scale, float((F.relu(p) - got).abs().max())))
