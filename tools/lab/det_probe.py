"""Run-to-run bit-determinism of a train step at several batch sizes (single process), per gradient tensor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from ava_amd import synthetic as syn
from gpu_util import build_model
for B in (8, 64, 256):
    x = torch.from_numpy(syn.spectrograms(B)).cuda()
    ew, ed = syn.noise(B, 32)
    outs = []
    for rep in range(3):
        model = build_model(32)
        model.noise_source = lambda b, zz: (ew, ed)
        model.optimizer.zero_grad()
        loss = model.forward(x)
        loss.backward()
        torch.cuda.synchronize()
        outs.append((float(loss.item()), {n: p.grad.clone() for n, p in model.named_parameters()}))
    for rep in (1, 2):
        bad = [n for n in outs[0][1] if not torch.equal(outs[0][1][n], outs[rep][1][n])]
        print("B=%d rep %d: loss equal %s, differing tensors: %s" % (B, rep, outs[0][0] == outs[rep][0], bad[:12]))
