#!/usr/bin/env python3
"""Forward passes only (no backward, no Adam) at batch B: for timing-only library variants whose results are wrong (kernel trace
under rocprofv3; AVA_HIP_LIB_TAG selects the variant).  usage: python tools/lab/fwd_loop.py [B] [n]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ava_amd import synthetic as syn
from gpu_util import build_model
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
model = build_model(32)
x = torch.from_numpy(syn.spectrograms(B)).cuda()
for _ in range(n):
    model._forward_device(x, need_grad=True)
torch.cuda.synchronize()
print("done", float(model._loss_buf[0]))
