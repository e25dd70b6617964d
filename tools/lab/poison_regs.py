#!/usr/bin/env python3
"""Does a training step read a vector register or an LDS word it has not written?  tools/lab/dirty_regs.hip (NaN patterns in all
256 VGPRs and in up to 160 KB of LDS per workgroup) runs over and over on a side stream while the step runs on the main one; loss,
gradients and parameters must stay bit-identical to the undisturbed step (and free of NaN).
Build first: (cd tools/lab && hipcc --offload-arch=gfx950 -O2 -shared -fPIC dirty_regs.hip -o libdirty.so).  Usage: poison_regs.py [batch] [repeats]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ava_amd import synthetic as syn
from gpu_util import build_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
z = 32
D = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libdirty.so"))
D.dirty_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
x = torch.from_numpy(syn.spectrograms(B)).cuda()
ew, ed = syn.noise(B, z)
side = torch.cuda.Stream()


def dirty(times, lds):
    rc = D.dirty_launch(1024, lds, times, ctypes.c_void_p(side.cuda_stream))
    assert rc == 0, rc


def run(disturb, lds=64 * 1024):
    model = build_model(z)
    model.noise_source = lambda b, zz: (ew[:b], ed[:b])
    out = []
    for step in (1, 2):
        model.optimizer.zero_grad()
        if disturb: dirty(40, lds)
        model._forward_device(x, need_grad=True)
        if disturb: dirty(40, lds)
        model._backward_device(x)
        if disturb: dirty(10, lds)
        model.optimizer.step()
        torch.cuda.synchronize()
        out.append((model._loss_buf.clone(), model._grads.clone(), model._params.clone()))
    return out


ref = run(False)
bad = 0
for rep in range(REPS):
    for lds in (0, 32 * 1024, 64 * 1024, 150 * 1024):
        got = run(True, lds)
        for step in (0, 1):
            same = [bool(torch.equal(a, b)) for a, b in zip(ref[step], got[step])]
            nan = bool(torch.isnan(got[step][1]).any())
            if not all(same) or nan:
                bad += 1
                print("rep %d lds %d step %d: loss/grads/params same %s, NaN in gradients %s, max |dgrad| %.4g" %
                      (rep, lds, step + 1, same, nan, float((ref[step][1].double() - got[step][1].double()).abs().max())))
print("batch %d: disturbed steps that differ from the undisturbed one: %d / %d" % (B, bad, REPS * 4 * 2))
