#!/usr/bin/env python3
"""Does a training step read workspace memory it has not written?  The same two steps on a model whose workspace was filled with
a byte pattern right after creation (0x00 / 0xFF = NaN floats, -1 counters / 0x7F = 3.4e38 floats) must give bit-identical loss,
gradients and parameters.  Usage: python tools/lab/poison_ws.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from ava_amd import synthetic as syn
from gpu_util import build_model

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
z = 32
x = torch.from_numpy(syn.spectrograms(B)).cuda()
ew, ed = syn.noise(B, z)


def run(fill, dtype="float32", parts=False):
    from ava_amd.vae import VAE
    model = build_model(z)
    if dtype != "float32":
        model = VAE(z_dim=z, device_name="cuda", act_dtype=dtype)
    model.noise_source = lambda b, zz: (ew[:b], ed[:b])
    model._ensure(B)
    if fill is not None:
        model._workspace.fill_(fill)
    out = []
    for step in (1, 2):
        model.optimizer.zero_grad()
        loss = model._forward_device(x, need_grad=True)
        if parts:
            from ava_amd import _lib
            lib = _lib.load()
            for part in range(lib.ava_backward_num_parts()):
                _lib.check(lib.ava_backward_part(model._handle, x.data_ptr(), B, part, _lib.stream()), "part")
            model._grad_state = "filled"
        else:
            model._backward_device(x)
        model.optimizer.step()
        torch.cuda.synchronize()
        out.append((float(model._last_loss) if hasattr(model, "_last_loss") else 0.0, model._grads.clone(), model._params.clone()))
    return out


for parts in (False, True):
  for dtype in ("float32",):
    ref = run(0x00, dtype)
    print("backward in parts:", parts)
    for fill in (None, 0xFF, 0x7F, 0x3F):
        got = run(fill, dtype, parts)
        for step in (0, 1):
            g = torch.equal(ref[step][1], got[step][1]); p = torch.equal(ref[step][2], got[step][2])
            nan = bool(torch.isnan(got[step][1]).any())
            dg = float((ref[step][1].double() - got[step][1].double()).abs().max()) if not nan else float("nan")
            print("%s fill %s step %d: grads %s params %s  max |dgrad| %.3g  nan %s" %
                  (dtype, "none" if fill is None else hex(fill), step + 1, "same" if g else "DIFFER", "same" if p else "DIFFER", dg, nan))
