#!/usr/bin/env python3
"""Per-buffer comparison of one HIP train step against the CPU oracle (prints, never asserts)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ava_amd import synthetic as syn
from ava_amd.vae import VAE
from ava_amd.layout import param_specs
from oracle import vae_oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
z = int(sys.argv[2]) if len(sys.argv) > 2 else 32
torch.set_num_threads(16)
model = VAE(z_dim=z, device_name="cuda")
fp = syn.fixture_parameters(z)
with torch.no_grad():
    for name, p in model.named_parameters():
        p.copy_(torch.from_numpy(fp[name]))
ew, ed = syn.noise(B, z)
model.noise_source = lambda b, zz: (ew, ed)
x = torch.from_numpy(syn.spectrograms(B))
model.train()
loss = model.forward(x)
torch.cuda.synchronize()
print("forward ok, loss", float(loss.item()))

P = O.to_params(fp, requires_grad=True)
rec = {}
running = O.fresh_running_stats()
out = O.forward(P, x, torch.from_numpy(ew), torch.from_numpy(ed), running, True, record=rec)
print("oracle loss", float(out["loss"]))


def rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def ws(name, shape):
    return model._workspace_tensor(name, shape).cpu().numpy()


enc_shapes = [(128, 8), (64, 8), (64, 16), (32, 16), (32, 24), (16, 24), (16, 32)]
for i, (h, c) in enumerate(enc_shapes):
    nm = "y%d" % (i + 1)
    got = ws(nm, (B, h, h, c))
    want = rec["conv%d.out" % (i + 1)].detach().permute(0, 2, 3, 1).numpy()
    print("%-6s rel %.2e" % (nm, rel(got, want)))
print("y7t    rel %.2e" % rel(ws("y7t", (B, 8192)), rec["conv7.out"].detach().reshape(B, 8192).numpy()))
print("h2     rel %.2e" % rel(ws("h2", (B, 256)), rec["fc2.out"].detach().numpy()))
print("mu     rel %.2e" % rel(ws("mu", (B, z)), out["mu"].detach().numpy()))
print("u      rel %.2e" % rel(ws("u", (B, z)), out["u"].detach().numpy()))
print("logd   rel %.2e" % rel(ws("logd", (B, z)), rec["logd"].detach().numpy()))
print("z      rel %.2e" % rel(ws("z", (B, z)), out["z"].detach().numpy()))
print("f8     rel %.2e" % rel(ws("f8", (B, 8192)), rec["fc8.out"].detach().numpy()))
dec_shapes = [(16, 24), (32, 24), (32, 16), (64, 16), (64, 8), (128, 8)]
for i, (h, c) in enumerate(dec_shapes):
    nm = "d%d" % (i + 1)
    got = ws(nm, (B, h, h, c))
    want = rec["convt%d.out" % (i + 1)].detach().permute(0, 2, 3, 1).numpy()
    print("%-6s rel %.2e" % (nm, rel(got, want)))
print("xrec   rel %.2e" % rel(ws("xrec", (B, 16384)), out["x_rec"].detach().numpy()))
lb = model._loss_buf.cpu().numpy()
print("loss %.6e vs %.6e rel %.2e | z2 %.2e sse %.2e H %.2e" % (
    lb[0], float(out["loss"]), rel(lb[0], float(out["loss"])), rel(lb[1], float(out["sum_z2"])),
    rel(lb[2], float(out["sse"])), rel(lb[3], float(out["sum_h"]))))
for i in range(1, 15):
    bn = getattr(model, "bn%d" % i)
    print("bn%-2d running mean %.2e var %.2e nbt %d" % (i, rel(bn.running_mean.cpu().numpy(), running["bn%d.running_mean" % i].numpy()),
          rel(bn.running_var.cpu().numpy(), running["bn%d.running_var" % i].numpy()), int(bn.num_batches_tracked)))

loss.backward()
torch.cuda.synchronize()
out["loss"].backward()
print("backward ok")
for s in param_specs(z):
    g = dict(model.named_parameters())[s.name].grad.cpu().numpy().ravel().astype(np.float64)
    w = P[s.name].grad.numpy().ravel().astype(np.float64)
    print("%-14s |g| %.4e  relL2 %.2e  relmax %.2e" % (s.name, np.linalg.norm(w), np.linalg.norm(g - w) / max(np.linalg.norm(w), 1e-30), rel(g, w)))
model.optimizer.step()
torch.cuda.synchronize()
print("adam ok")
