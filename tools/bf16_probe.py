import sys, os
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
from ava_amd import synthetic as syn
from ava_amd.vae import VAE
from oracle import vae_oracle as O
shape, B, z = (128, 128), 8, 32
fp = syn.fixture_parameters(z, shape)
ew, ed = syn.noise(B, z, 21, 22)
x = torch.from_numpy(syn.spectrograms(B, salt=55, shape=shape))
def hip(act):
    m = VAE(z_dim=z, device_name="cuda", x_shape=shape, act_dtype=act)
    with torch.no_grad():
        for n, p in m.named_parameters(): p.copy_(torch.from_numpy(fp[n]))
    m.noise_source = lambda b, zz: (ew, ed)
    m.train(); m.optimizer.zero_grad()
    l = m.forward(x); l.backward()
    return {n: p.grad.detach().cpu().double().numpy().ravel() for n, p in m.named_parameters()}, float(l.item())
def orc(act, dtype=torch.float32):
    P = O.to_params(fp, dtype=dtype, requires_grad=True)
    out = O.forward(P, x.to(dtype), torch.from_numpy(ew).to(dtype), torch.from_numpy(ed).to(dtype), None, True, act_dtype=act)
    out["loss"].backward()
    return {n: p.grad.double().numpy().ravel() for n, p in P.items()}, float(out["loss"].detach())
gh, lh = hip("bfloat16")
go, lo = orc(torch.bfloat16)
go64, lo64 = orc(torch.bfloat16, torch.float64)
g32, l32 = orc(None)
print("loss hip_bf16 %.9g  oracle_bf16(f32) %.9g  oracle_bf16(f64) %.9g  oracle_f32 %.9g" % (lh, lo, lo64, l32))
print("%-14s %10s %10s %10s %10s" % ("tensor", "hip-vs-o32r", "o64r-vs-o32r", "bf16-vs-f32", "|g|"))
for n in gh:
    r = go[n]; nr = max(np.linalg.norm(r), 1e-300)
    print("%-14s %10.2e %10.2e %10.2e %10.2e" % (n, np.linalg.norm(gh[n]-r)/nr, np.linalg.norm(go64[n]-r)/nr, np.linalg.norm(r-g32[n])/max(np.linalg.norm(g32[n]),1e-300), nr))
