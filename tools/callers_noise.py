"""How far the trained BatchNorm running means of the callers' sequence (tests/test_gpu_callers.py) are from the
reference's, in units of the reference's own run-to-run difference (8 vs 1 CPU threads, stored in the golden)."""
import os, sys, io, contextlib, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import load_golden
from ava_amd import synthetic as syn
from ava_amd.vae import VAE
G = load_golden("callers.npz")
z, B, nb = 32, 8, 2
root = tempfile.mkdtemp()
model = VAE(save_dir=root)
fp = syn.fixture_parameters(z)
with torch.no_grad():
    for name, p in model.named_parameters():
        p.copy_(torch.from_numpy(fp[name]))
ds = syn.SyntheticSpecDataset(B * nb)
loaders = {"train": torch.utils.data.DataLoader(ds, batch_size=B, shuffle=False)}
loaders["test"] = loaders["train"]
queue = []
for ep in range(2):
    for k in range(nb):
        queue.append(syn.noise(B, z, 2002 + 10 * k + ep, 3003 + 10 * k + ep))
    queue.append(syn.noise(5, z, 2500 + ep, 3500 + ep))
model.noise_source = lambda b, zz: queue.pop(0)
np.random.seed(1234)
with contextlib.redirect_stdout(io.StringIO()):
    model.train_loop(loaders, epochs=2, test_freq=None)
noise = max(float(G["selfnoise.trained.bn%d.running_mean" % i]) for i in range(1, 15))
r = [float(np.abs(getattr(model, "bn%d" % i).running_mean.cpu().numpy() - G["trained.bn%d.running_mean" % i]).max()) / noise for i in range(1, 15)]
print("lib tag %s AVA_BN_ACC=%s: max |running_mean - reference| / reference self-noise per layer:" % (os.environ.get("AVA_HIP_LIB_TAG"), os.environ.get("AVA_BN_ACC")))
print(" ".join("%.2f" % v for v in r), " max %.2f" % max(r), " train_loss[1] ratio %.2f" % (abs(model.loss["train"][1] - G["train_loss"][1]) / float(G["selfnoise.train_loss"])))
