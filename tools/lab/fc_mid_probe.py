"""Stage time stamps of fc_mid_fwd_kernel (workgroup 0, thread 0; s_memrealtime at 100 MHz): needs libava_hip_lab.so built
with tools/lab/build_variant.sh lab fc_mid.hip -DAVA_LAB, run with AVA_HIP_LIB_TAG=lab."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ava_amd import _lib, synthetic as syn
from ava_amd.vae import VAE
lib = _lib.load()
C = ctypes.CDLL(_lib.LIB_PATH)
stamps = torch.zeros(16, dtype=torch.int64, device="cuda")
C.ava_fc_mid_debug_stamps(ctypes.c_void_p(stamps.data_ptr()))
model = VAE(z_dim=32, device_name="cuda")
x = torch.from_numpy(syn.spectrograms(256)).cuda()
for it in range(6):
    model.optimizer.zero_grad()
    model._forward_device(x, need_grad=True)
    model._backward_device(x)
    model.optimizer.step()
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()
    print("iter", it, "stage us:", [round((int(s[i + 1]) - int(s[i])) * 0.01, 2) for i in range(6)], "total", round((int(s[6]) - int(s[0])) * 0.01, 2))
