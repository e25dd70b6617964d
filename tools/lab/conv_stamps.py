"""Where a wave-specialised forward conv kernel's time goes (lab build: make lab; AVA_HIP_LIB_TAG=lab): s_memrealtime stamps
(100 MHz) of workgroup 0 of every conv3x3_mfma_ws_kernel launch of a training step at batch B (argv[1], default 256).
Columns (us, relative to the kernel's first stamp): coefficients ready / first tile staged (staging wave) / matrix waves reach
barrier A / pass it / first four tiles done / loop done / statistics written / exit; `gap` = this kernel's entry minus the
previous stamped kernel's exit (includes whatever un-stamped kernels ran in between)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from ava_amd import _lib, synthetic as syn
from ava_amd.vae import VAE
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
_lib.load()
C = ctypes.CDLL(_lib.LIB_PATH)
N = 24
stamps = torch.zeros(N * 16, dtype=torch.int64, device="cuda")
model = VAE(z_dim=32, device_name="cuda")
x = torch.from_numpy(syn.spectrograms(B)).cuda()
names = ["conv2", "conv3", "conv4", "conv5", "conv6", "convt1", "convt2", "convt3", "convt4", "convt5",
         "convt6'", "convt5'", "convt4'", "convt3'", "convt2'", "convt1'", "conv7'", "conv6'", "conv5'", "conv4'", "conv3'", "conv2'"]
for it in range(5):
    stamps.zero_()
    torch.cuda.synchronize()
    C.ava_lab_conv_stamps(ctypes.c_void_p(stamps.data_ptr()), N)
    model.optimizer.zero_grad()
    model._forward_device(x, need_grad=True)
    model._backward_device(x)
    model.optimizer.step()
    torch.cuda.synchronize()
    C.ava_lab_conv_stamps(None, 0)
    if it < 3:
        continue
    s = stamps.cpu().numpy().reshape(N, 16)
    prev_exit = None
    print("iter", it)
    for k in range(N):
        r = s[k]
        if r[0] == 0:
            continue
        t0 = int(r[0])
        rel = lambda i: ("%6.2f" % ((int(r[i]) - t0) * 0.01)) if r[i] else "   -  "
        gap = "%6.2f" % ((t0 - prev_exit) * 0.01) if prev_exit else "   -  "
        print("%-7s gap %s | coef %s stage0 %s | A-in %s A-out %s | tiles %s %s %s %s | loop %s stats %s exit %s | wgrad A-in %s" % (
            names[k] if k < len(names) else str(k), gap, rel(1), rel(2), rel(3), rel(4), rel(5), rel(6), rel(7), rel(8), rel(10), rel(11), rel(12), rel(13)))
        prev_exit = int(r[12]) if r[12] else None
