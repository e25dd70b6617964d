"""The C ABI captures into a HIP graph (rejected item 8): a captured forward + backward replays to
the bit-identical gradients of the eager launches -- in particular the two-slot alternation of bn1's accumulator, which
is host state, must not be frozen into the graph (the captured forward keeps bn1's finalisation launch instead)."""
import numpy as np
import pytest
import torch

from ava_amd import synthetic as syn

pytestmark = pytest.mark.gpu


def test_captured_forward_backward_replays_bit_identically():
    from ava_amd.vae import VAE
    B, z = 16, 32
    model = VAE(save_dir="", z_dim=z, device_name="cuda")
    model.train()
    x = torch.from_numpy(syn.spectrograms(B)).cuda()
    ew, ed = syn.noise(B, z)
    ew_d, ed_d = torch.from_numpy(ew).cuda(), torch.from_numpy(ed).cuda()
    model.noise_source = lambda b, zz: (ew_d, ed_d)            # the same device tensors every call: capturable

    def fwd_bwd():
        model.optimizer.zero_grad()
        model._forward_device(x, need_grad=True, accumulate=False)
        model._backward_device(x)

    for _ in range(3):                                          # eager steps: the accumulator slots alternate
        fwd_bwd()
    torch.cuda.synchronize()
    want = model._grads.clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        fwd_bwd()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fwd_bwd()
    for _ in range(3):                                          # replays must not accumulate anything across each other
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(model._grads, want)
    fwd_bwd()                                                   # and the eager path picks up again afterwards
    torch.cuda.synchronize()
    assert torch.equal(model._grads, want)
    assert np.isfinite(float(want.double().norm()))
