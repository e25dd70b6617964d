"""Pinned double-buffered host -> device feeding of the epoch loops (ava_amd/feed.py; replaces the synchronous
``data.to(self.device)`` of vae.py:349 / 374 / 540).  The feeder must hand over bit-identical batches in loader
order, cope with ragged / float64 / uint8 batches and early exits, surface loader errors, and leave the epoch
loops' results bit-identical to the synchronous path."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from gpu_util import build_model
from ava_amd import synthetic as syn
from ava_amd.feed import DeviceFeeder


class _Loader:
    """Minimal loader: an iterable of CPU tensors with ``.dataset`` (what train_epoch needs)."""

    def __init__(self, batches):
        self.batches = batches
        self.dataset = list(range(sum(len(b) for b in batches)))

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def _batches(sizes, dtype=torch.float32):
    out, start = [], 0
    for n in sizes:
        out.append(torch.from_numpy(syn.spectrograms(n, start_item=start)).to(dtype))
        start += n
    return out


def test_feeder_yields_identical_batches_in_order():
    sizes = [16, 16, 16, 16, 16, 16, 16, 5]                # more batches than slots, ragged tail
    batches = _batches(sizes)
    feeder = DeviceFeeder(_Loader(batches), "cuda", depth=2)
    assert len(feeder) == len(batches) and len(feeder.dataset) == sum(sizes)
    seen = []
    for dev in feeder:
        assert dev.is_cuda and dev.dtype == torch.float32
        seen.append(dev.clone())                           # the slot is recycled after the next() call
    assert len(seen) == len(batches)
    for got, want in zip(seen, batches):
        assert torch.equal(got.cpu(), want)


def test_feeder_converts_dtypes_and_grows():
    b64 = _batches([4, 9], torch.float64)                  # second batch larger than the first: slots regrow
    got = [d.clone() for d in DeviceFeeder(_Loader(b64), "cuda")]
    for g, w in zip(got, b64):
        assert torch.equal(g.cpu(), w.to(torch.float32))
    u8 = [(torch.rand(3, 128, 128) * 255).to(torch.uint8)]
    got = [d.clone() for d in DeviceFeeder(_Loader(u8), "cuda")]
    assert torch.equal(got[0].cpu(), u8[0].to(torch.float32))


def test_feeder_early_exit_and_errors():
    feeder = DeviceFeeder(_Loader(_batches([8] * 6)), "cuda", depth=2)
    for i, dev in enumerate(feeder):
        if i == 1:
            break                                          # generator closed with batches still queued: no hang
    def bad():
        yield torch.zeros(2, 128, 128)
        raise RuntimeError("loader failed")
    class Bad:
        dataset = [0, 1]
        def __iter__(self):
            return bad()
    with pytest.raises(RuntimeError, match="loader failed"):
        for _ in DeviceFeeder(Bad(), "cuda"):
            pass
    with pytest.raises(ValueError):
        for _ in DeviceFeeder(_Loader([torch.zeros(2, 16384)]), "cuda"):
            pass


def test_epoch_loops_identical_with_and_without_prefetch():
    B, nb, z = 8, 5, 32
    batches = _batches([B] * (nb - 1) + [3])
    results = []
    for prefetch in (False, True):
        model = build_model(z)
        model.prefetch = prefetch
        noise = [syn.noise(len(b), z, 2002 + k, 3003 + k) for k, b in enumerate(batches)] * 2
        model.noise_source = lambda b, zz: noise.pop(0)
        train = model.train_epoch(_Loader(batches))
        test = model.test_epoch(_Loader(batches))
        model.noise_source = None
        lat = model.get_latent(_Loader(batches))
        results.append((train, test, lat, model._params.detach().cpu().clone()))
    (t0, e0, l0, p0), (t1, e1, l1, p1) = results
    assert t0 == t1 and e0 == e1                            # same kernels, same inputs: bit-identical
    assert np.array_equal(l0, l1) and l0.shape == (sum(len(b) for b in batches), z) and l0.dtype == np.float64
    assert torch.equal(p0, p1)


def test_get_latent_bn_mode():
    """bn_mode=None keeps the reference's quirk (current mode, here train: batch statistics, running stats move);
    'eval' uses the running statistics, is independent of the batching and leaves mode and buffers untouched."""
    z = 32
    data = _batches([16, 16, 16, 16])
    model = build_model(z)
    model.train()
    before = model._bn_running.clone()
    a = model.get_latent(_Loader(data), bn_mode='eval')
    assert model.training and torch.equal(model._bn_running, before)
    merged = [torch.cat(data[:2]), torch.cat(data[2:])]
    b = model.get_latent(_Loader(merged), bn_mode='eval')
    assert np.allclose(a, b, rtol=1e-4, atol=1e-5)                     # same statistics whatever the batching
    c = model.get_latent(_Loader(data))                                 # reference behaviour: train mode
    assert not torch.equal(model._bn_running, before) and not np.allclose(a, c, rtol=1e-3, atol=1e-4)
    with pytest.raises(ValueError):
        model.get_latent(_Loader(data), bn_mode='nope')


@pytest.mark.parametrize("dtype", [torch.float64, torch.uint8, torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("n", [1, 5, 16, 1000, 3 * 128 * 128 + 7])
def test_device_cast_matches_numpy_to_tensor(dtype, n):
    """ava_cast_to_f32 == the reference's per-item conversion torch.from_numpy(x).type(torch.FloatTensor)
    (ava/models/utils.py:444-446), bit for bit, including lengths that are not a multiple of the vector width."""
    from ava_amd.feed import cast_to_f32
    g = torch.Generator().manual_seed(n)
    if dtype == torch.uint8:
        src = torch.randint(0, 256, (n,), generator=g, dtype=torch.uint8)
    else:
        src = (torch.rand(n, generator=g, dtype=torch.float64) * 2.5 - 0.7).to(dtype)
        if dtype == torch.float64 and n >= 5:
            src[:5] = torch.tensor([0.1, 1e-310, 3e38 * 10, -0.0, 16777217.0], dtype=torch.float64)   # rounding, subnormal, overflow
    want = src.type(torch.FloatTensor)
    got = cast_to_f32(src.cuda()).cpu()
    assert got.dtype == torch.float32 and torch.equal(got.view(torch.int32), want.view(torch.int32))


@pytest.mark.parametrize("dtype", [torch.float64, torch.uint8])
def test_ring_loader_plus_feeder_equals_reference_loader_path(dtype):
    """PinnedBatchLoader (items collated in their own dtype into the page-locked ring) + DeviceFeeder (raw DMA + device
    cast) hands the model exactly the tensors the reference's path produces: per-item numpy_to_tensor, default
    collation, .to(device)."""
    from ava_amd.feed import PinnedBatchLoader
    n, bs = 37, 8
    base = syn.spectrograms(n)
    items = [(base[i] * 255).astype(np.uint8) if dtype == torch.uint8 else base[i].astype(np.float64) * (1 + 1e-9 * i)
             for i in range(n)]
    ref_items = [torch.from_numpy(x).type(torch.FloatTensor) for x in items]          # numpy_to_tensor
    ref_batches = [torch.stack(ref_items[s:s + bs]) for s in range(0, n, bs)]
    loader = PinnedBatchLoader(items, batch_size=bs, depth=3)
    got = [d.clone() for d in DeviceFeeder(loader, "cuda", depth=2)]
    assert len(got) == len(ref_batches) == 5
    for g_, w in zip(got, ref_batches):
        assert g_.dtype == torch.float32 and torch.equal(g_.cpu(), w)
    first = next(iter(loader))
    assert first.is_pinned() and first.dtype == dtype


def test_train_epoch_over_float64_ring_is_bit_identical_to_reference_loader():
    from ava_amd.feed import PinnedBatchLoader
    B, nb, z = 8, 4, 32
    base = syn.spectrograms(B * nb)
    items64 = [base[i].astype(np.float64) for i in range(B * nb)]
    ref_loader = _Loader([torch.from_numpy(base[s:s + B]) for s in range(0, B * nb, B)])
    res = []
    for loader in (ref_loader, PinnedBatchLoader(items64, batch_size=B)):
        model = build_model(z)
        noise = [syn.noise(B, z, 2002 + k, 3003 + k) for k in range(nb)]
        model.noise_source = lambda b, zz: noise.pop(0)
        res.append((model.train_epoch(loader), model._params.detach().cpu().clone()))
    assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1])
