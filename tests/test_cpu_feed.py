"""DeviceFeeder host logic without a GPU: on a non-CUDA device the loader is passed through with the same
dtype conversion ``x.to(device, torch.float32)`` performs (ava_amd/feed.py; reference hand-over vae.py:349)."""
import torch

from ava_amd.feed import DeviceFeeder


class _Loader:
    def __init__(self, batches):
        self.batches, self.dataset = batches, list(range(sum(len(b) for b in batches)))

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def test_cpu_passthrough_keeps_order_len_and_converts():
    batches = [torch.rand(4, 128, 128, dtype=torch.float64), torch.rand(2, 128, 128)]
    feeder = DeviceFeeder(_Loader(batches), "cpu")
    assert len(feeder) == 2 and len(feeder.dataset) == 6
    got = list(feeder)
    assert [g.dtype for g in got] == [torch.float32, torch.float32]
    assert torch.equal(got[0], batches[0].to(torch.float32)) and torch.equal(got[1], batches[1])
    assert list(DeviceFeeder(_Loader([]), "cpu")) == []


def test_pinned_batch_loader_collates_in_item_dtype():
    """PinnedBatchLoader host logic (ring reuse, ragged tail, shuffling, dtype kept): no GPU needed -- the ring is
    ordinary memory here, page-locked on a GPU box."""
    import numpy as np
    from ava_amd.feed import PinnedBatchLoader
    items = [np.full((128, 128), i, dtype=np.float64) for i in range(11)]
    loader = PinnedBatchLoader(items, batch_size=4, depth=3)
    assert len(loader) == 3 and loader.dataset is items and loader.batch_size == 4
    seen = []
    for b in loader:
        assert b.dtype == torch.float64 and b.shape[1:] == (128, 128)      # NOT converted on the CPU
        seen.append(b[:, 0, 0].clone())
    assert [len(s) for s in seen] == [4, 4, 3]
    assert torch.equal(torch.cat(seen), torch.arange(11, dtype=torch.float64))
    # more batches than ring slots: slots are reused, contents stay right
    loader = PinnedBatchLoader([np.full((2, 2), i, dtype=np.uint8) for i in range(40)], batch_size=4, depth=3)
    got = torch.cat([b[:, 0, 0].clone() for b in loader])
    assert got.dtype == torch.uint8 and torch.equal(got, torch.arange(40, dtype=torch.uint8))
    g = torch.Generator().manual_seed(5)
    shuffled = torch.cat([b[:, 0, 0].clone() for b in PinnedBatchLoader(items, batch_size=4, shuffle=True, generator=g)])
    assert sorted(shuffled.tolist()) == list(range(11)) and shuffled.tolist() != list(range(11))
    assert shuffled.tolist() == torch.randperm(11, generator=torch.Generator().manual_seed(5)).tolist()


def test_pinned_batch_loader_array_fast_path():
    import numpy as np
    from ava_amd.feed import PinnedBatchLoader
    data = (np.arange(50, dtype=np.float32)[:, None, None] * np.ones((1, 64, 256), np.float32))     # 64 KB items
    for prefetch in (False, True):
        g = torch.Generator().manual_seed(9)
        loader = PinnedBatchLoader(data, batch_size=16, shuffle=True, generator=g, workers=3, prefetch=prefetch, depth=3)
        got = torch.cat([b[:, 5, 7].clone() for b in loader])
        assert got.tolist() == torch.randperm(50, generator=torch.Generator().manual_seed(9)).float().tolist()
        seq = torch.cat([b[:, 0, 0].clone() for b in PinnedBatchLoader(data, batch_size=16, prefetch=prefetch)])
        assert seq.tolist() == list(range(50))
    big = np.zeros((40, 512, 512), np.float32)                  # 1 MB items: the multi-threaded gather (>= 4 MB per batch)
    big += np.arange(40, dtype=np.float32)[:, None, None]
    got = torch.cat([b[:, 100, 100].clone() for b in PinnedBatchLoader(big, batch_size=16, shuffle=True, generator=torch.Generator().manual_seed(1))])
    assert sorted(got.tolist()) == list(range(40))

    class Bad:
        def __len__(self): return 8
        def __getitem__(self, i):
            if i == 5:
                raise RuntimeError("item failed")
            return np.zeros((4, 4), np.float32)
    import pytest
    with pytest.raises(RuntimeError, match="item failed"):
        list(PinnedBatchLoader(Bad(), batch_size=2))
