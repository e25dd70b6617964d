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
