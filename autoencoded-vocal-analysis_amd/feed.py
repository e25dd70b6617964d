"""Prefetching host -> device feeding for the epoch loops.

The reference moves every batch with a synchronous ``data.to(self.device)`` on the compute stream
(``ava/models/vae.py:349, 374, 540``), which puts the 16 MiB copy of a 256-spectrogram batch (0.31 ms over
PCIe) in series with every 2.1 ms step.  ``DeviceFeeder`` wraps any iterable of ``[B,128,128]`` CPU tensors
and yields device-resident fp32 tensors instead, software-pipelined by one batch:

* before batch k is handed out, batch k+1 is pulled from the loader and its H2D copy into one of ``depth``
  device slots is enqueued on a dedicated copy stream, so the transfer runs under the kernels of step k.
  Batches in page-locked memory (``DataLoader(pin_memory=True)``) go as one asynchronous DMA; pageable batches
  go through the runtime's own staging (the call then blocks the host for ~0.5 ms, which is hidden as long as
  the host stays ahead of the GPU);
* the consumer's stream waits on the copy's event (no host synchronisation); a slot is only overwritten after
  the HOST has seen the consumer's kernels on it finish (``released.synchronize()``), which also bounds how far
  the host runs ahead (``depth - 1`` steps).

Measured dead ends (tools/feed_probe.py): staging through own pinned buffers costs the caller a 3 ms memcpy per
batch (longer than the step); a helper thread for it slows the ~100 ctypes launches of a step through GIL
hand-overs (2.1 -> 3.5 ms); making the COPY stream wait for the release event turns the enqueue into a 6 ms
host-side wait.

Values are bit-identical to ``x.to(device, torch.float32)``; ragged last batches and float64 / uint8
loaders are handled by the same ``copy_``.  On a non-CUDA device the loader is passed through.
"""
import torch

__all__ = ["DeviceFeeder"]


class _Slot:
    __slots__ = ("dev", "ready", "released", "n")

    def __init__(self, shape, device):
        self.dev = torch.empty(shape, dtype=torch.float32, device=device)
        self.ready = torch.cuda.Event()        # H2D copy of this slot finished (recorded on the copy stream)
        self.released = torch.cuda.Event()     # consumer's kernels that read ``dev`` are done (compute stream)
        self.n = 0


class DeviceFeeder:
    """Iterate ``loader`` with batches already on ``device`` (see module docstring).

    ``len()`` and ``.dataset`` are forwarded so the feeder can stand in for the loader in
    ``train_epoch`` / ``test_epoch`` / ``get_latent``."""

    def __init__(self, loader, device, depth=3):
        self.loader = loader
        self.device = torch.device(device)
        if self.device.type == "cuda" and self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.depth = max(2, int(depth))

    def __len__(self):
        return len(self.loader)

    @property
    def dataset(self):
        return self.loader.dataset

    def __iter__(self):
        if self.device.type != "cuda":
            for batch in self.loader:
                yield batch.to(device=self.device, dtype=torch.float32)
            return
        yield from self._iter_cuda()

    def _iter_cuda(self):
        copy_stream = torch.cuda.Stream(self.device)
        slots = []

        def stage(batch, k):
            """enqueue the H2D copy of the k-th batch on the copy stream; returns its slot"""
            batch = torch.as_tensor(batch)
            if batch.dim() != 3:
                raise ValueError("expected [batch,128,128] spectrograms, got %s" % (tuple(batch.shape),))
            n = batch.shape[0]
            if len(slots) < self.depth:
                slots.append(_Slot(tuple(batch.shape), self.device))
            slot = slots[k % self.depth]
            slot.released.synchronize()                      # the consumer's kernels on this slot have finished
            if n > slot.dev.shape[0]:                        # a later batch is larger than the first one
                slot.dev = torch.empty(tuple(batch.shape), dtype=torch.float32, device=self.device)
            with torch.cuda.stream(copy_stream):
                slot.dev[:n].copy_(batch, non_blocking=True)
                slot.ready.record(copy_stream)
            slot.n = n
            return slot

        it = iter(self.loader)
        k = 0
        try:
            cur = stage(next(it), k)
        except StopIteration:
            return
        while cur is not None:
            k += 1
            try:
                nxt = stage(next(it), k)                     # in flight while the caller works on ``cur``
            except StopIteration:
                nxt = None
            compute = torch.cuda.current_stream(self.device)
            compute.wait_event(cur.ready)
            yield cur.dev[:cur.n]
            # the caller has enqueued its work on this batch by the time it asks for the next one
            cur.released.record(torch.cuda.current_stream(self.device))
            cur = nxt
