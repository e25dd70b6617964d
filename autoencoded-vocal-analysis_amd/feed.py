"""Host -> device feeding for the epoch loops.

The reference converts every item to float32 on the CPU (``numpy_to_tensor``, ``ava/models/utils.py:444-446``,
applied per item by ``SyllableDataset.__getitem__``, ``ava/models/vae_dataset.py:125-145``), collates, and moves each
batch with a synchronous ``data.to(self.device)`` on the compute stream (``ava/models/vae.py:349, 374, 540``): a 4M-element
CPU pass plus a 16 MiB copy in series with every 2 ms step.  Two pieces replace that:

``DeviceFeeder``  wraps any iterable of ``[B,H,W]`` CPU batches and yields device-resident fp32 tensors,
  software-pipelined by one batch on a dedicated copy stream:

  * the batch crosses PCIe in the dtype the loader hands over -- float32 as is; float64 / uint8 / float16 / bfloat16
    as RAW bytes into a device staging slot, followed by the device-side cast ``ava_cast_to_f32`` (same rounding as
    torch's ``.type(torch.FloatTensor)``) on the copy stream.  No element is converted on the CPU;
  * batches in page-locked memory (``DataLoader(pin_memory=True)`` or ``PinnedBatchLoader`` below) go as one
    asynchronous DMA; pageable batches go through the runtime's own staging (the call then blocks the host for
    ~0.5 ms, hidden as long as the host stays ahead of the GPU);
  * the consumer's stream waits on the copy's event (no host synchronisation); a device slot is only overwritten
    after the HOST has seen the consumer's kernels on it finish (``released.synchronize()``), which also bounds how
    far the host runs ahead (``depth - 1`` steps).

``PinnedBatchLoader``  a build-owned page-locked ring the batches are COLLATED INTO: items of a dataset (numpy arrays
  or tensors of any supported dtype, e.g. the float64 spectrograms the preprocessing step writes) are written straight
  into slot ``k % depth`` of the ring -- the one copy collation needs anyway -- and the feeder DMAs the slot without any
  further staging.  A slot is reused only after the DMA that read it has completed (per-slot event).

Measured dead ends (tools/feed_probe.py): copying finished pageable batches into own pinned buffers costs the caller a
3 ms memcpy per batch (longer than the step) -- hence the ring is filled at collation time instead; a helper thread
for it slows the ~100 ctypes launches of a step through GIL hand-overs (2.1 -> 3.5 ms); making the COPY stream wait
for the release event turns the enqueue into a 6 ms host-side wait.

Values are bit-identical to ``x.to(device).type(torch.float32)``; ragged last batches are handled.  On a non-CUDA
device the loader is passed through.
"""
import numpy as np
import torch

from . import _lib

__all__ = ["DeviceFeeder", "PinnedBatchLoader", "cast_to_f32"]

# torch dtype -> (ava_cast_to_f32 code, bytes per element)
_CAST = {torch.float32: (0, 4), torch.float64: (1, 8), torch.uint8: (2, 1), torch.float16: (3, 2), torch.bfloat16: (4, 2)}

# page-locked ring slots by data pointer: lets the feeder tell the ring when the DMA out of a slot is done
_RING_SLOTS = {}


def cast_to_f32(src, out=None):
    """Device-side ``numpy_to_tensor``: ``src`` (device tensor, float64 / uint8 / float16 / bfloat16 / float32) ->
    float32 tensor of the same shape, on the current stream."""
    if src.dtype not in _CAST:
        raise TypeError("unsupported loader dtype %s" % src.dtype)
    src = src.contiguous()
    if out is None:
        out = torch.empty(src.shape, dtype=torch.float32, device=src.device)
    code, _ = _CAST[src.dtype]
    _lib.check(_lib.load().ava_cast_to_f32(src.data_ptr(), code, src.numel(), out.data_ptr(), _lib.stream()),
               "ava_cast_to_f32")
    return out


class _Slot:
    __slots__ = ("dev", "raw", "ready", "released", "n")

    def __init__(self, shape, device):
        self.dev = torch.empty(shape, dtype=torch.float32, device=device)
        self.raw = None                        # uint8 staging for non-fp32 batches (allocated on first use)
        self.ready = torch.cuda.Event()        # H2D copy (+ cast) of this slot finished (recorded on the copy stream)
        self.released = torch.cuda.Event()     # consumer's kernels that read ``dev`` are done (compute stream)
        self.n = 0


class DeviceFeeder:
    """Iterate ``loader`` with batches already on ``device`` as fp32 (see module docstring).

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
        lib = _lib.load()

        def stage(batch, k):
            """enqueue the H2D copy (and, for non-fp32 data, the device cast) of the k-th batch on the copy stream"""
            batch = torch.as_tensor(batch)
            if batch.dim() != 3:
                raise ValueError("expected [batch,H,W] spectrograms, got %s" % (tuple(batch.shape),))
            n = batch.shape[0]
            if len(slots) < self.depth:
                slots.append(_Slot(tuple(batch.shape), self.device))
            slot = slots[k % self.depth]
            slot.released.synchronize()                      # the consumer's kernels on this slot have finished
            if n > slot.dev.shape[0] or tuple(batch.shape[1:]) != tuple(slot.dev.shape[1:]):
                slot.dev = torch.empty(tuple(batch.shape), dtype=torch.float32, device=self.device)
            ring_slot = _RING_SLOTS.get(batch.data_ptr()) if batch.device.type == "cpu" else None
            if batch.device.type != "cpu":                   # produced on the device by the caller's stream: order the copy
                copy_stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(copy_stream):
                if batch.dtype == torch.float32 or batch.dtype not in _CAST or batch.device.type != "cpu":
                    slot.dev[:n].copy_(batch, non_blocking=True)             # fp32 (or an exotic dtype: torch converts)
                else:
                    code, esz = _CAST[batch.dtype]
                    src = batch.contiguous()
                    nbytes = src.numel() * esz
                    if slot.raw is None or slot.raw.numel() < nbytes:
                        slot.raw = torch.empty(nbytes + 64, dtype=torch.uint8, device=self.device)
                    raw = slot.raw[:nbytes]
                    raw.copy_(src.view(-1).view(torch.uint8), non_blocking=True)      # bytes as stored, no CPU conversion
                    _lib.check(lib.ava_cast_to_f32(raw.data_ptr(), code, src.numel(), slot.dev.data_ptr(),
                                                   copy_stream.cuda_stream), "ava_cast_to_f32")
                slot.ready.record(copy_stream)
                if ring_slot is not None:
                    ring_slot.copied.record(copy_stream)     # the page-locked ring may refill this slot after the DMA
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


class _RingSlot:
    __slots__ = ("host", "copied", "busy")

    def __init__(self, shape, dtype):
        self.host = torch.empty(shape, dtype=dtype)
        if torch.cuda.is_available():          # without a GPU (host-logic tests) the ring is ordinary memory
            self.host = self.host.pin_memory()
        self.copied = torch.cuda.Event() if torch.cuda.is_available() else None
        self.busy = False


class PinnedBatchLoader:
    """Batches collated straight into a build-owned page-locked ring (replaces ``DataLoader`` +
    ``SyllableDataset.__getitem__`` + ``numpy_to_tensor`` of ``ava/models/vae_dataset.py:62-145`` for in-memory or
    memory-mapped datasets).  Items are written into the ring slot in their OWN dtype -- no float32 conversion on the
    CPU -- and the batch is handed out as a view of the slot, which ``DeviceFeeder`` DMAs asynchronously and casts on
    the device.

    ``dataset``: a C-contiguous ``[N,H,W]`` numpy array / memmap (fast path: every batch is ONE native gather into the
    slot, ``ava_host_gather_rows`` on ``workers`` host threads with the GIL released) or any indexable whose ``dataset[i]`` is an
    ``[H,W]`` numpy array or tensor (per-item copies).  With ``prefetch=True`` a producer thread fills the ring one or
    two batches ahead of the consumer, so collation overlaps the consumer's kernel launches.

    Same iteration contract as the reference's loaders: ``len()``, ``.dataset``, ``.batch_size``, optional shuffling
    (``torch.randperm`` per epoch, like ``RandomSampler``), ragged last batch kept.  A slot is refilled only after the
    DMA that read it has completed (``DeviceFeeder`` records a per-slot event)."""

    def __init__(self, dataset, batch_size=64, shuffle=False, depth=4, generator=None, workers=4, prefetch=True):
        self.dataset = dataset
        self.batch_size = int(batch_size)
        self.shuffle = bool(shuffle)
        self.depth = max(3, int(depth))
        self.generator = generator
        self.workers = max(1, int(workers))
        self.prefetch = bool(prefetch)
        self._slots = None
        self._idx = None

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def _ring(self, item):
        if self._slots is None:
            item = torch.as_tensor(item)
            shape = (self.batch_size,) + tuple(item.shape)
            self._slots = [_RingSlot(shape, item.dtype) for _ in range(self.depth)]
            for s in self._slots:
                _RING_SLOTS[s.host.data_ptr()] = s
        return self._slots

    def __del__(self):
        ring = _RING_SLOTS                     # None once the interpreter is tearing the module down
        if ring is None:
            return
        for s in (self._slots or []):
            ring.pop(s.host.data_ptr(), None)

    def _fill(self, slot, idx):
        """write items ``idx`` of the dataset into the slot (its own dtype); returns the batch view"""
        if slot.busy and slot.copied is not None:
            slot.copied.synchronize()                        # the DMA that read this slot `depth` batches ago is done
        host = slot.host
        n = len(idx)
        if isinstance(self.dataset, np.ndarray) and self.dataset.flags["C_CONTIGUOUS"]:
            # one native call per batch (csrc/feed.hip: ava_host_gather_rows, std::threads; the GIL is released)
            data = self.dataset
            row_bytes = data[0].nbytes
            contiguous = n > 0 and idx[-1] - idx[0] == n - 1 and all(b - a == 1 for a, b in zip(idx, idx[1:]))
            if contiguous:
                ip, first = None, int(idx[0])
            else:
                self._idx = np.ascontiguousarray(idx, dtype=np.int64)
                ip, first = self._idx.ctypes.data, 0
            _lib.check(_lib.load().ava_host_gather_rows(host.data_ptr(), data.ctypes.data, ip, first, n, row_bytes,
                                                        self.workers), "ava_host_gather_rows")
        else:
            for j, i in enumerate(idx):
                host[j].copy_(torch.as_tensor(self.dataset[i]))
        slot.busy = True
        return host[:n]

    def _batches(self):
        n = len(self.dataset)
        order = torch.randperm(n, generator=self.generator).tolist() if self.shuffle else list(range(n))
        if n == 0:
            return
        slots = self._ring(self.dataset[0])
        for k, start in enumerate(range(0, n, self.batch_size)):
            yield self._fill(slots[k % self.depth], order[start:start + self.batch_size])

    def __iter__(self):
        if not self.prefetch:
            yield from self._batches()
            return
        # producer thread: at most depth - 2 filled slots wait in the queue, one more is in the consumer's hands and one
        # may still be the source of an in-flight DMA, so the producer never overwrites a slot that is in use
        import queue
        import threading
        q = queue.Queue(maxsize=max(1, self.depth - 2))
        stop = threading.Event()
        end = object()

        def put(item):
            """stop-aware put (batches, the end marker and exceptions alike): never blocks past a consumer that left"""
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    continue
            return False

        def produce():
            try:
                for b in self._batches():
                    if not put(b):
                        return
                put(end)
            except BaseException as e:                       # surface loader errors in the consumer
                put(e)

        # one producer per loader at a time: a previous iteration's thread (consumer left early, e.g. on an exception in the
        # step) may still be inside _fill(); it shares the ring slots, so it must have finished before a new one starts
        prev = getattr(self, "_producer", None)
        if prev is not None and prev.is_alive():
            prev.join(timeout=30)
            if prev.is_alive():
                raise RuntimeError("PinnedBatchLoader: the producer thread of the previous iteration is still running")
        th = threading.Thread(target=produce, daemon=True)
        self._producer = th
        th.start()
        try:
            while True:
                item = q.get()
                if item is end:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield item
        finally:
            stop.set()
            th.join(timeout=5)
