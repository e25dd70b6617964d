"""Data-parallel glue: one process per GPU, gradient SUM all-reduce over RCCL/xGMI.

The reference is single-device (``ava/models/vae.py:112-115``); pure data parallelism is
this build's extension.  Samples are independent except for the
BatchNorm batch statistics (kept per rank) and the parameter update, so the only exchange
per step is ONE all-reduce of the flat fp32 gradient arena (69.7 MB at z=32).  It is a SUM,
not a mean, because the reference loss is a sum over the batch (vae.py:316-323).

Nothing here is needed (or touched) when torch.distributed is not initialised.
"""
import math

import torch
import torch.distributed as td

from .layout import X_DIM


def active():
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def rank():
    return td.get_rank() if (td.is_available() and td.is_initialized()) else 0


def world_size():
    return td.get_world_size() if (td.is_available() and td.is_initialized()) else 1


def cu_reserve():
    """CUs every persistent conv / GEMM grid leaves free while a collective is in flight, so that the collective's own
    persistent workgroups (RCCL: one per channel, resident for the length of an all-reduce) find wave slots instead of
    pushing part of a statically partitioned conv grid into a second wave (``include/ava_hip.h: ava_model_set_cu_reserve``).
    Default 0: the reservation costs +0.27 % of the reserved launches' time per CU unconditionally and has only been
    sized against a stand-in for RCCL (``ava_occupy_cus``), never against RCCL itself; it stays off
    until an N >= 2 RCCL run shows a net win.  ``bench.py --gpus N`` measures exactly that (``dist.reserve_sweep``: 0, 16,
    32) and runs its timed region on the best; ``AVA_CU_RESERVE`` sets it by hand."""
    import os
    try:
        return max(0, min(128, int(os.environ.get("AVA_CU_RESERVE", "0"))))
    except ValueError:
        return 0


def apply_cu_reserve(lib, handle, collectives_in_flight=True):
    """Size the persistent grids of the model ``handle`` for the kernels about to be launched (the C entry points read the
    model's setting once, at their start).  The reservation is only worth its price while a collective IS in flight:
    ``VAE._backward_kernels`` applies it to the backward parts that run beside the gradient buckets' all-reduces (parts 1..3)
    and launches everything else -- the forward, backward part 0 -- on grids sized for the whole chip.  The per-bucket Adam
    launches that ``FlatAdam.step`` issues while later buckets are still on the wire are deliberately NOT covered: the
    reserve exists for launches that are ONE resident wave over a static tile partition (a workgroup that finds no slot
    becomes a second wave), whereas ``adam_flat_kernel`` is a grid-stride launch of 4096 small blocks that simply fill
    the slots a collective leaves free (``bench.py`` says so under ``dist.cu_reserve_applies_to``).  No-op in a
    single-process run and with the default reserve of 0."""
    if handle is None or not active():
        return 0
    want = cu_reserve() if collectives_in_flight else 0
    if want == 0 and cu_reserve() == 0:
        return 0                                   # nothing was ever reserved: the model keeps following the process default
    lib.ava_model_set_cu_reserve(handle, want)
    return want


def sharded_adam():
    """Optional form of the exchange (``AVA_DP_SHARDED_ADAM=1``): every gradient bucket is reduce-scattered instead of
    all-reduced, each rank runs Adam on its 1/N slice of the bucket only, and the updated parameters are all-gathered.
    Same bytes on the wire (a ring all-reduce IS a reduce-scatter followed by an all-gather), Adam's 488 MB of HBM
    traffic per step divided by N; exp_avg / exp_avg_sq are complete on a rank only for its own slices until
    ``gather_adam_state`` (called before checkpoints)."""
    import os
    return active() and os.environ.get("AVA_DP_SHARDED_ADAM", "0") not in ("", "0")


def shard_of(offset, count):
    """(offset, count) of this rank's slice of a bucket, or None when the bucket does not split into N 16-byte-aligned parts"""
    n = world_size()
    if count % (4 * n) != 0:
        return None
    part = count // n
    return offset + rank() * part, part


# ---- the three collective primitives everything below goes through -------------------------------------------------------
# (one seam: tests that put two ranks on ONE GPU replace these three with host-staged, turn-taking forms of their own --
# tests/gpu_util.py: take_turns -- the product module carries no test code)
def _all_reduce(t, op, async_op=False):
    """td.all_reduce; with ``async_op`` the work handle (RCCL: ordered on the compute stream by ``wait()``)."""
    return td.all_reduce(t, op=op, async_op=async_op)


def _all_gather_into(bucket, mine):
    """Every rank's ``mine`` (equal sizes) into ``bucket``, asynchronously; returns the work handle."""
    try:
        return td.all_gather_into_tensor(bucket, mine, async_op=True)
    except (RuntimeError, NotImplementedError):
        return td.all_gather(list(bucket.chunk(world_size())), mine, async_op=True)


def _broadcast(t, src):
    td.broadcast(t, src=src)


def _all_reduce_async(t, op):
    return _all_reduce(t, op, async_op=True)


def reduce_scatter_bucket_async(flat_grads, offset, count):
    """SUM of one gradient bucket over the ranks, delivered to its owner slices only (in place: rank r's slice of the
    bucket receives the reduced values).  RCCL: reduce_scatter_tensor; gloo has no reduce-scatter: all-reduce."""
    bucket = flat_grads[offset:offset + count]
    sh = shard_of(offset, count)
    if sh is None or td.get_backend() != "nccl":
        return _all_reduce_async(bucket, td.ReduceOp.SUM)
    return td.reduce_scatter_tensor(flat_grads[sh[0]:sh[0] + sh[1]], bucket, op=td.ReduceOp.SUM, async_op=True)


def all_gather_bucket_async(flat, offset, count):
    """Every rank's slice of flat[offset:offset+count] to every rank (in place)."""
    sh = shard_of(offset, count)
    return _all_gather_into(flat[offset:offset + count], flat[sh[0]:sh[0] + sh[1]].clone())


def allreduce_gradients(flat_grads):
    """In-place SUM over ranks of the flat gradient arena (backend 'nccl' is RCCL on ROCm)."""
    if active():
        _all_reduce(flat_grads, td.ReduceOp.SUM)
    return flat_grads


def allreduce_gradients_async(flat_slice):
    """Enqueue the SUM all-reduce of one gradient bucket on the communication stream (it starts once the
    kernels already enqueued on the current stream have produced the bucket) and return the work handle;
    compute enqueued afterwards overlaps with it."""
    if not active():
        return None
    return _all_reduce_async(flat_slice, td.ReduceOp.SUM)


def wait_all(handles):
    """Make the current stream (and, for CPU backends, the host) wait for the given all-reduces."""
    waited = False
    for h in handles:
        if h is not None:
            h.wait()
            waited = True
    # gloo stages device tensors through the host on streams of its own; RCCL's collectives are ordered on the compute
    # stream by wait().  For the CPU-backed test backend make the hand-back explicit.
    if waited and td.get_backend() != "nccl" and torch.cuda.is_available():
        torch.cuda.synchronize()


def allreduce_max_(t):
    """In-place MAX over ranks (status words)."""
    if active():
        _all_reduce(t, td.ReduceOp.MAX)
    return t


def allreduce_max_async(t):
    """MAX over ranks enqueued like a gradient bucket (status words); returns the work handle."""
    if not active():
        return None
    return _all_reduce_async(t, td.ReduceOp.MAX)


def broadcast_parameters(model, src=0):
    """Identical weights / BatchNorm buffers / Adam state on every rank."""
    if active():
        for t in (model._params, model._exp_avg, model._exp_avg_sq, model._bn_running, model._bn_batches):
            _broadcast(t, src)


def per_call_constants(z_dim, model_precision, x_dim=X_DIM):
    """The two constants the reference adds once per forward call (vae.py:316,318)."""
    return 0.5 * z_dim * math.log(2 * math.pi) + 0.5 * x_dim * math.log(2 * math.pi / model_precision)


def global_loss(local_sum, z_dim, model_precision, num_batches, x_dim=X_DIM):
    """Loss of the global batch from per-rank sums: every rank's forward added the per-call
    constants once per batch, a single-process run adds them once per *global* batch:
    ``L = sum_r L_r - (N-1) * num_batches * (c1 + c2)``."""
    if not active():
        return float(local_sum.item())
    t = local_sum.detach().clone().double().reshape(1)
    _all_reduce(t, td.ReduceOp.SUM)
    n = world_size()
    return float(t.item()) - (n - 1) * num_batches * per_call_constants(z_dim, model_precision, x_dim)


def check_equal_batches(num_batches, what="epoch"):
    """Every rank must run the SAME number of steps per epoch: each step holds collectives (the gradient buckets, the
    status word's MAX), and ``global_loss`` removes ``(N-1) * num_batches`` copies of the per-call constants -- with ragged
    shards the ranks would deadlock in mismatched collectives or report different losses.  One MAX all-reduce of
    ``[n, -n]`` before the first step; raises ``ValueError`` on every rank when the counts differ (shard with
    ``DistributedSampler(drop_last=True)`` or pad the shorter shard)."""
    if not active():
        return num_batches
    t = torch.tensor([float(num_batches), -float(num_batches)], dtype=torch.float64)
    if td.get_backend() == "nccl":
        t = t.cuda()
    _all_reduce(t, td.ReduceOp.MAX)
    hi, lo = int(round(float(t[0].item()))), int(round(-float(t[1].item())))
    if hi != lo:
        raise ValueError("data parallel %s: the ranks' loaders have different numbers of batches (%d..%d, this rank %d); "
                         "every rank must run the same number of steps" % (what, lo, hi, num_batches))
    return num_batches


def global_dataset_len(local_len):
    """len(loader.dataset) summed over ranks (each rank iterates its own shard)."""
    if not active():
        return local_len
    t = torch.tensor([float(local_len)], dtype=torch.float64)
    if td.get_backend() == "nccl":
        t = t.cuda()
    _all_reduce(t, td.ReduceOp.SUM)
    return int(round(float(t.item())))
