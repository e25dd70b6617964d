"""Test-only: two ranks that share ONE GPU take turns on it.

The gpurun boxes have one GPU, so the two-rank `-m gpu` tests put both ranks on cuda:0 and run the collectives over gloo.
Two PROCESSES with kernels on the chip at the same time are not bit-reproducible on this pool (profiles/NOTES.md items 43
and 44: one repeat in ten, a few long-lived register accumulators of one kernel; the stand-alone reproducer and its verdict
are in tools/lab/two_proc_repro.hip), while the tests assert bit-identity between two ways of running the same step.  So a
rank holds a lock shared by the ranks of the test while it computes and gives it up inside collectives, which are staged
through the host here (gloo then only ever sees CPU tensors, and copies happen while the rank holds its turn).

This lives in tests/ on purpose: `ava_amd.dist` carries no lock and no host staging; `take_turns` replaces the module's
three collective primitives (`_all_reduce`, `_all_gather_into`, `_broadcast`) in the worker process.  The product's own
asynchronous path (work handles of `async_op=True` collectives consumed bucket by bucket) is exercised by
tests/test_cpu_dist.py (CPU tensors over gloo) and by the concurrent variant in tests/test_gpu_dist.py."""
import contextlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_turn = None


class _Done:
    """Handle of a collective that is complete already."""
    def wait(self):
        return True

    def is_completed(self):
        return True


@contextlib.contextmanager
def off_gpu():
    """Around a blocking host-side collective: this rank's kernels are done, the other rank may have the GPU meanwhile."""
    import torch
    lock = _turn
    if lock is None:
        yield
        return
    torch.cuda.synchronize()
    lock.release()
    try:
        yield
    finally:
        lock.acquire()


def take_turns(turn):
    """First thing in a worker: this rank holds the lock `turn` from init_process_group to destroy_process_group, except
    inside the collectives of ava_amd.dist, which are replaced by host-staged blocking forms.  `turn=None`: nothing is
    changed (the ranks run concurrently on the product's own asynchronous path)."""
    global _turn
    if turn is None:
        return
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as td
    from ava_amd import dist as adist

    def all_reduce(t, op, async_op=False):
        if t.is_cuda:
            host = t.detach().cpu()                    # waits for the kernels that produce t (current stream)
            with off_gpu():
                td.all_reduce(host, op=op)
            t.copy_(host)
        else:
            with off_gpu():
                td.all_reduce(t, op=op)
        return _Done() if async_op else None

    def all_gather_into(bucket, mine):
        host = mine.detach().cpu()
        parts = [torch.empty_like(host) for _ in range(td.get_world_size())]
        with off_gpu():
            td.all_gather(parts, host)
        bucket.copy_(torch.cat(parts))
        return _Done()

    def broadcast(t, src):
        host = t.detach().cpu()
        with off_gpu():
            td.broadcast(host, src=src)
        t.copy_(host)

    adist._all_reduce, adist._all_gather_into, adist._broadcast = all_reduce, all_gather_into, broadcast
    init, destroy = td.init_process_group, td.destroy_process_group

    def init_then_take(*a, **k):
        global _turn
        r = init(*a, **k)
        _turn = turn
        turn.acquire()
        return r

    def give_then_destroy(*a, **k):
        global _turn
        if _turn is not None:
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            _turn = None
            turn.release()
        return destroy(*a, **k)
    td.init_process_group, td.destroy_process_group = init_then_take, give_then_destroy
