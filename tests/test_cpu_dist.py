"""World-size-2 tests of the data-parallel glue on CPU (gloo): gradient SUM all-reduce of the flat arena
and the global-loss assembly, fed with per-shard results of the CPU oracle and pinned by the
reference-generated two-shard golden (tests/golden/ddp2.npz)."""
import os
import sys

import numpy as np
import torch
import torch.multiprocessing as mp

from conftest import load_golden, ROOT


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    from ava_amd import dist as adist, synthetic as syn, layout
    from oracle import vae_oracle as O
    torch.set_num_threads(2)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert adist.active() and adist.rank() == rank and adist.world_size() == world
        z, B = 32, 8
        x = syn.spectrograms(B * world)
        ew, ed = syn.noise(B * world, z)
        sl = slice(B * rank, B * rank + B)
        P = O.to_params(syn.fixture_parameters(z), requires_grad=True)
        out = O.forward(P, torch.from_numpy(x[sl]), torch.from_numpy(ew[sl]), torch.from_numpy(ed[sl]), None, True)
        out["loss"].backward()
        offs, total = layout.arena_offsets(z)
        flat = torch.zeros(total)
        for s in layout.param_specs(z):
            flat[offs[s.name]:offs[s.name] + s.numel] = P[s.name].grad.reshape(-1)
        # four buckets (tail, fc1's weight, the rest of the middle, head of the arena), asynchronously, exactly like
        # VAE._backward_device does under data parallelism (ava_grad_bucket: fc8 + decoder, fc1.weight, fc1.bias..fc7, encoder)
        s8, s1, s1b = offs["fc8.weight"], offs["fc1.weight"], offs["fc1.bias"]
        pending = [adist.allreduce_gradients_async(flat[s8:]), adist.allreduce_gradients_async(flat[s1:s1b]),
                   adist.allreduce_gradients_async(flat[s1b:s8]), adist.allreduce_gradients_async(flat[:s1])]
        adist.wait_all(pending)                               # SUM, not mean (the loss is a batch sum)
        norms = {s.name: float(flat[offs[s.name]:offs[s.name] + s.numel].double().norm()) for s in layout.param_specs(z)}
        gl = adist.global_loss(out["loss"].detach().double(), z, 10.0, 1)
        n = adist.global_dataset_len(B)
        # the "d not positive" status word travels with the buckets: MAX over ranks, asynchronously
        # sharded-optimizer helpers: this rank's slice of a bucket, and the in-place all-gather of the slices
        assert adist.shard_of(64, 128) == (64 + rank * 64, 64) and adist.shard_of(0, 6) is None
        buf = torch.zeros(8 + 128)
        o, c = adist.shard_of(8, 128)
        buf[o:o + c] = float(rank + 1)
        adist.wait_all([adist.all_gather_bucket_async(buf, 8, 128)])
        assert buf[:8].abs().sum() == 0 and bool((buf[8:72] == 1).all()) and bool((buf[72:] == 2).all())
        st = torch.tensor([1 if rank == 1 else 0, 0], dtype=torch.int32)
        adist.wait_all([adist.allreduce_max_async(st)])
        assert st.tolist() == [1, 0]
        # ragged shards are refused on EVERY rank before the first step: equal counts pass through
        assert adist.check_equal_batches(5) == 5
        try:
            adist.check_equal_batches(5 + rank, "test")
            ragged = False
        except ValueError as e:
            ragged = "5..6" in str(e)
        assert ragged
        q.put((rank, float(out["loss"]), norms, gl, n))
    finally:
        td.destroy_process_group()


def test_two_rank_gradient_sum_and_global_loss():
    G = load_golden("ddp2.npz")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort()
    from ava_amd import dist as adist
    c = adist.per_call_constants(32, 10.0)
    assert abs(c - (29.406 - 3806.888)) < 1e-2             # SURVEY Appendix B constants at z=32, prec=10
    for rank, loss, norms, gl, n in res:
        assert abs(loss - float(G["shard%d.loss" % rank])) / abs(loss) < 1e-5
        assert n == 16
        # every rank added the per-call constants once; a single-process step adds them once per global batch
        want = float(G["shard0.loss"]) + float(G["shard1.loss"]) - c
        assert abs(gl - want) / abs(want) < 1e-5
        for name, v in norms.items():
            sens = name.split(".")[0] in ("conv1", "bn1")
            ref = float(G["gradnorm." + name])
            scale = max(ref, float(G["gradnorm.conv1.bias"]) if sens else 0.0)
            assert abs(v - ref) < (2e-2 if sens else 1e-3) * scale, name
    assert res[0][2] == res[1][2]                            # both ranks hold the identical reduced gradient


def test_single_process_is_passthrough():
    from ava_amd import dist as adist
    t = torch.arange(4.0)
    assert adist.allreduce_gradients(t) is t and not adist.active()
    assert adist.rank() == 0 and adist.world_size() == 1 and adist.global_dataset_len(7) == 7
    assert adist.global_loss(torch.tensor(5.0), 32, 10.0, 3) == 5.0


def test_cu_reserve_is_applied_only_where_a_collective_is_in_flight(monkeypatch):
    """dist.apply_cu_reserve: OFF by default (never validated against RCCL: ADVICE r3); when AVA_CU_RESERVE asks for r, the
    model's grids are sized for 256 - r CUs only for the launches that run beside a collective (backward parts 1..3); the
    forward and backward part 0 get the whole chip; nothing is touched in a single-process run or without a model handle."""
    from ava_amd import dist as adist

    class Lib:
        def __init__(self): self.calls = []
        def ava_model_set_cu_reserve(self, h, v): self.calls.append((h, v)); return 0

    lib = Lib()
    monkeypatch.setattr(adist, "active", lambda: False)
    monkeypatch.setenv("AVA_CU_RESERVE", "32")
    assert adist.apply_cu_reserve(lib, 7, True) == 0 and adist.apply_cu_reserve(lib, 7, False) == 0 and lib.calls == []
    monkeypatch.setattr(adist, "active", lambda: True)
    monkeypatch.delenv("AVA_CU_RESERVE", raising=False)
    assert adist.cu_reserve() == 0
    assert [adist.apply_cu_reserve(lib, 7, f) for f in (False, True, True, False)] == [0, 0, 0, 0] and lib.calls == []
    monkeypatch.setenv("AVA_CU_RESERVE", "32")
    seq = [adist.apply_cu_reserve(lib, 7, f) for f in (False, False, True, True, True, False)]     # forward, parts 0..3, after
    assert seq == [0, 0, 32, 32, 32, 0] and [v for _, v in lib.calls] == seq and all(h == 7 for h, _ in lib.calls)
    assert adist.apply_cu_reserve(lib, None, True) == 0
    monkeypatch.setenv("AVA_CU_RESERVE", "16")
    assert adist.apply_cu_reserve(lib, 7, True) == 16 and lib.calls[-1] == (7, 16)


# ---------------------------------------------------------------------------------------------------------------------------
# The deferred path with GENUINELY asynchronous handles: `VAE._backward_device(defer_comm=True)`
# leaves one `async_op=True` work handle per gradient bucket (+ the status word's MAX) in flight, `FlatAdam.step` waits for
# and updates ONE bucket at a time.  No GPU here, so the native library is replaced by a stand-in with the same entry points
# that does the arithmetic on the CPU arenas in torch (test-side restatement of torch/optim/adam.py:414-547); what is under
# test is the product's Python bookkeeping over real gloo handles: issue order, which handle guards which launch, per-bucket
# Adam == one flat Adam, nothing left pending, `zero_grad()` / a new forward never leave a collective in flight.
class _CpuLib:
    """ava_backward_part / ava_adam_step_range / ava_adam_step on CPU tensors; every call is logged."""

    def __init__(self, model, buckets, local_grad):
        self.m, self.buckets, self.local_grad, self.log = model, buckets, local_grad, []

    def ava_backward_num_parts(self):
        return len(self.buckets)

    def ava_backward_part(self, handle, xptr, B, part, stream):
        o, c = self.buckets[part]
        self.m._grads[o:o + c] = self.local_grad[o:o + c]
        self.log.append(("part", part))
        return 0

    def _adam(self, o, c, lr, b1, b2, eps, step):
        m = self.m
        if int(m._status[0]) != 0:                       # the device-side guard: skipped launches are counted
            m._status[1] += 1
            return 0
        g = m._grads[o:o + c]
        m._exp_avg[o:o + c].mul_(b1).add_(g, alpha=1 - b1)
        m._exp_avg_sq[o:o + c].mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
        denom = (m._exp_avg_sq[o:o + c].sqrt() / (bc2 ** 0.5)).add_(eps)
        m._params[o:o + c].addcdiv_(m._exp_avg[o:o + c], denom, value=-lr / bc1)
        return 0

    def ava_adam_step_range(self, handle, o, c, lr, b1, b2, eps, step, stream):
        self.log.append(("adam", (o, c)))
        return self._adam(o, c, lr, b1, b2, eps, step)

    def ava_adam_step(self, handle, lr, b1, b2, eps, step, stream):
        self.log.append(("adam", "flat"))
        return self._adam(0, self.m._params.numel(), lr, b1, b2, eps, step)

    def ava_model_set_cu_reserve(self, handle, v):
        return 0

    def ava_model_destroy(self, handle):
        return 0


class _SpyHandle:
    """wraps a gloo work handle: records when it is waited for"""

    def __init__(self, h, tag, log):
        self.h, self.tag, self.log = h, tag, log

    def wait(self):
        self.log.append(("wait", self.tag))
        return self.h.wait()


def _async_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, ROOT)
    import torch.distributed as td
    from ava_amd import dist as adist, _lib, layout
    from ava_amd.vae import VAE
    torch.set_num_threads(2)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z = 32
        offs, total = layout.arena_offsets(z)
        s8, s1, s1b = offs["fc8.weight"], offs["fc1.weight"], offs["fc1.bias"]
        buckets = [(s8, total - s8), (s1, s1b - s1), (s1b, s8 - s1b), (0, s1)]           # ava_grad_bucket's four, in part order
        res = {}
        real_load, real_stream = _lib.load, _lib.stream
        for mode in ("deferred", "flat"):
            torch.manual_seed(5)
            _lib.load, _lib.stream = real_load, real_stream
            model = VAE(z_dim=z, device_name="cpu")
            model._handle = 1                                  # a "native model" exists: FlatAdam takes its data-parallel path
            model._bucket_cache = (model._handle, buckets)
            adist.broadcast_parameters(model)
            local = torch.randn(total, generator=torch.Generator().manual_seed(100 + rank))
            lib = _CpuLib(model, buckets, local)
            _lib.load, _lib.stream = (lambda: lib), (lambda: None)
            real = adist._all_reduce

            def spying(t, op, async_op=False, _real=real, _lib_=lib):
                h = _real(t, op, async_op=async_op)
                if not async_op:
                    return h
                tag = "status" if t.dtype == torch.int32 else (t.data_ptr() - model._grads.data_ptr()) // 4
                _lib_.log.append(("issue", tag))
                return _SpyHandle(h, tag, _lib_.log)
            adist._all_reduce = spying
            x = torch.zeros(8, 128, 128)
            for step in (1, 2):
                model.optimizer.zero_grad()
                lib.log.clear()
                if mode == "deferred":
                    model._backward_device(x, defer_comm=True)
                    assert model._pending_comm is not None and len(model._pending_comm) == 1 + len(buckets)
                    assert [e for e in lib.log if e[0] == "wait"] == []          # nothing waited for yet: all in flight
                    model.optimizer.step()
                    assert not model._pending_comm
                    # issue order: part k, then its bucket (the status word behind part 0)
                    want_issue = [("part", 0), ("issue", "status"), ("issue", buckets[0][0]), ("part", 1), ("issue", buckets[1][0]),
                                  ("part", 2), ("issue", buckets[2][0]), ("part", 3), ("issue", buckets[3][0])]
                    assert lib.log[:9] == want_issue, lib.log[:9]
                    # consumption: the status word first, then for every bucket ITS wait directly in front of ITS update
                    tail = lib.log[9:]
                    assert tail[0] == ("wait", "status"), tail
                    assert tail[1:] == [e for o, c in buckets for e in (("wait", o), ("adam", (o, c)))], tail
                else:
                    model._backward_device(x)                                    # complete on return
                    assert not model._pending_comm
                    assert len([e for e in lib.log if e[0] == "wait"]) == 1 + len(buckets)
                    _lib.check(lib.ava_adam_step(model._handle, 1e-3, 0.9, 0.999, 1e-8, step, None), "adam")
            adist._all_reduce = real
            res[mode] = (model._grads.clone(), model._params.clone(), model._exp_avg.clone(), model._exp_avg_sq.clone())
            if mode == "deferred":
                # a deferred backward nobody consumed: zero_grad() must finish it (never a collective landing in a cleared arena)
                model.optimizer.zero_grad()
                model._backward_device(x, defer_comm=True)
                assert model._pending_comm
                model.optimizer.zero_grad()
                assert not model._pending_comm
        same = [bool(torch.equal(a, b)) for a, b in zip(res["deferred"], res["flat"])]
        # against torch's own Adam on the SUMMED gradient (validates the stand-in's arithmetic, 1 step of difference in order only)
        both = sum(torch.randn(total, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
        gsum_ok = bool(torch.equal(res["flat"][0], both))
        q.put((rank, same, gsum_ok, float(res["deferred"][1].double().sum())))
    except BaseException as e:                              # (the parent must not wait out its time-out for a dead worker)
        import traceback
        q.put((rank, "error", traceback.format_exc(), repr(e)))
        raise
    finally:
        td.destroy_process_group()


def test_deferred_buckets_are_real_async_handles_consumed_one_by_one():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 41500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_async_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    assert all(r[1] != "error" for r in res), [r[2] for r in res if r[1] == "error"]
    res.sort()
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, gsum_ok, psum in res:
        assert all(same), same                              # per-bucket Adam behind its own handle == flat Adam, bit for bit
        assert gsum_ok                                      # every bucket really was summed over the two ranks
    assert res[0][3] == res[1][3]
