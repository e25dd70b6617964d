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
