#!/usr/bin/env python3
"""Which side of test_per_bucket_adam_behind_its_own_allreduce_equals_flat_adam is the one that differs when it fails (1 run in 10
on the shared-GPU gloo set-up): flat / deferred, each twice, gradients snapshotted after EACH of two steps; prints, per rank, which
runs disagree with the majority at which step and in which bucket.  Usage: python tools/lab/dp_flake_probe.py [repeats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.multiprocessing as mp


def worker(rank, world, port, q, sync_mode):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    from ava_amd import dist as adist, synthetic as syn, _lib
    from gpu_util import build_model
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    try:
        z, B = 32, 8
        x = torch.from_numpy(syn.spectrograms(B * world)[B * rank:B * rank + B]).cuda()
        ew, ed = syn.noise(B * world, z)
        sl = slice(B * rank, B * rank + B)
        lib = _lib.load()
        snaps = []
        # local (pre-reduction) gradient buckets of every step, stashed as they are handed to the collective
        import ava_amd.vae as vae_mod
        local = []
        orig = vae_mod._dist.allreduce_gradients_async
        def spy(flat_slice):
            local.append(flat_slice.detach().clone())
            return orig(flat_slice)
        vae_mod._dist.allreduce_gradients_async = spy
        locs = []
        for mode in ("flat", "deferred", "flat", "deferred"):
            model = build_model(z)
            adist.broadcast_parameters(model)
            model.noise_source = lambda b, zz: (ew[sl], ed[sl])
            s = []
            for step in (1, 2):
                model.optimizer.zero_grad()
                model._forward_device(x, need_grad=True)
                if mode == "deferred":
                    model._backward_device(x, defer_comm=True)
                    model.optimizer.step()
                else:
                    model._backward_device(x)
                    _lib.check(lib.ava_adam_step(model._handle, 1e-3, 0.9, 0.999, 1e-8, step, _lib.stream()), "adam")
                torch.cuda.synchronize()
                s.append((model._grads.clone(), model._params.clone()))
            snaps.append(s)
            locs.append(list(local)); del local[:]
        bk = model._buckets()
        out = []
        for step in (0, 1):
            for what in (0, 1):
                ref = snaps[0][step][what]
                for r in range(1, 4):
                    if not torch.equal(ref, snaps[r][step][what]):
                        d = [float((ref[o:o + c].double() - snaps[r][step][what][o:o + c].double()).abs().max()) for o, c in bk]
                        out.append("step %d %s: run %d (%s) != run 0 (flat): per-bucket max diff %s" %
                                   (step + 1, "grads" if what == 0 else "params", r, ("flat", "deferred")[r & 1], d))
        for r in range(1, 4):
            for k in range(len(locs[0])):
                if not torch.equal(locs[0][k], locs[r][k]):
                    out.append("LOCAL gradient, step %d bucket %d: run %d != run 0, max diff %.4g" %
                               (k // 4 + 1, k % 4, r, float((locs[0][k].double() - locs[r][k].double()).abs().max())))
        q.put((rank, out))
    finally:
        td.destroy_process_group()


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    ctx = mp.get_context("spawn")
    bad = 0
    for i in range(reps):
        q = ctx.Queue()
        port = 36000 + (os.getpid() + i) % 2000
        ps = [ctx.Process(target=worker, args=(r, 2, port, q, 0)) for r in range(2)]
        for p in ps: p.start()
        res = sorted(q.get(timeout=600) for _ in ps)
        for p in ps: p.join(timeout=120)
        if any(o for _, o in res):
            bad += 1
            for rank, o in res:
                for line in o: print("rep %d rank %d: %s" % (i, rank, line))
    print("repeats with a disagreement: %d / %d" % (bad, reps))
