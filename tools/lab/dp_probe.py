"""Diagnostic for tests/test_gpu_dist.py::test_per_bucket_adam...: two gloo ranks on one GPU, deferred vs flat, which buckets differ."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.multiprocessing as mp


def worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as td
    from ava_amd import dist as adist, synthetic as syn, _lib
    from gpu_util import build_model
    torch.cuda.set_device(0)
    td.init_process_group("gloo", rank=rank, world_size=world)
    z, B = 32, 8
    x = torch.from_numpy(syn.spectrograms(B * world)[B * rank:B * rank + B]).cuda()
    ew, ed = syn.noise(B * world, z)
    sl = slice(B * rank, B * rank + B)
    lib = _lib.load()
    res = {}
    for mode in ("deferred", "flat", "deferred2", "flat2"):
        model = build_model(z)
        adist.broadcast_parameters(model)
        model.noise_source = lambda b, zz: (ew[sl], ed[sl])
        snaps = []
        for step in (1, 2):
            model.optimizer.zero_grad()
            model._forward_device(x, need_grad=True)
            if mode.startswith("deferred"):
                model._backward_device(x, defer_comm=True)
                model.optimizer.step()
            else:
                model._backward_device(x)
                _lib.check(lib.ava_adam_step(model._handle, 1e-3, 0.9, 0.999, 1e-8, step, _lib.stream()), "adam")
            torch.cuda.synchronize()
            snaps.append((float(model._loss_buf[0].item()), model._grads.clone(), model._params.clone()))
        res[mode] = (snaps, model._buckets())
    out = []
    for a, b in (("deferred", "flat"), ("deferred", "deferred2"), ("flat", "flat2")):
        for s in (0, 1):
            la, ga, pa = res[a][0][s]; lb, gb, pb = res[b][0][s]
            bk = res[a][1]
            gd = [float((ga[o:o + c].double() - gb[o:o + c].double()).abs().max()) for o, c in bk]
            pd = [float((pa[o:o + c].double() - pb[o:o + c].double()).abs().max()) for o, c in bk]
            out.append("rank %d %s vs %s step %d: loss %s grads/bucket %s params/bucket %s" % (rank, a, b, s + 1, la == lb, gd, pd))
    q.put("\n".join(out))
    td.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 38000 + os.getpid() % 1000
    ps = [ctx.Process(target=worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps: p.start()
    for _ in ps: print(q.get(timeout=600))
    for p in ps: p.join()
