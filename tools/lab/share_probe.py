#!/usr/bin/env python3
"""Two processes on ONE GPU, no collectives at all: each runs the same two training steps four times (whole backward, or backward
in parts) and compares its own gradients / parameters run against run.  Separates "the kernels are not deterministic when another
process's kernels share the chip" from "the collective layer".  Usage: python tools/lab/share_probe.py [repeats] [parts 0|1] [turns]
("turns": a lock makes the processes take turns step by step -- both alive, never kernels of both on the chip)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.multiprocessing as mp


def worker(rank, q, parts, barrier, lock=None):
    from ava_amd import synthetic as syn, _lib
    from gpu_util import build_model
    torch.cuda.set_device(0)
    z, B = 32, 8
    x = torch.from_numpy(syn.spectrograms(B * 2)[B * rank:B * rank + B]).cuda()
    ew, ed = syn.noise(B * 2, z)
    sl = slice(B * rank, B * rank + B)
    lib = _lib.load()
    snaps = []
    ws = []; regions = []
    barrier.wait()
    whole = os.environ.get("SHARE_PROBE_WHOLE_RUN_TURNS") == "1"     # the lock is held for a whole run (model set-up + both steps)
    for run in range(6):
        if whole and lock is not None: lock.acquire()
        model = build_model(z)
        model.noise_source = lambda b, zz: (ew[sl], ed[sl])
        s = []
        for step in (1, 2):
            if not whole: barrier.wait()                # lock-step with the other process, as blocking collectives would force
            if lock is not None and not whole: lock.acquire()         # ... but only one process at a time has kernels on the GPU
            model.optimizer.zero_grad()
            model._forward_device(x, need_grad=True)
            fwd = (model._loss_buf.clone(), model._workspace_tensor("xrec", (B, 128, 128)).clone())
            torch.cuda.synchronize()
            if step == 1: ws_f = model._workspace.clone()
            if parts:
                for part in range(lib.ava_backward_num_parts()):
                    _lib.check(lib.ava_backward_part(model._handle, x.data_ptr(), B, part, _lib.stream()), "part")
                model._grad_state = "filled"
            else:
                model._backward_device(x)
            model.optimizer.step()
            torch.cuda.synchronize()
            if lock is not None and not whole: lock.release()
            s.append((model._grads.clone(), model._params.clone(), fwd[0], fwd[1]))
            if step == 1:
                ws.append((ws_f, model._workspace.clone(), model._workspace.data_ptr()))
                if run == 0:
                    import ctypes
                    C = ctypes.CDLL(_lib.LIB_PATH)
                    C.ava_debug_buffer.restype = ctypes.c_void_p
                    for nm in ("y2","y3","y4","y5","y6","y7","y7t","h1","h2","h3","mu","z","h5","h6","h7","f8","f8t","d1","d2","d3","d4","d5","d6","xrec","seed","bn_save","bn_bwd","dz","dF8","wg13"):
                        n = ctypes.c_int64()
                        ptr = C.ava_debug_buffer(model._handle, nm.encode(), ctypes.byref(n))
                        if ptr: regions.append(((ptr - model._workspace.data_ptr()) // 4, n.value, nm))
        snaps.append(s)
        if whole and lock is not None: lock.release()
    out = []
    for step in (0, 1):
        for what in (0, 1, 2, 3):
            ref = snaps[0][step][what]
            for r in range(1, len(snaps)):
                if not torch.equal(ref, snaps[r][step][what]):
                    out.append("step %d %s: run %d != run 0, max diff %.4g" % (step + 1, ("grads", "params", "loss", "xrec")[what], r,
                               float((ref.double() - snaps[r][step][what].double()).abs().max())))
    regions.sort()
    def where(off):
        best = None
        for o, n, nm in regions:
            if o <= off: best = (o, n, nm)
        if best is None: return "before %s" % regions[0][2]
        o, n, nm = best
        return ("%s+%d" % (nm, off - o)) if off < o + n else ("%d floats behind the end of %s" % (off - o - n, nm))
    reg = {nm: (o, n) for o, n, nm in regions}
    wg13 = reg["wg13"][0]                              # convt7's weight-gradient partial rows (model.hip: debug buffer "wg13", round 6)
    slot27 = reg["bn_bwd"][0] + reg["bn_bwd"][1] + 27 * 3200      # bn_acc slot 27 (1600 int64): what the fold adds bn14's backward sums to
    fold_on = os.environ.get("AVA_FOLD13", "1") != "0"       # (lab build: with the fold off the forward leaves these regions unwritten)
    for r in range(1, len(ws) if fold_on else 0):
        a = ws[0][0].view(torch.float32); b = ws[r][0].view(torch.float32)
        wa = a[wg13:wg13 + 128 * 73].view(128, 73); wb = b[wg13:wg13 + 128 * 73].view(128, 73)
        if not torch.equal(wa, wb):
            rows = (wa != wb).any(dim=1).nonzero().flatten().tolist()
            out.append("after forward, run %d: fold weight-gradient partial rows that differ: %s" % (r, rows))
            for row in rows[:3]:
                cols = (wa[row] != wb[row]).nonzero().flatten().tolist()
                out.append("   row %d: %d of 73 columns differ (column 72 = sum of the seed: %s); first: %s" % (row, len(cols),
                           "differs" if 72 in cols else "equal", ["%d: %.6g / %.6g" % (c, float(wa[row, c]), float(wb[row, c])) for c in cols[:10]]))
        la = ws[0][0].view(torch.int64)[slot27 // 2:slot27 // 2 + 1600]; lb = ws[r][0].view(torch.int64)[slot27 // 2:slot27 // 2 + 1600]
        if not torch.equal(la, lb):
            ii = (la != lb).nonzero().flatten().tolist()
            out.append("after forward, run %d: bn_acc slot 27 words that differ: %s" % (r, ["%d: %d / %d" % (k, int(la[k]), int(lb[k])) for k in ii[:16]]))
    cls = []
    for r in range(len(snaps)):
        for k, rep in enumerate(cls):
            if torch.equal(snaps[rep][0][0], snaps[r][0][0]): break
        else:
            cls.append(r)
    if len(cls) > 1:
        out.append("step-1 gradients: %d distinct results over %d runs (first run of each: %s)" % (len(cls), len(snaps), cls))
    # which parameters' gradients differ at step 1 (backward order: convt7 ... convt1, fc8 ... fc1, conv7 ... conv1)
    from ava_amd import layout
    offs, total = layout.arena_offsets(z)
    for r in range(1, len(snaps)):
        if not torch.equal(snaps[0][0][0], snaps[r][0][0]):
            names = []
            for sp in layout.param_specs(z):
                a = snaps[0][0][0][offs[sp.name]:offs[sp.name] + sp.numel]; b = snaps[r][0][0][offs[sp.name]:offs[sp.name] + sp.numel]
                if not torch.equal(a, b):
                    names.append("%s(%.3g/%.3g)" % (sp.name, float((a.double() - b.double()).abs().max()), float(a.abs().max())))
            same = [sp.name for sp in layout.param_specs(z) if torch.equal(snaps[0][0][0][offs[sp.name]:offs[sp.name] + sp.numel],
                                                                             snaps[r][0][0][offs[sp.name]:offs[sp.name] + sp.numel])]
            dec = [n for n in names if n.startswith("convt") or n.startswith("bn1") and n[2:4] in ("10", "11", "12", "13", "14") or n.startswith("bn8") or n.startswith("bn9")]
            out.append("step 1, run %d: %d of %d gradient tensors differ; EQUAL: %s; decoder side: %s" % (r, len(names), len(layout.param_specs(z)), " ".join(same), " ".join(dec)))
    q.put((rank, out))


if __name__ == "__main__":
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ctx = mp.get_context("spawn")
    bad = 0
    for i in range(reps):
        q = ctx.Queue(); barrier = ctx.Barrier(2)
        lock = ctx.Lock() if len(sys.argv) > 3 and sys.argv[3] == "turns" else None
        ps = [ctx.Process(target=worker, args=(r, q, parts, barrier, lock)) for r in range(2)]
        for p in ps: p.start()
        res = sorted(q.get(timeout=600) for _ in ps)
        for p in ps: p.join(timeout=120)
        if any(o for _, o in res):
            bad += 1
            for rank, o in res:
                for line in o: print("rep %d rank %d: %s" % (i, rank, line))
    print("parts=%d: repeats with a disagreement: %d / %d" % (parts, bad, reps))
