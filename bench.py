#!/usr/bin/env python3
"""Benchmark of the VAE training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full optimisation step (zero_grad, forward, backward, gradient
all-reduce when N > 1, Adam) of the reference's VAE (``ava/models/vae.py:347-353``)
on one synthetic batch of 256 spectrograms of 128x128 per GPU, z_dim 32, fp32
(BASELINE.json ``configs[1]``), with the batches already resident in HBM.

N > 1: one process per GPU over RCCL.  Either the caller starts the ranks
(``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N``: RANK /
LOCAL_RANK / WORLD_SIZE are in the environment) or ``bench.py --gpus N`` is run
plainly, in which case it starts the N ranks itself (a child ``torch.distributed.run``,
launched before this process touches the GPU) and relays rank 0's line.  Per-GPU work
is fixed (``"scaling": "weak"``, ``--per-gpu-batch``, default 256) and ``value`` is the
whole-job rate; the same run also times the strong-scaling reading of configs[3]
(``--global-batch``, default 1024, split over the ranks) and reports it under
``strong_scaling``.

Rank 0 prints ONE JSON line: metric/value/unit as the contract requires plus
``roofline`` (conv + convT + BatchNorm kernels forward+backward against the HBM
roofline), ``dist`` (backend, ranks seen, exposed all-reduce
time per step) and, at N = 1, ``cpu_baseline`` (the CPU oracle timed on the host
cores on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

A_CONV_PER_PIXEL = 175.25 * 4              # algorithmic HBM bytes / pixel of a spectrogram, conv fwd+bwd, fp32
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s
CATS = ["conv_fwd", "conv_bwd_data", "conv_wgrad", "bn", "gemm", "layout", "latent_loss", "adam", "pack"]
CONV_FAMILY = ("conv_fwd", "conv_bwd_data", "conv_wgrad", "bn", "pack")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--per-gpu-batch", "--batch", dest="batch", type=int, default=256,
                    help="spectrograms per GPU and step (weak-scaling reading of the metric; this is `value`)")
    ap.add_argument("--global-batch", type=int, default=1024,
                    help="global batch of the strong-scaling leg (configs[3]); 0 skips the leg")
    ap.add_argument("--z-dim", type=int, default=32)
    ap.add_argument("--height", type=int, default=128, help="spectrogram height (reference: 128; configs[4]: 256)")
    ap.add_argument("--width", type=int, default=128, help="spectrogram width (128 or 256)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="storage of the activations between the conv layers (configs[4]: bf16); arithmetic is fp32 either way")
    ap.add_argument("--pool", type=int, default=8, help="distinct device-resident batches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true", help="skip the two HIP-event passes")
    ap.add_argument("--no-loader-path", action="store_true", help="skip the PCIe-inclusive loader-path leg (N = 1 only)")
    ap.add_argument("--cpu-protocol", choices=["bounded", "full"], default="full",
                    help="full (default): BASELINE.md section 3's 3 warm-up + 10 timed steps at the best thread count and on "
                         "all physical cores, for configs[1], plus configs[0] and configs[2] at the best thread count (~3 min); "
                         "bounded: 1 warm-up + 4 timed steps (best thread count) and 1 + 2 (all cores), configs[1] only, ~40 s")
    ap.add_argument("--lr", type=float, default=1e-3, help=argparse.SUPPRESS)   # what-if probes of the lab build use 0
    ap.add_argument("--backend", default="nccl", help=argparse.SUPPRESS)    # 'gloo': ranks share cuda:0 (tests on a 1-GPU box)
    ap.add_argument("--master-port", type=int, default=0, help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def launch_ranks(args):
    """`bench.py --gpus N` without a launcher: start the N ranks as a child torch.distributed.run (this process has
    not initialised the GPU and never does) and relay rank 0's JSON line."""
    import socket
    port = args.master_port
    if not port:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in res.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        else:
            sys.stderr.write(ln + "\n")
    if line is not None:
        print(line)
    raise SystemExit(res.returncode if res.returncode else (0 if line is not None else 1))


def csrc_sha16():
    """Hash of the kernel sources of the running tree (what a PMC summary must have been measured on to be quoted)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "autoencoded-vocal-analysis_amd", "csrc")
    for p in sorted(glob.glob(os.path.join(d, "*.hip")) + glob.glob(os.path.join(d, "*.h"))):
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def one_step(model, x):
    """The body of VAE.train_epoch's loop (vae.py:347-353)."""
    model.optimizer.zero_grad()
    model._forward_device(x, need_grad=True, accumulate=True)
    model._backward_device(x, defer_comm=True)       # data parallel: the Adam step consumes the gradient buckets one by one
    model.optimizer.step()


def loader_path(model, B, z_dim, shape):
    """PCIe-inclusive rate of VAE.train_epoch fed from HOST memory (never `value`): items collated into the build's
    page-locked ring in their stored dtype (PinnedBatchLoader), raw DMA on a copy stream and device-side cast
    (DeviceFeeder + ava_cast_to_f32), against the reference's hand-over (CPU float32 batches, synchronous
    data.to(device) per step, vae.py:349)."""
    import contextlib
    import numpy as np
    import torch
    from ava_amd import synthetic as syn
    from ava_amd.feed import PinnedBatchLoader
    nb, epochs = 8, 4
    base = syn.spectrograms(B * nb, salt=1001, shape=shape)
    out = {"unit": "spectrograms/s", "batches_per_epoch": nb, "epochs_timed": epochs, "note": "host-resident data, H2D inside the timed region"}

    def run(loader, prefetch):
        nb = len(loader)
        model.prefetch = prefetch
        with contextlib.redirect_stdout(sys.stderr):
            model.train_epoch(loader)                          # warm-up epoch (ring allocation, slots)
            torch.cuda.synchronize()
            best = 0.0
            for _ in range(2):                                 # best of two timed runs (host-side jitter)
                t0 = time.perf_counter()
                for _ in range(epochs):
                    model.train_epoch(loader)                  # ends with a host sync (loss read-back)
                torch.cuda.synchronize()
                best = max(best, B * nb * epochs / (time.perf_counter() - t0))
        return round(best, 1)

    class Ref:                                                 # the reference's loader contract: CPU float32 batches
        dataset = range(B * nb)
        def __init__(self): self.b = [torch.from_numpy(base[i * B:(i + 1) * B]) for i in range(nb)]
        def __iter__(self): return iter(self.b)
        def __len__(self): return nb
    out["reference_handover_sync_to_device"] = run(Ref(), False)
    out["pinned_ring_float32"] = run(PinnedBatchLoader(base, batch_size=B, shuffle=True), True)
    # an epoch boundary (loss read-back, new permutation, first collation + first copy with the GPU idle) costs ~1.4 ms, about one
    # step: tools/lab/epoch_bubble.py.  The same loader with epochs four times as long shows what is left of it
    epochs = 1
    out["pinned_ring_float32_%d_batches_per_epoch" % (4 * nb)] = run(PinnedBatchLoader(np.concatenate([base] * 4), batch_size=B, shuffle=True), True)
    epochs = 4
    out["pinned_ring_float64_device_cast"] = run(PinnedBatchLoader(base.astype(np.float64), batch_size=B, shuffle=True), True)
    # uint8 items are 0 or 1 here (numpy_to_tensor casts, it does not rescale: 0..255-valued data would drive the loss to 1e9
    # and time Adam on a model being wrecked); the bytes over PCIe and the device-side cast are the same
    out["pinned_ring_uint8_device_cast"] = run(PinnedBatchLoader((base > 0.5).astype(np.uint8), batch_size=B, shuffle=True), True)
    out["uint8_fixture"] = "binary {0,1} spectrograms (unnormalised 0..255 data is not what the cast would be fed in practice)"
    model.prefetch = True
    return out


def shotgun_path(model, B):
    """SURVEY 8 f4 (never `value`): VAE.train_epoch fed by spectrograms computed ON THE DEVICE from HBM-resident audio
    (ava_amd.spec.DeviceWindowLoader: the reference's FixedWindowDataset + get_spec + DataLoader workers,
    window_vae_dataset.py:102-256, preprocessing/utils.py:18-110), with the finch_window_mwe.py parameters; beside it
    the oracle's get_spec on one host core, which is what one DataLoader worker of the reference does per item."""
    import contextlib
    import torch
    from ava_amd import synthetic as syn
    from ava_amd import spec as sp
    from oracle import spec_oracle as so
    from ava_amd.vae import VAE
    # a fresh model: the one of the timed region has by now taken hundreds of Adam steps on uniform-noise spectrograms
    model = VAE(save_dir="", z_dim=model.z_dim, device_name="cuda")
    p = dict(syn.FINCH_PARAMS)
    nb = 16
    audio, rois = syn.recordings(n_files=4, fs=p['fs'], seconds=20.0)
    ds = sp.DeviceWindowDataset.from_arrays(audio, p['fs'], rois, p, dataset_length=B * nb)
    loader = sp.DeviceWindowLoader(ds, batch_size=B)
    idx = list(range(B))
    for _ in range(3):
        ds[idx]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(nb):
        ds[idx]
    e1.record()
    torch.cuda.synchronize()
    spec_ms = e0.elapsed_time(e1) / nb
    with contextlib.redirect_stdout(sys.stderr):
        model.train_epoch(loader)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            model.train_epoch(loader)
        torch.cuda.synchronize()
    rate = 2 * B * nb / (time.perf_counter() - t0)
    oracle = so.FixedWindowOracle(audio, p['fs'], rois, p)
    oracle.getitem(list(range(8)), seed=1)
    n_cpu = 400
    t0 = time.perf_counter()
    oracle.getitem(list(range(n_cpu)), seed=2)
    cpu_rate = n_cpu / (time.perf_counter() - t0)
    return {"unit": "spectrograms/s", "workload": "finch_window_mwe.py parameters: 0.12 s windows at 32 kHz, nperseg 512 / noverlap 256, "
            "batch %d, spectrograms made on the device from HBM-resident int16 audio" % B,
            "train_epoch_fed_by_device_spectrograms": round(rate, 1),
            "get_spec_batch_ms_per_batch": round(spec_ms, 4), "get_spec_batch_rate": round(B / (1e-3 * spec_ms), 1),
            "cpu_get_spec_one_core": round(cpu_rate, 1), "cpu_sample": "%d windows through oracle/spec_oracle.py" % n_cpu}


def physical_cores():
    """physical host cores (unique (package, core) pairs); falls back to os.cpu_count()"""
    try:
        pairs, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":")[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":")[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
        if pairs:
            return len(pairs)
    except OSError:
        pass
    return os.cpu_count() or 1


def cpu_model_string():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(batch, z_dim, protocol, shape=(128, 128), extra_configs=True):
    """CPU oracle (stock PyTorch-CPU ops = the ATen kernels the reference dispatches to, autograd backward, restated
    Adam) on the SAME batch shape; median of the timed steps.  Two thread settings: the fastest one found on this host
    class (tools/cpu_threads.py: 16 of 8/16/32/64/128 on the 2x64-core EPYC 9575F box) and all cores."""
    import numpy as np
    import torch
    from ava_amd import synthetic as syn
    from oracle import vae_oracle as O
    cores = os.cpu_count() or 1
    phys = physical_cores()
    x = torch.from_numpy(syn.spectrograms(batch, shape=shape))
    ew, ed = [torch.from_numpy(a) for a in syn.noise(batch, z_dim)]

    def run(threads, warm, timed):
        torch.set_num_threads(threads)
        P = O.to_params(syn.fixture_parameters(z_dim, shape), requires_grad=True)
        running = O.fresh_running_stats()
        opt = {"step": 0, "m": {}, "v": {}}
        t0 = time.perf_counter()
        O.train_step(P, x, ew, ed, running, opt)                   # first warm-up step, timed to bound the rest
        first = time.perf_counter() - t0
        if first > 8.0:                                            # oversubscribed thread count: keep the run bounded
            warm, timed = 1, min(timed, 3)
        for _ in range(warm - 1):
            O.train_step(P, x, ew, ed, running, opt)
        ts = []
        for _ in range(timed):
            t0 = time.perf_counter()
            O.train_step(P, x, ew, ed, running, opt)
            ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        return {"threads": threads, "torch_num_threads": torch.get_num_threads(), "warmup": warm, "timed": timed,
                "median_ms_per_step": round(1e3 * med, 1), "value": round(batch / med, 2)}

    best_t = min(cores, 16)
    plan = {"bounded": ((1, 4), (1, 2)), "full": ((3, 10), (3, 10))}[protocol]
    best = run(best_t, *plan[0])
    allc = run(phys, *plan[1]) if phys != best_t else best        # all PHYSICAL host cores
    top = best if best["value"] >= allc["value"] else allc
    out = {"value": top["value"], "unit": "spectrograms/s", "cores": top["threads"], "kind": "port",
           "sample": "median of %d train steps of batch %d (z=%d) after %d warm-up, oracle/vae_oracle.py on torch-CPU"
                     % (top["timed"], batch, z_dim, top["warmup"]),
           "cpu_model": cpu_model_string(), "os_cpu_count": cores, "physical_cores": phys, "protocol": protocol,
           "best_thread_count": best, "all_cores": allc}
    if protocol == "full" and shape == (128, 128) and extra_configs:
        # BASELINE.md section 3 asks for configs 1-3; configs[1] is the headline above
        def cfg(b, zz):
            nonlocal x, ew, ed, batch, z_dim
            keep = (x, ew, ed, batch, z_dim)
            batch, z_dim = b, zz
            x = torch.from_numpy(syn.spectrograms(b, shape=shape))
            ew, ed = [torch.from_numpy(a) for a in syn.noise(b, zz)]
            r = run(best_t, 3, 10)
            x, ew, ed, batch, z_dim = keep
            return r
        out["configs"] = {"configs[0]: batch 64, z=32": cfg(64, 32), "configs[2]: batch 256, z=64": cfg(256, 64)}
    return out


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args)                                    # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the VAE hot path has no CPU fallback")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % ndev if args.backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    if world > 1:
        import torch.distributed as td
        if args.backend == "nccl":
            td.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))
        else:
            td.init_process_group(backend=args.backend)
    from ava_amd import _lib, synthetic as syn
    from ava_amd import dist as adist
    from ava_amd.vae import VAE

    torch.manual_seed(1234)
    H, W = args.height, args.width
    model = VAE(z_dim=args.z_dim, device_name="cuda", x_shape=(H, W), lr=args.lr,
                act_dtype="bfloat16" if args.dtype == "bf16" else "float32")
    adist.broadcast_parameters(model)
    model.train()

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def make_pool(B):
        # device-resident pool of distinct synthetic batches (hash recipe, salt 1001; each rank its own shard)
        return [torch.from_numpy(syn.spectrograms(B, salt=1001, start_item=(rank * args.pool + i) * B, shape=(H, W))).cuda()
                for i in range(args.pool)]

    def timed(pool, steps, warmup):
        for i in range(warmup):
            one_step(model, pool[i % len(pool)])
        sync()
        t0 = time.perf_counter()
        for i in range(steps):
            one_step(model, pool[i % len(pool)])
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device="cuda")
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    B = args.batch
    pool = make_pool(B)
    # CU reserve (dist.cu_reserve): measured, not assumed -- with more than one rank and no AVA_CU_RESERVE given, short timed
    # runs at 0 / 16 / 32 reserved CUs decide which one the timed region below runs with (every rank takes the same decision:
    # the times are MAX-reduced).  Part of the warm-up as far as the contract goes: nothing here is counted in `value`.
    reserve_sweep = None
    if world > 1 and "AVA_CU_RESERVE" not in os.environ:
        reserve_sweep = {}
        ssteps = max(5, args.steps // 2)
        for r in (0, 16, 32):
            os.environ["AVA_CU_RESERVE"] = str(r)
            reserve_sweep[str(r)] = round(1e3 * timed(pool, ssteps, max(3, args.warmup // 2)) / ssteps, 4)
        # every rank holds the same (MAX-reduced) times, so every rank takes the same decision; ties go to the smaller reserve
        best_r = min(reserve_sweep, key=lambda k: (reserve_sweep[k], int(k)))
        os.environ["AVA_CU_RESERVE"] = best_r
        reserve_sweep = {"ms_per_step": reserve_sweep, "steps_each": ssteps, "chosen": int(best_r)}
    dt = timed(pool, args.steps, args.warmup)
    model._check_status()
    ms_per_step = 1e3 * dt / args.steps
    value = world * B * args.steps / dt
    elbo = float(model._loss_buf[0].item()) / B

    # ---- exposed communication: time the compute stream spends waiting for the gradient all-reduces --------------
    dist_info = {"backend": (torch.distributed.get_backend() if world > 1 else None), "world_size": world,
                 "ranks_seen": world, "grad_allreduce_bytes_per_step": int(model._grads.numel()) * 4 if world > 1 else 0,
                 "buckets": _lib.load().ava_backward_num_parts() if world > 1 else 0,
                 # CUs every persistent grid leaves free for the collective's workgroups (dist.cu_reserve); sharded optimizer?
                 "cu_reserve": adist.cu_reserve() if world > 1 else 0, "reserve_sweep": reserve_sweep,
                 "cu_reserve_applies_to": ("backward parts 1..3 (beside the bucket all-reduces); NOT the per-bucket Adam launches: "
                                           "grid-stride blocks without a static tile partition, they fill whatever slots are free") if world > 1 else None,
                 "adam": "per bucket, each behind its own all-reduce" if world > 1 else "one flat launch",
                 "sharded_adam": bool(model._sharded_adam()) if world > 1 else False}
    if world > 1:
        t = torch.ones(1, device="cuda")
        torch.distributed.all_reduce(t)
        dist_info["ranks_seen"] = int(round(float(t.item())))
        model._comm_events = []
        n_comm = min(args.steps, 50)
        for i in range(n_comm):
            one_step(model, pool[i % len(pool)])
        sync()
        ev = model._comm_events
        model._comm_events = None
        exposed = sum(a.elapsed_time(b) for a, b in ev) / max(n_comm, 1)      # waits of a step: one per bucket
        te = torch.tensor([exposed], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(te, op=torch.distributed.ReduceOp.MAX)
        dist_info["exposed_comm_ms_per_step"] = round(float(te.item()), 4)

    # ---- roofline leg: same K steps with HIP events on the launch stream -----------------------------------------
    roofline = None
    if not args.no_roofline:
        lib = _lib.load()
        ms = (ctypes.c_float * 16)()
        cnt = (ctypes.c_int * 16)()
        lib.ava_profile_enable(model._handle, 1)                # fine pass: an event after every launch group
        for i in range(args.steps):
            one_step(model, pool[i % len(pool)])
            lib.ava_profile_read(model._handle, ms, cnt)
        lib.ava_profile_enable(model._handle, 0)
        torch.cuda.synchronize()
        ev_sum_ms = sum(ms[i] for i in range(len(CATS))) / args.steps
        per_step = {c: ms[i] / args.steps for i, c in enumerate(CATS)}
        launches = {c: cnt[i] // args.steps for i, c in enumerate(CATS)}
        conv_ms_fine = sum(per_step[c] for c in CONV_FAMILY)
        # coarse pass: events only where the kernel family changes (~10 records per step instead of ~100, which
        # stretch the step by 10-15 %): the conv family's kernel time as it is inside the timed step
        ms2 = (ctypes.c_float * 16)()
        cnt2 = (ctypes.c_int * 16)()
        lib.ava_profile_enable(model._handle, 2)
        for i in range(args.steps):
            one_step(model, pool[i % len(pool)])
            lib.ava_profile_read(model._handle, ms2, cnt2)
        lib.ava_profile_enable(model._handle, 0)
        torch.cuda.synchronize()
        conv_ms = sum(ms2[i] for i, c in enumerate(CATS) if c in CONV_FAMILY) / args.steps
        coarse_sum_ms = sum(ms2[i] for i in range(len(CATS))) / args.steps
        coarse_events = sum(cnt2[i] for i in range(len(CATS))) // args.steps
        # per layer forward reads in + writes out (activations), backward reads dOut + saved in and writes
        # dIn; sum over layers of in = out = 35.25*H*W elements, conv1 forms no dIn.  Activations take s_a bytes (4, or
        # 2 with bf16 storage), gradients always 4:  A_conv = H*W*(35.25*(3*s_a + 8) - 4)  (= 175.25*4*H*W in fp32)
        s_a = 2 if args.dtype == "bf16" else 4
        a_conv = H * W * (35.25 * (3 * s_a + 8) - 4)
        achieved = B * a_conv / (conv_ms * 1e-3) / 1e9 if conv_ms > 0 else 0.0
        # HBM bytes of the same kernels from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc
        # passes of this command; tools/pmc_traffic.py) -- counters cannot be read from inside this process, so the
        # committed summary of the latest pass is quoted; null when there is none for this configuration
        # The summary records the hash of the kernel sources it was measured on (tools/pmc_traffic.py); it is quoted
        # only when that is the running tree's, otherwise `traffic` is null and `traffic_stale` says what exists.
        traffic, traffic_src, traffic_stale = None, None, None
        import glob
        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))[::-1]:
            try:
                tj = json.load(open(path))
                if B == 256 and world == 1 and args.z_dim == 32 and (H, W) == (128, 128):
                    if tj.get("csrc_sha16") == csrc_sha16():
                        traffic, traffic_src = tj["conv_family_bytes_per_step"], os.path.relpath(path, ROOT)
                    else:
                        traffic_stale = {"source": os.path.relpath(path, ROOT), "recorded_on_csrc_sha16": tj.get("csrc_sha16"),
                                         "running_csrc_sha16": csrc_sha16(),
                                         "conv_family_bytes_per_step": tj.get("conv_family_bytes_per_step")}
                break
            except Exception:
                pass
        roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "traffic_stale": traffic_stale,
                    "kernel": "conv+convT+BatchNorm kernels, forward+backward (SURVEY 8d aggregate)",
                    "algorithmic_bytes_per_step": B * a_conv, "conv_family_ms_per_step": round(conv_ms, 4),
                    # `achieved` uses the coarse pass (HIP events only at kernel-family boundaries); the fine pass
                    # (an event after each of the ~100 launch groups) stretches the step and is informational
                    "coarse_pass": {"events_per_step": coarse_events, "all_kernels_ms_per_step": round(coarse_sum_ms, 4)},
                    "fine_pass": {"all_kernels_ms_per_step": round(ev_sum_ms, 4), "conv_family_ms_per_step": round(conv_ms_fine, 4)},
                    "ms_per_step_by_category": {k: round(v, 4) for k, v in per_step.items()},
                    "launch_groups_per_step": launches}

    # ---- strong-scaling reading of configs[3]: a fixed global batch split over the ranks --------------------------
    strong = None
    if args.global_batch and args.global_batch % world == 0 and args.global_batch // world != B:
        Bs = args.global_batch // world
        del pool
        spool = make_pool(Bs)
        ssteps, swarm = max(10, args.steps // 2), max(5, args.warmup // 2)
        sdt = timed(spool, ssteps, swarm)
        strong = {"global_batch": args.global_batch, "per_gpu_batch": Bs, "steps": ssteps, "warmup": swarm,
                  "ms_per_step": round(1e3 * sdt / ssteps, 4), "value": round(args.global_batch * ssteps / sdt, 1),
                  "unit": "spectrograms/s", "scaling": "strong"}
        del spool

    out = {"metric": "spectrograms/sec, VAE train step (fwd+bwd+Adam), %dx%d, batch %d per GPU" % (H, W, B),
           "value": round(value, 1), "unit": "spectrograms/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None,
           "dtype": (("f32 (fc1/fc8 products, the forward convolutions with >= 8 input channels and the fused backward kernels "
                      "(data gradient AND weight gradient): fp32 operands as three bf16 limbs on bf16 MFMA, six limb products, fp32 "
                      "accumulate; the 1<->8-channel layers, convt6's forward, conv7's forward and the small fully connected products "
                      "on packed fp32 FMA / fp32 MFMA)") if args.dtype == "f32" else
                     ("bf16 conv arithmetic + bf16 activation storage, fp32 everything else (the twelve convolutions with >= 8 "
                      "channels on both sides: weights and BatchNorm outputs rounded to bf16, ONE bf16 MFMA product per forward "
                      "term, fp32 accumulate; their backward: fp32 gradients as three bf16 limbs against the rounded operands, "
                      "three products; conv1 / convt7, BatchNorm statistics, fc layers (three-limb), ELBO and Adam fp32)")), "data": "synthetic",
           "config": {"workload": "configs[%d]: mouse_sylls VAE, batch %d synthetic %dx%d fp32 spectrograms per GPU, z=%d, "
                                  "train step = zero_grad+forward+backward%s+Adam, device-resident batches"
                                  % (4 if (H, W) != (128, 128) else (3 if world > 1 else (2 if args.z_dim == 64 else 1)), B, H, W, args.z_dim,
                                     "+RCCL grad all-reduce" if world > 1 else ""),
                      "global_batch": world * B, "per_gpu_batch": B, "z_dim": args.z_dim, "parallelism": "dp%d" % world},
           "elbo_last_batch_mean": round(elbo, 3), "dist": dist_info}
    if roofline is not None:
        out["roofline"] = roofline
    if strong is not None:
        out["strong_scaling"] = strong
    if world == 1 and not args.no_loader_path:
        model._ensure(B)
        out["loader_path"] = loader_path(model, B, args.z_dim, (H, W))
        if (H, W) == (128, 128):
            try:
                out["shotgun_path"] = shotgun_path(model, B)
            except ValueError as e:           # a diverged posterior on the synthetic recordings must not cost the headline line
                out["shotgun_path"] = {"error": str(e)}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B, args.z_dim, args.cpu_protocol, (H, W))
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
