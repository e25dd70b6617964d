#!/usr/bin/env python3
"""Benchmark of the VAE training hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one full optimisation step (zero_grad, forward, backward, gradient
all-reduce when N > 1, Adam) of the reference's VAE (``ava/models/vae.py:347-353``)
on one synthetic batch of 256 spectrograms of 128x128 per GPU, z_dim 32, fp32
(BASELINE.json ``configs[1]``), with the batches already resident in HBM.
For N > 1 launch through ``python -m torch.distributed.run`` (one rank per GPU);
per-GPU work is fixed (weak scaling) and the value is the whole-job rate.

Rank 0 prints ONE JSON line: metric/value/unit as the contract requires plus
``roofline`` (conv + convT + BatchNorm kernels forward+backward against the HBM
roofline) and, at N = 1, ``cpu_baseline`` (the CPU oracle
timed on the host cores on a bounded sample of the same workload).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch

A_CONV_BYTES = 175.25 * 4 * 128 * 128      # algorithmic HBM bytes / spectrogram, conv fwd+bwd
HBM_PEAK_GBS = 8000.0                      # MI355X_MICROARCH.md: HBM3E 8 TB/s
CATS = ["conv_fwd", "conv_bwd_data", "conv_wgrad", "bn", "gemm", "layout", "latent_loss", "adam", "pack"]
CONV_FAMILY = ("conv_fwd", "conv_bwd_data", "conv_wgrad", "bn", "pack")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--z-dim", type=int, default=32)
    ap.add_argument("--pool", type=int, default=8, help="distinct device-resident batches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=6)
    return ap.parse_args()


def one_step(model, x):
    """The body of VAE.train_epoch's loop (vae.py:347-353)."""
    model.optimizer.zero_grad()
    model._forward_device(x, need_grad=True, accumulate=True)
    model._backward_device(x)
    model.optimizer.step()


def cpu_baseline(batch, z_dim, steps):
    """CPU oracle (stock PyTorch-CPU ops, autograd backward, restated Adam) on `steps` batches."""
    from ava_amd import synthetic as syn
    from oracle import vae_oracle as O
    cores = os.cpu_count() or 1
    threads = min(cores, 16)     # fastest of 8/16/32/64/128 on the 2x64-core EPYC 9575F box (tools/cpu_threads.py)
    torch.set_num_threads(threads)
    P = O.to_params(syn.fixture_parameters(z_dim), requires_grad=True)
    running = O.fresh_running_stats()
    opt = {"step": 0, "m": {}, "v": {}}
    x = torch.from_numpy(syn.spectrograms(batch))
    ew, ed = [torch.from_numpy(a) for a in syn.noise(batch, z_dim)]
    O.train_step(P, x, ew, ed, running, opt)                   # warm-up
    t0 = time.perf_counter()
    for _ in range(steps):
        O.train_step(P, x, ew, ed, running, opt)
    dt = time.perf_counter() - t0
    return {"value": batch * steps / dt, "unit": "spectrograms/s", "cores": threads, "kind": "port",
            "sample": "%d train steps of batch %d (z=%d) after 1 warm-up, oracle/vae_oracle.py on torch-CPU, %d host cores visible"
                      % (steps, batch, z_dim, cores)}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the VAE hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as td
        td.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    from ava_amd import _lib, synthetic as syn
    from ava_amd import dist as adist
    from ava_amd.vae import VAE

    torch.manual_seed(1234)
    model = VAE(z_dim=args.z_dim, device_name="cuda")
    adist.broadcast_parameters(model)
    model.train()
    B = args.batch
    # device-resident pool of distinct synthetic batches (hash recipe, salt 1001; each rank its own shard)
    pool = [torch.from_numpy(syn.spectrograms(B, salt=1001, start_item=(rank * args.pool + i) * B)).cuda()
            for i in range(args.pool)]

    def sync():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        one_step(model, pool[i % len(pool)])
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step(model, pool[i % len(pool)])
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    model._check_status()
    ms_per_step = 1e3 * dt / args.steps
    value = world * B * args.steps / dt

    # ---- roofline leg: same K steps with HIP events around every launch group of the driver -----------
    lib = _lib.load()
    ms = (ctypes.c_float * 16)()
    cnt = (ctypes.c_int * 16)()
    lib.ava_profile_enable(model._handle, 1)                # fine pass: an event after every launch group
    for i in range(args.steps):
        one_step(model, pool[i % len(pool)])
        lib.ava_profile_read(model._handle, ms, cnt)
    lib.ava_profile_enable(model._handle, 0)
    torch.cuda.synchronize()
    ev_sum_ms = sum(ms[i] for i in range(len(CATS))) / args.steps          # all categories, event-bracketed
    per_step = {c: ms[i] / args.steps for i, c in enumerate(CATS)}
    launches = {c: cnt[i] // args.steps for i, c in enumerate(CATS)}
    conv_ms_fine = sum(per_step[c] for c in CONV_FAMILY)
    # coarse pass: events only where the kernel family changes (~20 records per step instead of ~100, which stretch
    # the step by 10-15 %): the conv family's kernel time as it is inside the timed step
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
    achieved = B * A_CONV_BYTES / (conv_ms * 1e-3) / 1e9 if conv_ms > 0 else 0.0
    # HBM bytes of the same kernels from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, separate rocprofv3 --pmc passes of
    # this command; tools/pmc_traffic.py) -- counters cannot be read from inside this process, so the committed summary of
    # the latest pass is quoted; null when there is none for this batch size
    traffic, traffic_src = None, None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_traffic.json")))[::-1]:
        try:
            t = json.load(open(path))
            if B == 256 and world == 1:
                traffic, traffic_src = t["conv_family_bytes_per_step"], os.path.relpath(path, ROOT)
            break
        except Exception:
            pass
    roofline = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "conv+convT+BatchNorm kernels, forward+backward (SURVEY 8d aggregate)",
                "algorithmic_bytes_per_step": B * A_CONV_BYTES, "conv_family_ms_per_step": round(conv_ms, 4),
                # `achieved` uses the coarse pass (HIP events only at kernel-family boundaries); the fine pass below
                # (an event after each of the ~100 launch groups) stretches the step and is informational
                "coarse_pass": {"events_per_step": coarse_events, "all_kernels_ms_per_step": round(coarse_sum_ms, 4)},
                "fine_pass": {"all_kernels_ms_per_step": round(ev_sum_ms, 4), "conv_family_ms_per_step": round(conv_ms_fine, 4)},
                "ms_per_step_by_category": {k: round(v, 4) for k, v in per_step.items()},
                "launch_groups_per_step": launches}

    out = {"metric": "spectrograms/sec, VAE train step (fwd+bwd+Adam), 128x128, batch %d per GPU" % B,
           "value": round(value, 1), "unit": "spectrograms/s", "n_gpus": world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "configs[1]: mouse_sylls VAE, batch 256 synthetic 128x128 fp32 spectrograms per GPU, z=%d, "
                                  "train step = zero_grad+forward+backward+Adam, device-resident batches" % args.z_dim,
                      "global_batch": world * B, "z_dim": args.z_dim, "parallelism": "dp%d" % world},
           "elbo_last_batch_mean": round(float(model._loss_buf[0].item()) / B, 3),
           "roofline": roofline}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(B, args.z_dim, args.cpu_steps)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
