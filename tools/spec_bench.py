#!/usr/bin/env python3
"""Shotgun-spectrogram path: device rate of get_spec_batch, the whole shotgun training loop fed by
DeviceWindowLoader, and the oracle (the reference's arithmetic) on one host core beside it.

    python tools/spec_bench.py [batch] [batches]        -> one JSON line per parameter set
"""
import contextlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from ava_amd import synthetic as syn, spec as sp
from ava_amd.vae import VAE
from oracle import spec_oracle as so

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NB = int(sys.argv[2]) if len(sys.argv) > 2 else 40

for name, params, seconds in (("finch_window_mwe", syn.FINCH_PARAMS, 20.0), ("mouse_window_mwe", syn.MOUSE_PARAMS, 4.0)):
    p = dict(params)
    fs = p['fs']
    audio, rois = syn.recordings(n_files=4, fs=fs, seconds=seconds)
    ds = sp.DeviceWindowDataset.from_arrays(audio, fs, rois, p, dataset_length=B * NB)
    # ---- get_spec_batch alone: GPU time per batch (events) and host time per call
    idx = list(range(B))
    for _ in range(3):
        ds[idx]
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(NB):
        ds[idx]
    e1.record()
    host_ms = 1e3 * (time.perf_counter() - t0) / NB
    torch.cuda.synchronize()
    gpu_ms = e0.elapsed_time(e1) / NB
    # ---- the training loop fed by the device loader (train_epoch: zero_grad, forward, backward, Adam per batch)
    loader = sp.DeviceWindowLoader(ds, batch_size=B)
    model = VAE(save_dir="", z_dim=32, device_name="cuda")
    with contextlib.redirect_stdout(io.StringIO()):
        model.train_epoch(loader)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        model.train_epoch(loader)
        torch.cuda.synchronize()
    train_ms = 1e3 * (time.perf_counter() - t0) / NB
    # ---- the oracle on one core: the same windows, one get_spec call each (what a DataLoader worker does per item)
    oracle = so.FixedWindowOracle(audio, fs, rois, p)
    n_cpu = 200
    oracle.getitem(list(range(8)), seed=1)
    t0 = time.perf_counter()
    oracle.getitem(list(range(n_cpu)), seed=2)
    cpu_ms = 1e3 * (time.perf_counter() - t0) / n_cpu
    frames = int(np.ceil((p['window_length'] + 0.1) * fs / (p['nperseg'] - p['noverlap']))) + 1
    print(json.dumps({
        "workload": "%s: batch %d windows of %.2f s at %d Hz, nperseg %d / noverlap %d, %d frames each -> [%d,128,128] fp32"
                    % (name, B, p['window_length'], fs, p['nperseg'], p['noverlap'], frames, B),
        "get_spec_batch": {"gpu_ms_per_batch": round(gpu_ms, 4), "host_ms_per_call": round(host_ms, 4),
                           "spectrograms_per_s": round(B / (1e-3 * max(gpu_ms, host_ms)), 1)},
        "shotgun_train_epoch": {"ms_per_step": round(train_ms, 4), "spectrograms_per_s": round(B / (1e-3 * train_ms), 1)},
        "cpu_baseline": {"kind": "port", "cores": 1, "ms_per_spectrogram": round(cpu_ms, 4),
                         "spectrograms_per_s": round(1e3 / cpu_ms, 1),
                         "sample": "%d windows through oracle/spec_oracle.py FixedWindowOracle (scipy stft + FITPACK)" % n_cpu},
    }))
