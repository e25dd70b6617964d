#!/usr/bin/env python3
"""HBM traffic of the conv family from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as
MI355X_MICROARCH.md prescribes).  Counter values are in KiB; on gfx950 FETCH_SIZE reports exactly half of the
bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is exact for 16 B/lane stores.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [steps] > profiles/rNN/pmc_traffic.json
(steps defaults to the number of adam_flat_kernel launches in the trace)
"""
import collections, csv, json, sys

# the kernels bench.py counts into the conv family (forward, backward, weight gradient, BatchNorm, weight packing)
CONV_PREFIX = ("conv3x3", "thin_", "up88_direct", "bn_stats", "bn_finalize", "pack_all", "pack_stats", "wgrad_reduce")


def per_kernel(path, counter):
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        d[k][0] += 1
        d[k][1] += float(r["Counter_Value"])
    return d


def main():
    fd = per_kernel(sys.argv[1], "FETCH_SIZE")
    wd = per_kernel(sys.argv[2], "WRITE_SIZE")
    # steps in the trace = launches of the once-per-step Adam kernel (warm-up + timed + the event-instrumented passes of bench.py)
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else max(fd.get("adam_flat_kernel", [0])[0], wd.get("adam_flat_kernel", [0])[0])
    kernels = {}
    for k in sorted(set(fd) | set(wd)):
        calls = max(fd.get(k, [0, 0])[0], wd.get(k, [0, 0])[0])
        if calls < steps:
            continue            # set-up kernels (pool creation, fills), not part of a step
        f = 2.0 * 1024 * fd.get(k, [0, 0.0])[1] / steps
        w = 1024.0 * wd.get(k, [0, 0.0])[1] / steps
        kernels[k] = {"launches_per_step": calls // steps, "read_bytes_per_step": round(f), "write_bytes_per_step": round(w)}
    conv = {k: v for k, v in kernels.items() if k.startswith(CONV_PREFIX)}
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import csrc_sha16
    out = {"steps": steps, "unit": "bytes per step (B=256 per GPU)", "csrc_sha16": csrc_sha16(),
           "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 halves wide streaming reads); WRITE_SIZE KiB x 1024",
           "conv_family_bytes_per_step": sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in conv.values()),
           "all_kernels_bytes_per_step": sum(v["read_bytes_per_step"] + v["write_bytes_per_step"] for v in kernels.values()),
           "kernels": kernels}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
