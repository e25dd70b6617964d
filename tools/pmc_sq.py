#!/usr/bin/env python3
"""Per-kernel summary of the SQ counter pass (tools/run_pmc_sq.sh): matrix-core busy fraction, wait fractions, LDS
bank conflicts.  SQ_VALU_MFMA_BUSY_CYCLES on gfx950 is the sum over the chip's 1024 SIMDs of the cycles their MFMA
pipe was busy (calibrated on the fc1 GEMM: 1,048,576 v_mfma_f32_32x32x2_f32 x 64 cycles = 6.711e7 = the counter), so
    mfma_busy = counter / (1024 SIMDs x kernel duration x 2.4 GHz)
with the duration of the same dispatch (timestamps of the counter pass itself).  The SQ_WAIT* / SQ_ACTIVE* counters
are per-wave cycle sums and are normalised by SQ_WAVE_CYCLES.

    python tools/pmc_sq.py gpurun_out/pmc_sq/sq_counter_collection.csv > profiles/rNN/pmc_sq.json
"""
import collections, csv, json, sys

CLOCK_GHZ, SIMDS = 2.4, 1024


def main():
    rows = collections.defaultdict(lambda: {"n": 0, "ns": 0.0, "c": collections.defaultdict(float)})
    seen = set()
    for r in csv.DictReader(open(sys.argv[1])):
        k = r["Kernel_Name"].replace("void ", "").split("(")[0]
        d = rows[k]
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"])
            d["n"] += 1
            d["ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        d["c"][r["Counter_Name"]] += float(r["Counter_Value"])
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else rows["adam_flat_kernel"]["n"]
    out = {"steps": steps, "clock_ghz": CLOCK_GHZ, "simds": SIMDS,
           "note": "durations are those of the counter pass (kernels run serialised under --pmc); "
                   "mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (simds * duration * clock)", "kernels": {}}
    tot_busy = tot_ns = 0.0
    for k in sorted(rows):
        d = rows[k]
        if d["n"] < steps or k.startswith(("at::", "__amd")):
            continue
        c, wave = d["c"], max(d["c"].get("SQ_WAVE_CYCLES", 0.0), 1.0)
        tot_busy += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        tot_ns += d["ns"]
        out["kernels"][k] = {
            "launches_per_step": d["n"] // steps,
            "avg_us": round(d["ns"] / d["n"] / 1e3, 2),
            "mfma_busy": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (SIMDS * d["ns"] * CLOCK_GHZ), 4),
            "wait_any": round(c.get("SQ_WAIT_ANY", 0.0) / wave, 4),
            "wait_inst_any": round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4),
            "wait_inst_lds": round(c.get("SQ_WAIT_INST_LDS", 0.0) / wave, 4),
            "active_inst_any": round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4),
            "lds_bank_conflict_cycles_per_launch": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / d["n"]),
        }
    out["step_mfma_busy"] = round(tot_busy / (SIMDS * max(tot_ns, 1.0) * CLOCK_GHZ), 4)
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
