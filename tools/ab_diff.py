#!/usr/bin/env python3
"""Per-kernel difference of two rocprofv3 kernel_stats.csv files in us per step.  Steps per trace = calls of the
once-per-step Adam kernel (bench.py under rocprof: warm-up + timed + the event-instrumented passes)."""
import csv, sys


def load(p):
    rows = list(csv.DictReader(open(p)))
    steps = 45.0
    for r in rows:
        if r['Name'].startswith('adam_flat_kernel'):
            steps = float(r['Calls'])
    return {r['Name']: int(r['TotalDurationNs']) / steps / 1e3 for r in rows}


a, b = load(sys.argv[1]), load(sys.argv[2])
print("total us/step: %.1f -> %.1f" % (sum(a.values()), sum(b.values())))
rows = [(b.get(n, 0) - a.get(n, 0), n) for n in set(a) | set(b)]
for d, n in sorted(rows):
    if abs(d) > 0.7:
        print("%+7.1f  %7.1f -> %7.1f  %s" % (d, a.get(n, 0), b.get(n, 0), n[:95]))
