#!/usr/bin/env python3
"""Per-kernel difference of two rocprofv3 kernel_stats.csv files (us per step; 45 steps per trace by default)."""
import csv, sys
def load(p):
    return {r['Name']: int(r['TotalDurationNs']) for r in csv.DictReader(open(p))}
a, b = load(sys.argv[1]), load(sys.argv[2])
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 45.0
print("total us/step: %.1f -> %.1f" % (sum(a.values()) / steps / 1e3, sum(b.values()) / steps / 1e3))
rows = [((b.get(n, 0) - a.get(n, 0)) / steps / 1e3, n) for n in set(a) | set(b)]
for d, n in sorted(rows):
    if abs(d) > 0.7:
        print("%+7.1f  %7.1f -> %7.1f  %s" % (d, a.get(n, 0) / steps / 1e3, b.get(n, 0) / steps / 1e3, n[:95]))
