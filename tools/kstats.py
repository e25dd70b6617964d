#!/usr/bin/env python3
"""kstats.py <kernel_stats.csv> <steps>: per-kernel microseconds per step from a rocprofv3 --stats summary."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = 0.0
for r in rows:
    per = float(r['TotalDurationNs']) / steps / 1000
    tot += per
    print("%8.1f us/step  calls/step %5.2f  avg %7.1f  %s" % (per, int(r['Calls']) / steps, float(r['AverageNs']) / 1000, r['Name'][:130]))
print("total %.1f us/step" % tot)
