#!/usr/bin/env python
"""Summarise a rocprofv3 *_kernel_stats.csv: per-kernel share of GPU time, per step."""
import csv
import sys

path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU kernel time %.2f ms / step (%d kernels)" % (tot / 1e6 / steps, len(rows)))
for r in rows[: int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-90s calls/step %7.1f  ms/step %8.3f  avg %9.1f us  %5.1f%%" % (
        r["Name"][:90], float(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps,
        float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
