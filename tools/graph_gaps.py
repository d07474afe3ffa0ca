#!/usr/bin/env python
"""From a rocprofv3 (rocpd) database of tools/sample_bench.py: kernels, busy time and idle time of the LAST graphed sampling
run (the last `n_it` repetitions of the per-iteration kernel sequence).   python tools/graph_gaps.py <results.db> <n_it>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
n_it = int(sys.argv[2])
rows = c.execute("select name, start, end from kernels order by start").fetchall()
# the graphed run is the tail of the trace: find the sampler_step kernels (one per scale per iteration)
idx = [i for i, r in enumerate(rows) if "sampler_step" in r[0]]
per = 1
# iterations end at the last sampler_step of each iteration; take the last n_it iterations
names = [rows[i][0] for i in idx]
scales = 1
while scales < 4 and len(idx) > scales and idx[-1] - idx[-1 - scales] < 8:
    scales += 1
ends = idx[scales - 1::scales] if len(idx) % scales == 0 else idx[::scales]
ends = idx[-1::-scales][::-1]
last = ends[-n_it:]
prev = ends[-n_it - 1]
seg = rows[prev + 1:last[-1] + 1]
wall = (seg[-1][2] - seg[0][1]) / 1e6
busy, cur_s, cur_e = 0, None, None
gaps = []
for r in seg:
    if cur_e is None or r[1] > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append(r[1] - cur_e)
        cur_s, cur_e = r[1], r[2]
    else:
        cur_e = max(cur_e, r[2])
busy += cur_e - cur_s
gaps.sort()
print("%d iterations: %d kernels, wall %.2f ms = %.2f per iteration, busy %.2f ms per iteration, idle %.2f ms per iteration" % (
    n_it, len(seg), wall, wall / n_it, busy / 1e6 / n_it, (wall - busy / 1e6) / n_it))
if gaps:
    print("gaps: %d, median %.1f us, 90th pct %.1f us, max %.1f us, sum %.2f ms" % (len(gaps), gaps[len(gaps) // 2] / 1e3, gaps[int(len(gaps) * 0.9)] / 1e3,
                                                                              gaps[-1] / 1e3, sum(gaps) / 1e6))
durs = sorted(r[2] - r[1] for r in seg)
print("kernel durations: median %.1f us, %d of %d below 10 us (%.2f ms of busy time)" % (durs[len(durs) // 2] / 1e3, sum(1 for d in durs if d < 10e3), len(durs),
                                                                                   sum(d for d in durs if d < 10e3) / 1e6))
