#!/usr/bin/env python
"""From a rocprofv3 (rocpd) database of bench.py: for each train step, the forward window (end of the weight re-pack of
the previous step -> the loss-backward kernel) and the backward window: wall time, sum of kernel durations per stream,
idle gaps on the main stream.   python tools/fwd_gaps.py <results.db>"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = c.execute("select name, start, end%s from kernels order by start" % (", " + qcol if qcol else "")).fetchall()
packs = [i for i, r in enumerate(rows) if "pack_weights_multi" in r[0]]
for a, b in zip(packs[:-1], packs[1:]):
    seg = rows[a + 1:b + 1]
    lb = next((i for i, r in enumerate(seg) if "loss_bwd_kernel" in r[0]), None)
    if lb is None:
        continue
    for tag, part in (("forward", seg[:lb]), ("backward+opt", seg[lb:])):
        if not part:
            continue
        wall = (max(r[2] for r in part) - part[0][1]) / 1e6
        ksum = sum(r[2] - r[1] for r in part) / 1e6
        # union of busy intervals (any stream)
        busy, cur_s, cur_e = 0, None, None
        for r in sorted(part, key=lambda r: r[1]):
            if cur_e is None or r[1] > cur_e:
                if cur_e is not None:
                    busy += cur_e - cur_s
                cur_s, cur_e = r[1], r[2]
            else:
                cur_e = max(cur_e, r[2])
        busy += cur_e - cur_s
        print("%-13s %4d kernels  wall %7.2f ms  sum of kernel durations %7.2f ms  GPU busy (union) %7.2f ms  idle %5.2f ms" % (
            tag, len(part), wall, ksum, busy / 1e6, wall - busy / 1e6))
        if "--gaps" in sys.argv and tag != "forward":
            # the largest intervals in which NO stream runs a kernel, with the kernel that ended before and the one that began after
            ev = sorted(part, key=lambda r: r[1])
            gaps, end, last = [], ev[0][2], ev[0]
            for r in ev[1:]:
                if r[1] > end:
                    gaps.append((r[1] - end, (end - part[0][1]) / 1e6, last[0][:60], r[0][:60]))
                if r[2] > end:
                    end, last = r[2], r
            gaps.sort(reverse=True)
            for g in gaps[:8]:
                print("      gap %6.1f us at %6.2f ms   after %-60s before %s" % (g[0] / 1e3, g[1], g[2], g[3]))
            print("      %d gaps, %.2f ms in gaps below 20 us" % (len(gaps), sum(g[0] for g in gaps if g[0] < 20e3) / 1e6))
            if qcol:
                # bubbles on each queue alone: between consecutive kernels of the SAME queue (the main stream is the critical
                # path of backward; a bubble there costs step time even while the other stream keeps the GPU busy)
                byq = {}
                for r in ev:
                    byq.setdefault(r[3], []).append(r)
                for qid, rs in sorted(byq.items(), key=lambda kv: -len(kv[1])):
                    gs = [b[1] - a[2] for a, b in zip(rs[:-1], rs[1:]) if b[1] > a[2]]
                    small = [g for g in gs if g < 30e3]
                    print("      queue %-4s %4d kernels  busy %6.2f ms  gaps below 30 us: %4d, %.2f ms (median %.1f us); larger: %d, %.2f ms" % (
                        qid, len(rs), sum(r[2] - r[1] for r in rs) / 1e6, len(small), sum(small) / 1e6,
                        sorted(small)[len(small) // 2] / 1e3 if small else 0.0, len(gs) - len(small), (sum(gs) - sum(small)) / 1e6))
            if "--around" in sys.argv and gaps:
                # the launches either side of the largest gap: start (ms into the window), duration (us), stream, name
                t_gap = gaps[0][1] * 1e6 + part[0][1]
                near = [r for r in ev if r[2] > t_gap - 1.0e6 and r[1] < t_gap + 4.5e6]
                for r in near:
                    print("        %8.3f ms %7.1f us  q%-3s %s" % ((r[1] - part[0][1]) / 1e6, (r[2] - r[1]) / 1e3, r[3] if len(r) > 3 else "?", r[0][:90]))
    print()
