#!/usr/bin/env python
"""Summarise the kernel table of a rocprofv3 (rocpd) results database: per-kernel launches, total and average
duration, share of GPU time.   python tools/kstats_db.py <results.db> [top_n] [--train-steps]
--train-steps restricts the table to the training steps of a bench.py run (the launches between the first and the
last optimizer kernel), i.e. leaves out warm-up compilation, the sampling leg and the roofline step's neighbours --
the window the live ``roofline.avg_launch_ms`` of bench.py refers to."""
import re
import sqlite3
import sys

args = [a for a in sys.argv[1:] if not a.startswith("--")]
db, top = args[0], int(args[1]) if len(args) > 1 else 40
c = sqlite3.connect(db)
where = ""
if "--train-steps" in sys.argv:
    t = c.execute("select min(start), max(end) from kernels where name like '%adamw_ema_kernel%'").fetchone()
    where = " where start >= %d and end <= %d" % (t[0], t[1])
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 "from kernels" + where + " group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("total GPU kernel time %.2f ms over %d distinct kernels%s" % (tot / 1e6, len(rows), " (training steps only)" if where else ""))
print("%-86s %8s %11s %10s %10s %10s %6s" % ("kernel", "calls", "total ms", "avg us", "min us", "max us", "%"))
for name, n, t, avg, mn, mx in rows[:top]:
    short = re.sub(r"\(.*", "", name)
    print("%-86s %8d %11.3f %10.1f %10.1f %10.1f %6.2f" % (short[:86], n, t / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
