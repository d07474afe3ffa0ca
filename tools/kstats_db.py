#!/usr/bin/env python
"""Summarise the kernel table of a rocprofv3 (rocpd) results database: per-kernel launches, total and average
duration, share of GPU time.   python tools/kstats_db.py <results.db> [top_n]"""
import re
import sqlite3
import sys

db, top = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("total GPU kernel time %.2f ms over %d distinct kernels" % (tot / 1e6, len(rows)))
print("%-86s %8s %11s %10s %10s %10s %6s" % ("kernel", "calls", "total ms", "avg us", "min us", "max us", "%"))
for name, n, t, avg, mn, mx in rows[:top]:
    short = re.sub(r"\(.*", "", name)
    print("%-86s %8d %11.3f %10.1f %10.1f %10.1f %6.2f" % (short[:86], n, t / 1e6, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / tot))
