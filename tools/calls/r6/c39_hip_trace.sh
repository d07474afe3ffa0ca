#!/bin/bash
# host API timeline against the kernel timeline around the idle gap at the end of backward
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r6
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --hip-runtime-trace -d $O/prof_h -o bench -- python /root/repo/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling --no-roofline > /dev/null ) 2> $O/prof_h.err
cd /root/repo
DB=$(find $O/prof_h -name "*.db" | head -1)
python - "$DB" <<'PY' > $O/hip_trace_gap.txt 2>&1
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
print([t for t in tabs if 'region' in t.lower() or 'api' in t.lower() or 'kernel' in t.lower()][:30])
kcols = [r[1] for r in c.execute("pragma table_info(kernels)")]
rows = c.execute("select name, start, end from kernels order by start").fetchall()
packs = [i for i, r in enumerate(rows) if "pack_weights_multi" in r[0]]
a, b = packs[-2], packs[-1]
seg = rows[a + 1:b + 1]
# largest all-stream idle gap in the segment
ev = sorted(seg, key=lambda r: r[1]); end = ev[0][2]; best = (0, 0, 0)
for r in ev[1:]:
    if r[1] > end and r[1] - end > best[0]: best = (r[1] - end, end, r[1])
    end = max(end, r[2])
print("largest gap %.1f us, from %d to %d" % (best[0] / 1e3, best[1], best[2]))
g0, g1 = best[1], best[2]
# host API calls overlapping [g0 - 300 us, g1 + 100 us]
for t in ("regions", "regions_and_samples", "api"):
    if t in tabs:
        cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
        print(t, cols)
        q = "select * from %s where end > %d and start < %d order by start" % (t, g0 - 1500000, g1 + 1500000)
        out = c.execute(q).fetchall()
        print(len(out), "host records")
        ni = cols.index("name") if "name" in cols else None
        si, ei = cols.index("start"), cols.index("end")
        run = [None, 0, 0.0, 0.0]
        def flush_run():
            if run[0] is not None:
                print("  %9.1f us  ... %9.1f us  x%-4d %s" % (run[2], run[3], run[1], run[0]))
        for r in out:
            nm = r[ni]
            if nm in ("hipGetDevice", "hipSetDevice", "hipGetLastError", "hipEventDestroy", "hipEventCreateWithFlags", "hipEventQuery"):
                continue
            t = (r[si] - g0) / 1e3
            if nm == run[0]:
                run[1] += 1; run[3] = t
            else:
                flush_run()
                run[:] = [nm, 1, t, t]
        flush_run()
        break
PY
rm -rf $O/prof_h
head -c 9000 $O/hip_trace_gap.txt
