#!/bin/bash
# where the GPU is idle in the backward + optimizer window (no stream running a kernel)
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r6
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_g -o bench -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling --no-roofline > /dev/null ) 2> /dev/null
cd /root/repo
DB=$(find $O/prof_g -name "*.db" | head -1)
python tools/fwd_gaps.py $DB --gaps --around > $O/stream_gaps.txt 2>&1
rm -rf $O/prof_g
python - <<'PY2'
t=open("/root/repo/gpurun_out/r6/stream_gaps.txt").read().split("\n\n")
print(t[-2][-6500:])
PY2
exit 0
python - <<'PY'
lines=open('/root/repo/gpurun_out/r6/stream_gaps.txt').read().split('\n\n')
print(lines[-2][:9000] if len(lines)>1 else lines[-1][:9000])
PY
