#!/bin/bash
# nested-256 train step: idle intervals (no stream running a kernel) and the host API mix of one step
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r6
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_n -o bench -- python /root/repo/bench.py --workload nested256 --steps 6 --warmup 2 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling --no-roofline > $O/bench_n256.json ) 2> /dev/null
cd /root/repo
grep '^{' $O/bench_n256.json | python -c "import json,sys; print('nested256 ms_per_step (profiled)', json.loads(sys.stdin.read())['ms_per_step'])"
DB=$(find $O/prof_n -name "*.db" | head -1)
python tools/fwd_gaps.py $DB --gaps > $O/stream_gaps_nested256.txt 2>&1
rm -rf $O/prof_n
tail -26 $O/stream_gaps_nested256.txt | cut -c1-200
