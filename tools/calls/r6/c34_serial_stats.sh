#!/bin/bash
# every kernel's UNCONTENDED duration: the train step with the weight gradients on the main stream, under rocprofv3
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r6
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_s -o bench -- python /root/repo/bench.py --serial-wgrad --steps 6 --warmup 2 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling --no-roofline > $O/bench_serial.json ) 2> $O/prof_s.err
cd /root/repo
DB=$(find $O/prof_s -name "*.db" | head -1)
python tools/kstats_db.py $DB 95 --train-steps > $O/kernel_stats_serial.txt 2>&1
rm -rf $O/prof_s
grep '^{' $O/bench_serial.json | python -c "import json,sys; print('serial-wgrad ms_per_step', json.loads(sys.stdin.read())['ms_per_step'])"
cut -c1-150 $O/kernel_stats_serial.txt | head -75
