#!/bin/bash
# the forced-collectives step (world of one, every bucket through RCCL): where the extra ~3 ms are -- idle gaps and kernel sums
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
O=/root/repo/gpurun_out/r6
cd /tmp
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_fc -o bench -- python /root/repo/bench.py --force-collectives --bucket-mb 64 --steps 6 --warmup 2 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling --no-roofline > $O/bench_fc_prof.json ) 2> /dev/null
cd /root/repo
grep '^{' $O/bench_fc_prof.json | python -c "import json,sys; print('forced-collectives ms_per_step (profiled)', json.loads(sys.stdin.read())['ms_per_step'])"
DB=$(find $O/prof_fc -name "*.db" | head -1)
python tools/fwd_gaps.py $DB --gaps > $O/stream_gaps_fc.txt 2>&1
python tools/kstats_db.py $DB 100 --train-steps 2>&1 | grep -i "nccl\|rccl\|copy\|Memcpy\|total GPU\|elementwise\|mul\|fill" | head -20
rm -rf $O/prof_fc
tail -24 $O/stream_gaps_fc.txt | cut -c1-210
