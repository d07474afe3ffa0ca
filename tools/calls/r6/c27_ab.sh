#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in product prev; do
( if [ $m != product ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling --no-roofline > /dev/null ) 2> /dev/null
cd /root/repo
DB=$(find /tmp/prof -name "*.db" | head -1)
python tools/kstats_db.py $DB 90 --train-steps 2>&1 | grep "ln_multi\|space_to_depth\|s2dgrad\|upconv\|masked_mean\|total GPU"
