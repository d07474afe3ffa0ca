#!/bin/bash
# round 6, call 21: the weight-gradient stream restricted to a share of the CUs (hipExtStreamCreateWithCUMask), so that the
# main stream's HBM-bound kernels always find free CUs
cd /root/repo
export TMPDIR=/tmp
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
for m in none 7/8 3/4 5/8 1/2; do
( if [ $m != none ]; then export MDM_HIP_SIDE_CU_KEEP=$m; fi; timeout 250 python bench.py $B 2>&1 | grep '^{\|Error\|error' | python -c "
import json,sys
t=sys.stdin.read()
try:
    d=json.loads([l for l in t.splitlines() if l.startswith('{')][-1]); print('step side_cu_keep=$m', d['ms_per_step'])
except Exception as e: print('step side_cu_keep=$m FAILED', t[-300:])" ) 2>&1 | tail -2
done
done
