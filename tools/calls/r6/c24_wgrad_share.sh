#!/bin/bash
# round 6, call 24: split weight-gradient launches sized for a SHARE of the CUs (fewer, longer blocks), so that the main
# stream's HBM-bound kernels find free CUs while they run
cd /root/repo
export TMPDIR=/tmp
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
for m in 100 75 50 38; do
( MDM_HIP_WGRAD_SLOT_PCT=$m timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step wgrad_slot_pct=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
