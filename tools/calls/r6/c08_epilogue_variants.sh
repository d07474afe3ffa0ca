#!/bin/bash
# round 6, call 8: epilogue variants V0 (knob as argument) .. V3 (+ no inline residual load, + explicit operand wait, + counted waits)
cd /root/repo
export TMPDIR=/tmp
for m in v0 v1 v2 v3 r5; do
echo "--- variant=$m"
( export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; if [ $m = r5 ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_r5gelu.so; fi; timeout 250 python tools/kbench.py rotate 2>&1 | grep "fresh" | grep "768->3072\|768->2304\|512->1536" | sed 's/| same.*//' )
done
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in v0 v1 v2 v3 r5; do
( export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; if [ $m = r5 ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_r5gelu.so; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
