#!/bin/bash
# round 6, call 20: phase stagger of the persistent GEMM blocks (now that the store loop is not acknowledgement-serialised)
cd /root/repo
export TMPDIR=/tmp
for m in 0 2 4 8; do
echo "--- MDM_HIP_STAGGER=$m"
( MDM_HIP_STAGGER=$m timeout 250 python tools/kbench.py rotate 2>&1 | grep "fresh" | grep "768->3072\|512->2048\|512->1536\|768->2304" | sed 's/| same.*//' )
done
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
for m in 0 2 4 8; do
( MDM_HIP_STAGGER=$m timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step stagger=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
