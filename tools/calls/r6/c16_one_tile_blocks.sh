#!/bin/bash
# round 6, call 16: persistent GEMM blocks vs one block per output tile (does a finished block's store drain overlap its successor?)
cd /root/repo
export TMPDIR=/tmp
for m in 0 1; do
echo "--- MDM_HIP_ONE_TILE_BLOCKS=$m"
( MDM_HIP_ONE_TILE_BLOCKS=$m timeout 250 python tools/kbench.py rotate 2>&1 | grep "fresh" | sed 's/| same.*//' )
( MDM_HIP_ONE_TILE_BLOCKS=$m timeout 250 python tools/kbench.py fwd 2>&1 | grep "3x3" )
done
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in 0 1; do
( MDM_HIP_ONE_TILE_BLOCKS=$m timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step one_tile_blocks=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
