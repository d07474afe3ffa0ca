#!/bin/bash
# attention forward at L = 256 (training batch) with 16 queries per wave (3 blocks per CU) against 32
cd /root/repo
export TMPDIR=/tmp
for m in product qt1 product qt1; do
( if [ $m != product ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; echo "== $m"; timeout 300 python tools/attn_debug.py timeonly 2>&1 | grep "^attn" | cut -c1-60 )
done
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
for m in product qt1; do
( if [ $m != product ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
