#!/bin/bash
# round 6, call 11: the other load-after-store waits tools/store_wait_scan.py found -- weight-gradient epilogue (slab form
# load-free, arena form pipelined), attention forward tail (cross part requested up front), AdamW trips pipelined -- vs the
# previous commit's library
cd /root/repo
export TMPDIR=/tmp
( timeout 1500 python -m pytest tests/test_ops_gpu.py tests/test_trainer_gpu.py -x -q ) 2>&1 | tail -3
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in product prev; do
( if [ $m != product ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
for m in product prev; do
echo "--- $m: adamw / attention"
( if [ $m != product ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; timeout 200 python tools/kbench.py adamw 2>&1 | tail -2; timeout 200 python tools/attn_debug.py time 2>&1 | grep "^attn B" )
done
