#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
L=gpurun_out/r4/splitk_rule.log
rm -f $L
for cfg in "6 16" "3 8" "2 6" "3 5"; do
set -- $cfg
export MDM_HIP_SPLIT_MINKT=$1 MDM_HIP_SPLIT_MINSAVE=$2
echo "== min k-tiles per split $1, min saving $2" >> $L
for mb in "unet64 4" "unet64 1" "nested1024 4"; do
timeout 300 python tools/sample_bench.py $mb 8 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('  ', d['model'], d['batch'], 'eager', d['eager_ms_per_step'], 'graphed', d['graphed_ms_per_step'])" >> $L
done
timeout 200 python bench.py --steps 8 --warmup 3 --workload nested256 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline --no-nested 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   nested256 train ms', d['ms_per_step'])" >> $L
done
cat $L
