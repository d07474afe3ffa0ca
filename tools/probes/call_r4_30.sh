#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g
mkdir -p $O
export PYTHONPATH=ml-mdm_amd
L=$O/split_per_cu.log
rm -f $L
for m in 2 1 2 1; do
export MDM_HIP_SPLIT_PER_CU=$m
for mb in "unet64 4" "unet64 1" "nested1024 4"; do
timeout 300 python tools/sample_bench.py $mb 8 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('per_cu=$m', d['model'], d['batch'], 'eager', d['eager_ms_per_step'], 'graphed', d['graphed_ms_per_step'])" >> $L
done
done
cat $L
