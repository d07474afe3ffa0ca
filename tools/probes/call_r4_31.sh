#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g
mkdir -p $O
export PYTHONPATH=ml-mdm_amd
L=$O/split_per_cu_train.log
rm -f $L
for m in 2 1 2 1; do
export MDM_HIP_SPLIT_PER_CU=$m
timeout 200 python bench.py --steps 10 --warmup 3 --workload nested256 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline --no-nested 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('per_cu=$m nested256 train ms', d['ms_per_step'])" >> $L
done
cat $L
