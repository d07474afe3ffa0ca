#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4g
mkdir -p $O
export PYTHONPATH=ml-mdm_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "conv or linear or ffn or text_kv" > $O/deep_tests.log 2>&1; echo "tests rc=$?" >> $O/deep_tests.log
tail -3 $O/deep_tests.log
L=$O/deep_pipe_sampling.log
rm -f $L
for m in 1 0 1 0; do
export MDM_HIP_DEEP_PIPE=$m
for mb in "unet64 4" "unet64 1" "nested1024 4"; do
timeout 300 python tools/sample_bench.py $mb 8 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('deep=$m', d['model'], d['batch'], 'eager', d['eager_ms_per_step'], 'graphed', d['graphed_ms_per_step'])" >> $L
done
done
cat $L
