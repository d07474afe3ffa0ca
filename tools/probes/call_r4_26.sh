#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > gpurun_out/r4/attn_qt1_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/attn_qt1_tests.log
tail -3 gpurun_out/r4/attn_qt1_tests.log
L=gpurun_out/r4/attn_qt1_sampling.log
rm -f $L
for mb in "unet64 4" "unet64 1" "nested1024 4" "nested256 16"; do
timeout 300 python tools/sample_bench.py $mb 8 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['model'], d['batch'], 'eager', d['eager_ms_per_step'], 'graphed', d['graphed_ms_per_step'])" >> $L
done
cat $L
timeout 300 python tools/sample_shapes.py 4 2>&1 | grep "attn_fwd" 
