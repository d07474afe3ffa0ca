#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > gpurun_out/r4/attn_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/attn_tests.log
tail -4 gpurun_out/r4/attn_tests.log
timeout 300 python tools/kbench.py attn 2>&1 | grep "^attn " | tee gpurun_out/r4/attn_kbench.log
timeout 300 python tools/shape_profile.py unet64 --serial 2>&1 | grep "attn\|GEMM-class" | tee gpurun_out/r4/attn_shapes.log
for i in 1 2; do
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['nested256'].get('ms_per_step'))" | tee -a gpurun_out/r4/attn_step.log
done
