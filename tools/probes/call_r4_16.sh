#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "wgrad_direct or (conv_fwd_bwd and 64-3-1) or (conv_fwd_bwd and 64-1-1)" > gpurun_out/r4/wgd_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/wgd_tests.log
tail -15 gpurun_out/r4/wgd_tests.log
timeout 300 python tools/shape_profile.py nested256 2>&1 | grep "wgrad\|GEMM-class" | head -30 | tee gpurun_out/r4/wgd_shapes.log
for m in 1 0 1 0; do
MDM_HIP_WGRAD_DIRECT=$m timeout 200 python bench.py --steps 10 --warmup 3 --workload nested256 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline --no-nested 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('direct=$m nested256 ms', d['ms_per_step'])" | tee -a gpurun_out/r4/wgd_step.log
done
