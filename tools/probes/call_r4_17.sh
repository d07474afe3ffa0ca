#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "wgrad_direct or (conv_fwd_bwd and 64-1-1)" > gpurun_out/r4/wgd_tests2.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/wgd_tests2.log
tail -4 gpurun_out/r4/wgd_tests2.log
timeout 300 python tools/shape_profile.py nested256 > gpurun_out/r4/shapes_nested256_b.txt 2>&1; grep "wgrad\|GEMM-class" gpurun_out/r4/shapes_nested256_b.txt | head -24
timeout 300 python tools/shape_profile.py nested256 --serial > gpurun_out/r4/shapes_nested256_serial_b.txt 2>&1; grep "GEMM-class" gpurun_out/r4/shapes_nested256_serial_b.txt
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['nested256'].get('ms_per_step'))" | tee -a gpurun_out/r4/wgd_step2.log
