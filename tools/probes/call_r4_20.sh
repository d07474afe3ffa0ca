#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
timeout 900 python -m pytest tests/test_model_gpu.py -x -q -k "lmhead or lm_head or mini_unet" > gpurun_out/r4/lmhead_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/lmhead_tests.log
tail -6 gpurun_out/r4/lmhead_tests.log
timeout 300 python bench.py --steps 8 --warmup 3 --force-collectives --wire-bf16 --bucket-mb 64 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline --no-nested > gpurun_out/r4/force_collectives.log 2>&1
grep '^{' gpurun_out/r4/force_collectives.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('force-collectives bf16 wire:', d['ms_per_step'], json.dumps(d['config']['comm'])[:900])"
tail -3 gpurun_out/r4/force_collectives.log | cut -c1-300
