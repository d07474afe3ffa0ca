#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4h
export PYTHONPATH=ml-mdm_amd
BENCH_STEP_TIMES=1 timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline --no-nested 2>&1 | grep "per-step\|^{" | cut -c1-700 | tee gpurun_out/r4h/settle_check.log
