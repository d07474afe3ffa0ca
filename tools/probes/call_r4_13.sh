#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export BENCH_STEP_TIMES=1
F="--no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline --no-nested"
for w in 3 3 12; do
sleep 15
echo "== warmup $w" >> gpurun_out/r4/step_times.log
timeout 200 python bench.py --steps 10 --warmup $w $F 2>&1 | grep "per-step\|^{" | cut -c1-260 >> gpurun_out/r4/step_times.log
done
cat gpurun_out/r4/step_times.log
