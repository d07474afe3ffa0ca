#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
export BENCH_STEP_TIMES=1
L=gpurun_out/r4/first_bench2.log
timeout 300 python tools/kbench.py gn > /dev/null 2>&1
F="--no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline"
for i in 1 2; do
echo "== bench run $i (default nested leg enabled)" >> $L
timeout 300 python bench.py --steps 10 --warmup 3 $F 2>&1 | grep "per-step\|^{" | cut -c1-230 >> $L
done
cat $L
