#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
export BENCH_STEP_TIMES=1
L=gpurun_out/r4/first_bench.log
timeout 300 python -m pytest tests/test_ops_gpu.py -x -q -k "attention" > /dev/null 2>&1
timeout 300 python tools/shape_profile.py unet64 --serial > /dev/null 2>&1
F="--no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline --no-nested"
for i in 1 2 3; do
echo "== bench run $i (after pytest + shape_profile)" >> $L
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4 >> $L
timeout 200 python bench.py --steps 10 --warmup 3 $F 2>&1 | grep "per-step\|^{" | cut -c1-230 >> $L
done
timeout 300 python tools/shape_profile.py unet64 --serial > /dev/null 2>&1
echo "== bench run 4 (again right after shape_profile), 20 steps" >> $L
timeout 200 python bench.py --steps 20 --warmup 3 $F 2>&1 | grep "per-step\|^{" | cut -c1-230 >> $L
cat $L
