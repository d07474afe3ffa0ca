#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
L=gpurun_out/r4/first_bench3.log
KB_GN_RES=1 timeout 300 python tools/kbench.py gn > /dev/null 2>&1
KB_GN_RES=1 timeout 300 python tools/kbench.py gn > /dev/null 2>&1
timeout 300 python tools/shape_profile.py unet64 --serial > /dev/null 2>&1
export BENCH_STEP_TIMES=1
F="--no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline"
for i in 1 2; do
echo "== bench run $i" >> $L
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -4 >> $L
timeout 300 python bench.py --steps 20 --warmup 3 $F 2>&1 | grep "per-step\|^{" | cut -c1-260 >> $L
done
cat $L
