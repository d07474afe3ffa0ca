#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "group_norm" > gpurun_out/r4/gn2_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/gn2_tests.log
tail -3 gpurun_out/r4/gn2_tests.log
L=gpurun_out/r4/gn2_kbench.log
echo "== new" > $L
KB_GN_RES=1 timeout 300 python tools/kbench.py gn 2>&1 | grep "^gn " >> $L
echo "== previous build" >> $L
KB_GN_RES=1 MDM_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libmdm_hip_prev.so timeout 300 python tools/kbench.py gn 2>&1 | grep "^gn " >> $L
cat $L
for m in new prev new prev; do
if [ $m = prev ]; then export MDM_HIP_LIB=$GRAFT_REPO_ROOT/ab_libs/libmdm_hip_prev.so; else unset MDM_HIP_LIB; fi
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$m step', d['ms_per_step'], d['nested256'].get('ms_per_step'))" | tee -a gpurun_out/r4/gn2_step.log
done
