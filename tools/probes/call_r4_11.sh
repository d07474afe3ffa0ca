#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -k "group_norm" > gpurun_out/r4/gn_tests.log 2>&1
echo "tests rc=$?" >> gpurun_out/r4/gn_tests.log
tail -5 gpurun_out/r4/gn_tests.log
echo "== kbench gn, cluster on" > gpurun_out/r4/gn_kbench.log
timeout 300 python tools/kbench.py gn 2>&1 | grep "^gn " >> gpurun_out/r4/gn_kbench.log
echo "== kbench gn, cluster off" >> gpurun_out/r4/gn_kbench.log
MDM_HIP_GN_CLUSTER=0 timeout 300 python tools/kbench.py gn 2>&1 | grep "^gn " >> gpurun_out/r4/gn_kbench.log
echo "== kbench gn + residual gradient, cluster on" >> gpurun_out/r4/gn_kbench.log
KB_GN_RES=1 timeout 300 python tools/kbench.py gn 2>&1 | grep "^gn " >> gpurun_out/r4/gn_kbench.log
echo "== kbench gn + residual gradient, cluster off" >> gpurun_out/r4/gn_kbench.log
KB_GN_RES=1 MDM_HIP_GN_CLUSTER=0 timeout 300 python tools/kbench.py gn 2>&1 | grep "^gn " >> gpurun_out/r4/gn_kbench.log
cat gpurun_out/r4/gn_kbench.log
for m in 1 0 1 0; do
MDM_HIP_GN_CLUSTER=$m timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested1024 --no-sampling --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cluster=$m', d['ms_per_step'], d['nested256'].get('ms_per_step'))" | tee -a gpurun_out/r4/gn_step_ab.log
done
