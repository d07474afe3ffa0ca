#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "long_horizon" -s > gpurun_out/r4/long_sampling.log 2>&1
echo "pytest rc=$?" >> gpurun_out/r4/long_sampling.log
grep -E "long sampling|passed|failed|rc=" gpurun_out/r4/long_sampling.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r4/bench_a.log 2>&1
echo "bench rc=$?"
grep '^{' gpurun_out/r4/bench_a.log | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'nested256', d.get('nested256',{}).get('ms_per_step'))
print('sampling', json.dumps(d['sampling']))
print('n1024', json.dumps(d['nested1024_sampling']))
print('roofline', json.dumps(d['roofline'])[:600])
"
tail -3 gpurun_out/r4/bench_a.log | cut -c1-300
