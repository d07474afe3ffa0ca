#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
export MDM_DIST_BACKEND=gloo MDM_BENCH_DEVICE=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-nested1024 --no-roofline > gpurun_out/r4/two_ranks_one_gpu.log 2>&1
echo "rc=$?" >> gpurun_out/r4/two_ranks_one_gpu.log
grep '^{' gpurun_out/r4/two_ranks_one_gpu.log | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('N=2 (one shared GPU, gloo):', d['n_gpus'], d['ms_per_step'], d['value'], json.dumps(d['config']['comm'])[:600]); print('sampling', d.get('sampling',{}).get('ms_per_denoise_step'), 'nested256', (d.get('nested256') or {}).get('ms_per_step'))"
tail -4 gpurun_out/r4/two_ranks_one_gpu.log | cut -c1-300
