#!/bin/bash
# bench.py's N = 2 path on ONE GPU (both ranks on device 0): over RCCL with a real peer, then over gloo.  Timing is
# meaningless (two ranks share the device); the check is that the contract holds: one JSON line from rank 0, n_gpus 2.
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
A="--gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
export BENCH_SETTLE_STEPS=2
( MDM_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py $A 2>&1 | grep '^{\|Error\|error' | tee gpurun_out/r6/bench_two_ranks_rccl.json | cut -c1-300 )
( MDM_BENCH_DEVICE=0 MDM_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py $A 2>&1 | grep '^{\|Error\|error' | tee gpurun_out/r6/bench_two_ranks_gloo.json | cut -c1-300 )
