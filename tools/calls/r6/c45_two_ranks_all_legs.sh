#!/bin/bash
# bench.py at N = 2 with EVERY default leg on (as the driver launches it), both ranks on the one GPU over RCCL: does the
# whole line come out?  (timing meaningless)
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 BENCH_SETTLE_STEPS=1
( MDM_BENCH_DEVICE=0 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r6/bench_two_ranks_all_legs.json 2> gpurun_out/r6/bench_two_ranks_all_legs.err )
echo "rc $?"
grep '^{' gpurun_out/r6/bench_two_ranks_all_legs.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('n_gpus', d['n_gpus'], 'ms_per_step', d['ms_per_step'])
for k in ('roofline','sampling','nested256','reference_loop','nested1024_sampling','cpu_baseline'):
    v=d.get(k)
    print(k, (json.dumps(v)[:260] if v is not None else None))
"
tail -5 gpurun_out/r6/bench_two_ranks_all_legs.err | cut -c1-300
