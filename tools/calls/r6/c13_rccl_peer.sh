#!/bin/bash
# round 6, call 13: RCCL with a real peer on one GPU (distinct NCCL_HOSTID per rank -> socket transport), the reducer with
# ncclAvg, and the forced-collectives bench (world of one) against the plain step
cd /root/repo
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_distributed_gpu.py -x -q -rs ) 2>&1 | tail -8
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
( timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step plain', d['ms_per_step'])" ) 2>&1 | tail -1
( timeout 250 python bench.py $B --force-collectives --bucket-mb 64 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); c=d['config']['comm']; print('step forced-collectives', d['ms_per_step'], c.get('backward_window'), [round(x[2],1) for x in (c.get('bucket_timeline_ms') or [])][:30])" ) 2>&1 | tail -1
done
