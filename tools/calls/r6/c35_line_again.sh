#!/bin/bash
# the default bench line of the final binary on another box
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
N=${1:-2}
( timeout 700 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r6/bench_default_box$N.json 2> gpurun_out/r6/bench_default_box$N.err
grep '^{' gpurun_out/r6/bench_default_box$N.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], d['binary']['sha256'][:16], 'roofline', d['roofline']['frac'], d['roofline']['avg_launch_ms'], 'nested256', d['nested256']['ms_per_step'], 'sampling', d['sampling']['ms_per_denoise_step'], d['sampling']['fp32_bf16x3']['ms_per_denoise_step'], 'n1024', d['nested1024_sampling']['ms_per_denoise_step'], d['nested1024_sampling']['fp32_bf16x3']['ms_per_denoise_step'])"
