#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
L=gpurun_out/r4/sampling_latency.jsonl
rm -f $L
timeout 300 python tools/sample_bench.py nested1024 4 8 2>&1 | grep '^{' >> $L
timeout 300 python tools/sample_bench.py unet64 4 8 2>&1 | grep '^{' >> $L
timeout 300 python tools/sample_bench.py unet64 1 8 2>&1 | grep '^{' >> $L
timeout 300 python tools/sample_bench.py nested256 16 8 2>&1 | grep '^{' >> $L
cat $L
timeout 300 python tools/sample_shapes.py 4 > gpurun_out/r4/shapes_nested1024_sampling_b4.txt 2>&1; head -12 gpurun_out/r4/shapes_nested1024_sampling_b4.txt
