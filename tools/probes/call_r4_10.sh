#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 tools/probes/gemm_probe 64 10 "1x1" 6 > gpurun_out/r4/probe7_ring.log 2>&1
echo "probe rc=$?" >> gpurun_out/r4/probe7_ring.log
cat gpurun_out/r4/probe7_ring.log
