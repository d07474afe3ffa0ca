#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
timeout 600 tools/probes/gemm_probe 64 10 "3x3" 4 > gpurun_out/r4/probe8_halo.log 2>&1
echo "probe rc=$?" >> gpurun_out/r4/probe8_halo.log
cut -c1-420 gpurun_out/r4/probe8_halo.log
