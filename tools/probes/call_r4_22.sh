#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export PYTHONPATH=ml-mdm_amd
timeout 300 python tools/find_aten.py unet64 > gpurun_out/r4/aten_unet64.txt 2>&1
head -30 gpurun_out/r4/aten_unet64.txt | cut -c1-250
