#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
( timeout 170 python -m pytest tests/test_ops_gpu.py tests/test_diffusion_ops_gpu.py tests/test_trainer_gpu.py tests/test_distributed_gpu.py -x -q ) 2>&1 | tail -3
