#!/bin/bash
# round 6, call 17: the whole GPU suite + smoke on the current build
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/c17_gpu_tests.txt 2>&1
tail -5 gpurun_out/c17_gpu_tests.txt
grep "long sampling" gpurun_out/c17_gpu_tests.txt | head
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -3
