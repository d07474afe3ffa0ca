#!/bin/bash
# round 6, call 25: the final binary -- the whole GPU suite, smoke, then the evidence battery of c19 + the default line
cd /root/repo
mkdir -p gpurun_out/r6
export TMPDIR=/tmp
( timeout 2400 python -m pytest tests -m gpu -q ) > gpurun_out/r6/gpu_tests.txt 2>&1
tail -3 gpurun_out/r6/gpu_tests.txt
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -2
bash tools/calls/r6/c19_final_profiles.sh 2>&1 | tail -32
