#!/bin/bash
# GPU_MAX_HW_QUEUES (ROCclr: hardware queues the streams of a process are mapped onto; default 4): the plain step and the
# forced-collectives step (5-6 streams: main, weight-gradient, communication, completion, RCCL's own) under 4 / 8
cd /root/repo
export TMPDIR=/tmp
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2; do
for q in default 4; do
( if [ $q != default ]; then export GPU_MAX_HW_QUEUES=$q; fi   # default = what the package sets (8)
  a=$(timeout 300 python bench.py $B | grep '^{' | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  b=$(timeout 300 python bench.py --force-collectives --bucket-mb 64 $B 2>/dev/null | grep '^{' | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  echo "GPU_MAX_HW_QUEUES=$q  plain $a  forced-collectives $b" )
done
done
