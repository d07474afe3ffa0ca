#!/bin/bash
cd /root/repo
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -k "conv or linear or wgrad or upconv or bias" ) 2>&1 | tail -3
( timeout 600 python -m pytest tests/test_trainer_gpu.py -x -q ) 2>&1 | tail -3
( timeout 250 python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step', d['ms_per_step'], d['binary']['git_describe'])" ) 2>/dev/null
