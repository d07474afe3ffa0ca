#!/bin/bash
# round 6, call 18: the lockstep test after the settle change, long-horizon errors (printed), what the store phase costs now
cd /root/repo
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_trainer_gpu.py -x -q ) 2>&1 | tail -2
( timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -s -k "long_horizon" ) 2>&1 | grep "long sampling\|passed\|failed"
( timeout 300 python tools/kbench.py stores ) 2>&1 | grep "1x1" | sed 's/| 128x128.*//'
