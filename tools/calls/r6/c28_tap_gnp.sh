#!/bin/bash
# skip-gradient tap in the stride-2 input gradient + 4-way split GroupNorm parameter reduce: tests, then A/B in one call
cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "conv_tap or group_norm or gn or conv_fwd_bwd" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_trainer_gpu.py -x -q -m gpu 2>&1 | tail -3
B="--steps 12 --warmup 3 --no-cpu-baseline --no-reference-loop --no-nested --no-nested1024 --no-roofline --no-sampling"
for i in 1 2 3; do
for m in product notap oldgnp; do
( if [ $m = oldgnp ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; if [ $m = notap ]; then export MDM_HIP_NO_CONV_TAP=1; fi; timeout 250 python bench.py $B | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step variant=$m', d['ms_per_step'])" ) 2>&1 | tail -1
done
done
cd /tmp
( timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof -o b -- python /root/repo/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-nested --no-reference-loop --no-nested1024 --no-sampling --no-roofline > /dev/null ) 2> /dev/null
cd /root/repo
DB=$(find /tmp/prof -name "*.db" | head -1)
python tools/kstats_db.py $DB 90 --train-steps 2>&1 | grep "gn_param\|elementwise\|sel4\|Lb1ELi4\|total GPU"
