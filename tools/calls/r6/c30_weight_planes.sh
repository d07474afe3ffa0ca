#!/bin/bash
# bf16x3 sampling with the weight operand pre-split into hi / lo planes: tests, then timings in one call
cd /root/repo
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -x -q -m gpu -k "split or planes or conv_tap" 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "long_horizon or fp32 or sampl" 2>&1 | tail -3
for m in product noplanes x3w41 product noplanes x3w41; do
( if [ $m = x3w41 ]; then export MDM_HIP_LIB=/root/repo/ab_libs/lib_$m.so; fi; if [ $m = noplanes ]; then export MDM_HIP_NO_WEIGHT_PLANES=1; fi; echo -n "$m: "; timeout 300 python tools/sample_x3_probe.py 4 2>&1 | tail -1 )
done
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-loop --no-nested --no-roofline 2>&1 | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(json.dumps(d.get('sampling'), indent=1)[:1500])
print(json.dumps(d.get('nested1024_sampling'), indent=1)[:1800])"
