#!/bin/bash
mkdir -p gpurun_out/r2c2; O=gpurun_out/r2c2
timeout 1500 python -m pytest tests -m gpu -q -s -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
python tools/kbench.py wgrad > $O/kbench_wgrad.log 2>&1
python tools/kbench.py attn > $O/kbench_attn.log 2>&1
grep -E "passed|failed" $O/tests.log | tail -3; grep -E "^FAILED|^ERROR" $O/tests.log | head -40
