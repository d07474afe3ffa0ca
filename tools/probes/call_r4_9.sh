#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r4; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_full.log 2>&1; echo "tests rc=$?" >> $O/tests_full.log
grep -E "passed|failed|rc=" $O/tests_full.log | tail -3; grep -E "^FAILED|^ERROR" $O/tests_full.log | head -20
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
