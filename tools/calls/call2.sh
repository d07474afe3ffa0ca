#!/bin/bash
cd /root/repo
mkdir -p gpurun_out
export TMPDIR=/tmp
( timeout 300 python tools/attn_debug.py time ) > gpurun_out/c2_attn_debug.txt 2>&1
( timeout 120 python tools/attn_debug.py stamps ) > gpurun_out/c2_stamps.txt 2>&1
( timeout 400 python -m pytest tests/test_ops_gpu.py -k "test_attention" -q ) > gpurun_out/c2_attn_tests.txt 2>&1
cd /tmp
( timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d /root/repo/gpurun_out/pmc_a1 -o pmc -- python /root/repo/tools/attn_debug.py pmc ) > /root/repo/gpurun_out/c2_pmc1.log 2>&1
( timeout 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d /root/repo/gpurun_out/pmc_a2 -o pmc -- python /root/repo/tools/attn_debug.py pmc ) > /root/repo/gpurun_out/c2_pmc2.log 2>&1
cd /root/repo
python tools/pmc_kernel_counters.py gpurun_out/pmc_a1 gpurun_out/pmc_a2 --match attn_bwd > gpurun_out/c2_pmc.txt 2>&1
rm -rf gpurun_out/pmc_a1 gpurun_out/pmc_a2
grep -v "^B=" gpurun_out/c2_attn_debug.txt | tail -8; grep -c " ok" gpurun_out/c2_attn_debug.txt; grep "FAIL" gpurun_out/c2_attn_debug.txt | head; cat gpurun_out/c2_stamps.txt; tail -4 gpurun_out/c2_attn_tests.txt; cat gpurun_out/c2_pmc.txt; tail -3 gpurun_out/c2_pmc1.log
