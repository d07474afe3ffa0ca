#!/bin/bash
# round 4, call 1: conv_gemm_x_kernel vs conv_gemm_bl_kernel, correctness + timing + SQ counters
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r4
timeout 300 tools/probes/gemm_probe 64 20 > gpurun_out/r4/probe1.log 2>&1
echo "probe rc=$?" >> gpurun_out/r4/probe1.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/r4/pmc_a -o pmc --output-format csv -- $GRAFT_REPO_ROOT/tools/probes/gemm_probe 64 3 > $GRAFT_REPO_ROOT/gpurun_out/r4/pmc_a.log 2>&1
timeout 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE -d $GRAFT_REPO_ROOT/gpurun_out/r4/pmc_b -o pmc --output-format csv -- $GRAFT_REPO_ROOT/tools/probes/gemm_probe 64 3 > $GRAFT_REPO_ROOT/gpurun_out/r4/pmc_b.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_kernel_counters.py gpurun_out/r4/pmc_a gpurun_out/r4/pmc_b > gpurun_out/r4/pmc_summary.txt 2>&1
# keep only the summaries (the raw csv files are large)
find gpurun_out/r4/pmc_a gpurun_out/r4/pmc_b -name "*.csv" -size +2M -delete
cat gpurun_out/r4/probe1.log
