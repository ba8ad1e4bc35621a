# rocprofv3 --pmc passes (one run each, counters only + kernel trace) over tools/pmc_workloads.py, round 3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_r03
mkdir -p $O
export PMC_OUT=$O
P="python3 $R/tools/pmc_workloads.py"
timeout -s KILL 400 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $O/s1 -o p -- $P > $O/s1.log 2>&1 &&
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/f -o p -- $P > $O/f.log 2>&1 &&
timeout -s KILL 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/w -o p -- $P > $O/w.log 2>&1 &&
timeout -s KILL 400 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/s2 -o p -- $P > $O/s2.log 2>&1 &&
cd $R && python tools/pmc_collect.py gpurun_out/pmc_r03 gpurun_out/pmc_r03/kernel_counters.json gpurun_out/pmc_r03/kernel_counters.md "rocprofv3 --pmc over tools/pmc_workloads.py, round 3"
