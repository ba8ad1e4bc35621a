cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
B="python3 $R/bench.py --steps 5 --warmup 1 --no-cpu --no-extra"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/f -o p -- $B > /dev/null 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/w -o p -- $B > /dev/null 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/s1 -o p -- $B > /dev/null 2>&1 &&
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/s2 -o p -- $B > /dev/null 2>&1 &&
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/pmc2/k -o p -- $B > $R/gpurun_out/pmc2/bench_headline.json 2>/dev/null &&
cd $R && for d in f w s1 s2; do python tools/pmc_summary.py gpurun_out/pmc2/$d/p_counter_collection.csv disp_band; done && head -3 gpurun_out/pmc2/k/p_kernel_stats.csv | cut -c1-200
