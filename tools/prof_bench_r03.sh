# rocprofv3 --kernel-trace --stats of the default bench.py command (the summary committed under profiles/)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r03
rm -rf $O && mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -o p -- python3 $R/bench.py > $O/bench_under_rocprof.json 2> $O/bench.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/head -o p -- python3 $R/bench.py --no-extra --no-cpu > $O/bench_headline_under_rocprof.json 2> $O/head.err &&
cp $O/bench/p_kernel_stats.csv $O/r03_rocprofv3_kernel_stats_bench.csv && cp $O/head/p_kernel_stats.csv $O/r03_rocprofv3_kernel_stats_headline.csv && head -4 $O/r03_rocprofv3_kernel_stats_headline.csv | cut -c1-160
