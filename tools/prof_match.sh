# kernel times of the 200k x 200k matching call (rocprofv3 kernel trace): tools/prof_match.sh <out name> [clustered]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/tools/match_time.py 200000 $2 > $O/run.log 2>&1 || { echo "failed"; tail -5 $O/run.log; exit 1; }
tail -1 $O/run.log
cut -d, -f1-4 $O/p_kernel_stats.csv | sed 's/(anonymous namespace):://' | cut -c1-110 | head -12
