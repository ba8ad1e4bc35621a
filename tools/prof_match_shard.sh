# kernel times of one rank's share of config 5 when list1 is sharded 8 ways (n1 = 25 000 against the whole list2)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
rm -rf $O && mkdir -p $O
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O -o p -- python3 $R/tools/exp/match_strong_scaling.py 8 > $O/run.log 2>&1 || { echo "failed"; tail -5 $O/run.log; exit 1; }
grep -v "^W2026\|amdgpu.ids" $O/run.log | tail -3
python3 - $O <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1]+'/p_kernel_stats.csv')):
    n=r['Name'].replace('(anonymous namespace)::','').split('(')[0]
    if 'match' in n: print("  %-34s calls %3s avg %10.1f us" % (n, r['Calls'], float(r['AverageNs'])/1e3))
PY
