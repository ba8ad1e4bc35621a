# kernel times of the matching call for experiment libraries (rocprofv3 kernel trace):
#   tools/exp/run_match_variants.sh name1 name2 ...   -> tools/exp/lib_<name>.so each
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in "$@"; do
  export STEREO_HIP_LIB=$R/tools/exp/lib_$v.so
  [ ! -f "$STEREO_HIP_LIB" ] && { echo "$v: no library"; continue; }
  rm -rf $R/gpurun_out/mv_$v
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mv_$v -o p -- python3 $R/tools/match_time.py > $R/gpurun_out/mv_$v.log 2>&1 || { echo "$v failed"; exit 1; }
  echo "== $v: $(tail -1 $R/gpurun_out/mv_$v.log)"
  grep -E "match_mfma_bf16x1|match_replay|match_exact" $R/gpurun_out/mv_$v/p_kernel_stats.csv | cut -d, -f1-4 | sed 's/(anonymous namespace):://' | cut -c1-100
done
