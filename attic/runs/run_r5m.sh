set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r5dec2 -o r5dec2 -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $OUT/prof_r5dec2.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/prof_r5dec2/r5dec2_results.db $OUT/r5dec2_stats.md | head -12
rm -rf $OUT/prof_r5dec2
