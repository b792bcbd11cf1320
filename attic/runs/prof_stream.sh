cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
rm -rf $OUT/stream_prof
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stream_prof -o st -- python $GRAFT_REPO_ROOT/tools/bench_stream_e2e.py > $OUT/stream_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/stream_prof/st_results.db $OUT/stream_stats.md | head -24
tail -3 $OUT/stream_prof.log
rm -rf $OUT/stream_prof
