# Final round-4 tree: kernel stats of the headline step (rocprofv3 --kernel-trace --stats of the bench command)
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r4c -o r4c -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/prof_r4c.log 2>&1)
python tools/rocprof_summary.py $OUT/prof_r4c/r4c_results.db $OUT/r4c_stats.md | head -12
rm -rf $OUT/prof_r4c
tail -1 $OUT/prof_r4c.log | cut -c1-300
