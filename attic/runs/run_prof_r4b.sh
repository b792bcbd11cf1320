# Round-4 profile of the bf16 steps (fragment hand-off): kernel stats of the shipped shape and of configs[1] under use_amp
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
for sh in c3_amp c2_amp; do
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r4b_$sh -o r4b -- python $GRAFT_REPO_ROOT/tools/r4_cfgs.py $sh > $OUT/prof_r4b_$sh.log 2>&1)
  python tools/rocprof_summary.py $OUT/prof_r4b_$sh/r4b_results.db $OUT/r4b_${sh}_stats.md | head -12
  rm -rf $OUT/prof_r4b_$sh
  tail -1 $OUT/prof_r4b_$sh.log
done
