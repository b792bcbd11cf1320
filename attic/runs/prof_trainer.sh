cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
python tools/rocprof_long.py $OUT/trainer_prof/tr_results.db 20000 | head -30; grep -v "^W2026\|^I2026\|^E2026" $OUT/trainer_prof.log | tail -4 | cut -c1-300; rm -rf $OUT/trainer_prof
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trainer_prof -o tr -- python $GRAFT_REPO_ROOT/tools/bench_trainer.py > $OUT/trainer_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/trainer_prof/tr_results.db $OUT/trainer_stats.md | head -45
tail -1 $OUT/trainer_prof.log | cut -c1-200
python tools/rocprof_long.py $OUT/trainer_prof/tr_results.db 20000 | head -30; grep -v "^W2026\|^I2026\|^E2026" $OUT/trainer_prof.log | tail -4 | cut -c1-300; rm -rf $OUT/trainer_prof
