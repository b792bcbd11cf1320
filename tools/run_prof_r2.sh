set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r2g -o r2g -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/prof_r2g.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r2_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r2_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r2_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r2_write.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/prof_r2g/r2g_results.db $OUT/r2g_stats.md | head -12
echo "== FETCH_SIZE" > $OUT/r2_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r2_fetch/f_results.db >> $OUT/r2_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r2_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r2_write/w_results.db >> $OUT/r2_pmc.txt
cat $OUT/r2_pmc.txt | head -30
rm -rf $OUT/pmc_r2_fetch $OUT/pmc_r2_write
tail -2 $OUT/prof_r2g.log
