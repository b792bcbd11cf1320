# Round-4 profiles (run through gpurun): (1) the headline step, kernel stats + FETCH / WRITE counters (separate passes);
# (2) the decode kernels: WFST cluster search / prune / finalize / lattice, lexicon prefix beam (tools/prof_decode.py).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r4a -o r4a -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/prof_r4a.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r4a_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r4a_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r4a_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r4a_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r4adec -o r4adec -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $OUT/prof_r4adec.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r4adec_fetch -o f -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r4adec_write -o w -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/prof_r4a/r4a_results.db $OUT/r4a_stats.md | head -14
python tools/rocprof_summary.py $OUT/prof_r4adec/r4adec_results.db $OUT/r4adec_stats.md | head -14
echo "== FETCH_SIZE" > $OUT/r4a_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r4a_fetch/f_results.db >> $OUT/r4a_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r4a_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r4a_write/w_results.db >> $OUT/r4a_pmc.txt
echo "== FETCH_SIZE" > $OUT/r4adec_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r4adec_fetch/f_results.db >> $OUT/r4adec_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r4adec_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r4adec_write/w_results.db >> $OUT/r4adec_pmc.txt
head -30 $OUT/r4adec_pmc.txt
rm -rf $OUT/pmc_r4a_fetch $OUT/pmc_r4a_write $OUT/pmc_r4adec_fetch $OUT/pmc_r4adec_write $OUT/prof_r4a $OUT/prof_r4adec
tail -2 $OUT/prof_r4a.log; tail -3 $OUT/prof_r4adec.log
