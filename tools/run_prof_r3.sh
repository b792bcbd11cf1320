# Round-3 profiles (run through gpurun): (1) the headline step, kernel stats + FETCH / WRITE counters (separate passes);
# (2) the decode kernels: WFST cluster search / prune / finalize / lattice, lexicon prefix beam (tools/prof_decode.py).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r3c -o r3c -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/prof_r3c.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r3c_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r3c_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r3c_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r3c_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r3cdec -o r3cdec -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $OUT/prof_r3cdec.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r3cdec_fetch -o f -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r3cdec_write -o w -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/prof_r3c/r3c_results.db $OUT/r3c_stats.md | head -14
python tools/rocprof_summary.py $OUT/prof_r3cdec/r3cdec_results.db $OUT/r3cdec_stats.md | head -14
echo "== FETCH_SIZE" > $OUT/r3c_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3c_fetch/f_results.db >> $OUT/r3c_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r3c_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3c_write/w_results.db >> $OUT/r3c_pmc.txt
echo "== FETCH_SIZE" > $OUT/r3cdec_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3cdec_fetch/f_results.db >> $OUT/r3cdec_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r3cdec_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3cdec_write/w_results.db >> $OUT/r3cdec_pmc.txt
head -30 $OUT/r3cdec_pmc.txt
rm -rf $OUT/pmc_r3c_fetch $OUT/pmc_r3c_write $OUT/pmc_r3cdec_fetch $OUT/pmc_r3cdec_write $OUT/prof_r3c $OUT/prof_r3cdec
tail -2 $OUT/prof_r3c.log; tail -3 $OUT/prof_r3cdec.log
