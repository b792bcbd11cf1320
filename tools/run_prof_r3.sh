# Round-3 profiles (run through gpurun): (1) the headline step, kernel stats + FETCH / WRITE counters (separate passes);
# (2) the decode kernels: WFST cluster search / prune / finalize / lattice, lexicon prefix beam (tools/prof_decode.py).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r3b -o r3b -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/prof_r3b.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r3b_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r3b_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r3b_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r3b_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r3bdec -o r3bdec -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $OUT/prof_r3bdec.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r3bdec_fetch -o f -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r3bdec_write -o w -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/prof_r3b/r3b_results.db $OUT/r3b_stats.md | head -14
python tools/rocprof_summary.py $OUT/prof_r3bdec/r3bdec_results.db $OUT/r3bdec_stats.md | head -14
echo "== FETCH_SIZE" > $OUT/r3b_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3b_fetch/f_results.db >> $OUT/r3b_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r3b_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3b_write/w_results.db >> $OUT/r3b_pmc.txt
echo "== FETCH_SIZE" > $OUT/r3bdec_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3bdec_fetch/f_results.db >> $OUT/r3bdec_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r3bdec_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3bdec_write/w_results.db >> $OUT/r3bdec_pmc.txt
head -30 $OUT/r3bdec_pmc.txt
rm -rf $OUT/pmc_r3b_fetch $OUT/pmc_r3b_write $OUT/pmc_r3bdec_fetch $OUT/pmc_r3bdec_write $OUT/prof_r3b $OUT/prof_r3bdec
tail -2 $OUT/prof_r3b.log; tail -3 $OUT/prof_r3bdec.log
