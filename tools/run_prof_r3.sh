# Round-3 profiles (run through gpurun): (1) the headline step, kernel stats + FETCH / WRITE counters (separate passes);
# (2) the decode kernels: WFST cluster search / prune / finalize / lattice, lexicon prefix beam (tools/prof_decode.py).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r3d -o r3d -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/prof_r3d.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r3d_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r3d_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r3d_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r3d_write.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r3ddec -o r3ddec -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $OUT/prof_r3ddec.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r3ddec_fetch -o f -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r3ddec_write -o w -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/prof_r3d/r3d_results.db $OUT/r3d_stats.md | head -14
python tools/rocprof_summary.py $OUT/prof_r3ddec/r3ddec_results.db $OUT/r3ddec_stats.md | head -14
echo "== FETCH_SIZE" > $OUT/r3d_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3d_fetch/f_results.db >> $OUT/r3d_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r3d_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3d_write/w_results.db >> $OUT/r3d_pmc.txt
echo "== FETCH_SIZE" > $OUT/r3ddec_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3ddec_fetch/f_results.db >> $OUT/r3ddec_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r3ddec_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r3ddec_write/w_results.db >> $OUT/r3ddec_pmc.txt
head -30 $OUT/r3ddec_pmc.txt
rm -rf $OUT/pmc_r3d_fetch $OUT/pmc_r3d_write $OUT/pmc_r3ddec_fetch $OUT/pmc_r3ddec_write $OUT/prof_r3d $OUT/prof_r3ddec
tail -2 $OUT/prof_r3d.log; tail -3 $OUT/prof_r3ddec.log
