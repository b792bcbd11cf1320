# Round-5 profiles of the FINAL tree (run through gpurun): (1) the headline step: kernel stats + FETCH / WRITE counters (separate passes);
# (2) the bf16 steps (shipped shape, configs[1]) with the 256-tile GEMM kernel.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r5 -o r5 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/prof_r5.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r5_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r5_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r5_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r5_write.log 2>&1
for sh in c3_amp c2_amp; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r5_$sh -o r5 -- python $GRAFT_REPO_ROOT/tools/r4_cfgs.py $sh > $OUT/prof_r5_$sh.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/prof_r5/r5_results.db $OUT/r5_stats.md | head -12
for sh in c3_amp c2_amp; do python tools/rocprof_summary.py $OUT/prof_r5_$sh/r5_results.db $OUT/r5_${sh}_stats.md | head -8; tail -1 $OUT/prof_r5_$sh.log; done
echo "== FETCH_SIZE" > $OUT/r5_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r5_fetch/f_results.db >> $OUT/r5_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r5_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r5_write/w_results.db >> $OUT/r5_pmc.txt
head -24 $OUT/r5_pmc.txt
rm -rf $OUT/pmc_r5_fetch $OUT/pmc_r5_write $OUT/prof_r5 $OUT/prof_r5_c3_amp $OUT/prof_r5_c2_amp
tail -1 $OUT/prof_r5.log | cut -c1-400
