# Round-6 profiles of the FINAL tree (run through gpurun): (1) the headline step: kernel stats + FETCH / WRITE counters (separate passes);
# (2) the bf16 steps (shipped shape, configs[1]) with the forward layer wavefront (one sweep launch for the whole stack).
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r6 -o r6 -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/prof_r6.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r6_fetch -o f -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r6_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r6_write -o w -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_r6_write.log 2>&1
for sh in c3_amp c2_amp; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r6_$sh -o r6 -- python $GRAFT_REPO_ROOT/tools/r4_cfgs.py $sh > $OUT/prof_r6_$sh.log 2>&1
done
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/prof_r6/r6_results.db $OUT/r6_stats.md | head -12
for sh in c3_amp c2_amp; do python tools/rocprof_summary.py $OUT/prof_r6_$sh/r6_results.db $OUT/r6_${sh}_stats.md | head -8; tail -1 $OUT/prof_r6_$sh.log; done
echo "== FETCH_SIZE" > $OUT/r6_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r6_fetch/f_results.db >> $OUT/r6_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r6_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r6_write/w_results.db >> $OUT/r6_pmc.txt
head -24 $OUT/r6_pmc.txt
rm -rf $OUT/pmc_r6_fetch $OUT/pmc_r6_write $OUT/prof_r6 $OUT/prof_r6_c3_amp $OUT/prof_r6_c2_amp
tail -1 $OUT/prof_r6.log | cut -c1-400
# counters of the bf16 steps too (the wavefront's memory-side bytes per launch)
cd /tmp
for sh in c2_amp c3_amp; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r6_${sh}_f -o f -- python $GRAFT_REPO_ROOT/tools/r4_cfgs.py $sh > $OUT/pmc_r6_${sh}_f.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r6_${sh}_w -o w -- python $GRAFT_REPO_ROOT/tools/r4_cfgs.py $sh > $OUT/pmc_r6_${sh}_w.log 2>&1
done
cd $GRAFT_REPO_ROOT
for sh in c2_amp c3_amp; do
  echo "== $sh FETCH_SIZE" >> $OUT/r6_pmc_amp.txt; python tools/rocprof_pmc.py $OUT/pmc_r6_${sh}_f/f_results.db >> $OUT/r6_pmc_amp.txt
  echo "== $sh WRITE_SIZE" >> $OUT/r6_pmc_amp.txt; python tools/rocprof_pmc.py $OUT/pmc_r6_${sh}_w/w_results.db >> $OUT/r6_pmc_amp.txt
  rm -rf $OUT/pmc_r6_${sh}_f $OUT/pmc_r6_${sh}_w
done
head -30 $OUT/r6_pmc_amp.txt
bash tools/prof_tl_cfg.sh r6_c2_amp c2_amp > /dev/null 2>&1; head -14 $OUT/tl_r6_c2_amp.txt
bash tools/prof_tl_cfg.sh r6_c3_amp c3_amp > /dev/null 2>&1; head -14 $OUT/tl_r6_c3_amp.txt
