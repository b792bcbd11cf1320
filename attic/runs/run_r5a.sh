# Round-5 first GPU call: jitter hazard hunt (verdict 1c), fresh decode profile of the FINAL kernels (1b), a headline bench.
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_step_parity.py -x -q -m gpu -k "timing or schedule_independent" 2>&1 | tail -5
timeout 900 python tools/r5_jitter.py 200 200 30 > $OUT/r5_jitter.log 2>&1; tail -6 $OUT/r5_jitter.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r5dec -o r5dec -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $OUT/prof_r5dec.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_r5dec_fetch -o f -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_r5dec_write -o w -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/prof_r5dec/r5dec_results.db $OUT/r5dec_stats.md | head -14
echo "== FETCH_SIZE" > $OUT/r5dec_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r5dec_fetch/f_results.db >> $OUT/r5dec_pmc.txt
echo "== WRITE_SIZE" >> $OUT/r5dec_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_r5dec_write/w_results.db >> $OUT/r5dec_pmc.txt
head -30 $OUT/r5dec_pmc.txt
rm -rf $OUT/pmc_r5dec_fetch $OUT/pmc_r5dec_write $OUT/prof_r5dec
tail -3 $OUT/prof_r5dec.log
timeout 600 python bench.py --no-cpu-baseline --no-secondary > $OUT/bench_r5a.json 2> $OUT/bench_r5a.err; tail -3 $OUT/bench_r5a.err; cat $OUT/bench_r5a.json
