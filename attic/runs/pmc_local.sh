cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
rm -rf $OUT/pmc_l_$c
B2T_GRU_LOCAL=fb timeout 300 rocprofv3 --pmc $c --kernel-trace -d $OUT/pmc_l_$c -o p -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $OUT/pmc_l_$c.log 2>&1
echo "== $c"; python $GRAFT_REPO_ROOT/tools/rocprof_pmc.py $OUT/pmc_l_$c/p_results.db | head -6
rm -rf $OUT/pmc_l_$c
done
