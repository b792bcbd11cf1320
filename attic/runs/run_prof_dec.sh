# Decode kernels only (WFST cluster search / prune / finalize / lattice, lexicon prefix beam; tools/prof_decode.py): kernel stats
# + FETCH / WRITE counters in separate passes.  Usage (through gpurun): bash tools/run_prof_dec.sh <tag>
set -x
TAG=${1:-r3edec}
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$TAG -o $TAG -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > $OUT/prof_$TAG.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_${TAG}_fetch -o f -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_${TAG}_write -o w -- python $GRAFT_REPO_ROOT/tools/prof_decode.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/prof_$TAG/${TAG}_results.db $OUT/${TAG}_stats.md | head -14
echo "== FETCH_SIZE" > $OUT/${TAG}_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_${TAG}_fetch/f_results.db >> $OUT/${TAG}_pmc.txt
echo "== WRITE_SIZE" >> $OUT/${TAG}_pmc.txt; python tools/rocprof_pmc.py $OUT/pmc_${TAG}_write/w_results.db >> $OUT/${TAG}_pmc.txt
head -12 $OUT/${TAG}_pmc.txt
rm -rf $OUT/pmc_${TAG}_fetch $OUT/pmc_${TAG}_write $OUT/prof_$TAG
grep algorithmic $OUT/prof_$TAG.log
