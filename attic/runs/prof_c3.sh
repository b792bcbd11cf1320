# kernel summary of the shipped-shape step (tools/bench_c3.py) under the given environment: prof_c3.sh NAME [ENV=VAL ...]
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
NAME=$1; shift
cd /tmp
rm -rf $OUT/c3_$NAME
env "$@" timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/c3_$NAME -o c3 -- python $GRAFT_REPO_ROOT/tools/bench_c3.py > $OUT/c3_$NAME.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_summary.py $OUT/c3_$NAME/c3_results.db $OUT/c3_${NAME}_stats.md | head -24
tail -1 $OUT/c3_$NAME.log
rm -rf $OUT/c3_$NAME
