# usage: prof_tl.sh NAME [ENV=VAL ...] : kernel trace of a short bench run under the given environment -> gpurun_out/tl_NAME.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
NAME=$1; shift
cd /tmp
rm -rf $OUT/tl_$NAME
env "$@" timeout 300 rocprofv3 --kernel-trace -d $OUT/tl_$NAME -o tl -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-secondary > $OUT/tl_$NAME.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocprof_timeline.py $OUT/tl_$NAME/tl_results.db 3 1 > $OUT/tl_$NAME.txt
head -12 $OUT/tl_$NAME.txt
rm -rf $OUT/tl_$NAME
