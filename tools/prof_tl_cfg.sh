# usage: prof_tl_cfg.sh NAME SHAPE [ENV=VAL ...] : kernel trace of a secondary training shape (c3, c3_amp, c2_amp ...) -> gpurun_out/tl_NAME.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
NAME=$1; SHAPE=$2; shift; shift
cd /tmp
rm -rf $OUT/tl_$NAME
(cd $GRAFT_REPO_ROOT && env "$@" timeout 300 rocprofv3 --kernel-trace -d $OUT/tl_$NAME -o tl -- python tools/r4_cfgs.py $SHAPE > $OUT/tl_$NAME.log 2>&1)
cd $GRAFT_REPO_ROOT
python tools/rocprof_timeline.py $OUT/tl_$NAME/tl_results.db 3 1 > $OUT/tl_$NAME.txt
head -16 $OUT/tl_$NAME.txt
rm -rf $OUT/tl_$NAME
