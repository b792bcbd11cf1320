#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
for sh in c2 c2_amp; do
rm -rf $OUT/tl_$sh
timeout 300 rocprofv3 --kernel-trace -d $OUT/tl_$sh -o tl -- python tools/r4_cfgs.py $sh > $OUT/tl_$sh.log 2>&1
python tools/rocprof_timeline.py $OUT/tl_$sh/tl_results.db 3 1 full > $OUT/tl_${sh}_full6.txt
head -14 $OUT/tl_${sh}_full6.txt
rm -rf $OUT/tl_$sh
done
