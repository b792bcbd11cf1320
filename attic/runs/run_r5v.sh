#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_step_parity.py tests/test_gpu_trainer.py -q -m gpu -x 2>&1 | tail -3
for e in "B2T_GI0_CHAIN=0 B2T_ZPACK=0" "B2T_GI0_CHAIN=0" "B2T_GI0_CHAIN=1" "B2T_GI0_CHAIN=0 B2T_ZPACK=0" "B2T_GI0_CHAIN=0" "B2T_GI0_CHAIN=1"; do
  echo "== $e"; env $e timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r5v_ab.log
for sh in c3_amp; do
rm -rf $OUT/tl_$sh
timeout 300 rocprofv3 --kernel-trace -d $OUT/tl_$sh -o tl -- python tools/r4_cfgs.py $sh > $OUT/tl_$sh.log 2>&1
python tools/rocprof_timeline.py $OUT/tl_$sh/tl_results.db 3 1 full > $OUT/tl_${sh}_full5.txt
head -16 $OUT/tl_${sh}_full5.txt
rm -rf $OUT/tl_$sh
done
