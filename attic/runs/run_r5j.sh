set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_step_parity.py -x -q -m gpu -k "prepacked" 2>&1 | tail -8
for e in "B2T_PREPACK=0" "B2T_PREPACK=1" "B2T_PREPACK=0" "B2T_PREPACK=1"; do
  echo "== $e"; env $e timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1
done
cd /tmp
rm -rf $OUT/tl_c3amp
(cd $GRAFT_REPO_ROOT && timeout 300 rocprofv3 --kernel-trace -d $OUT/tl_c3amp -o tl -- python tools/r4_cfgs.py c3_amp > $OUT/tl_c3amp.log 2>&1)
cd $GRAFT_REPO_ROOT
python tools/rocprof_timeline.py $OUT/tl_c3amp/tl_results.db 3 1 full > $OUT/tl_c3amp_full2.txt
head -16 $OUT/tl_c3amp_full2.txt
rm -rf $OUT/tl_c3amp
