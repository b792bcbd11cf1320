set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 400 python -m pytest tests/test_gpu_dp_procs.py -x -q -m gpu -s -k "delayed" > $OUT/r5h_skew.log 2>&1; grep -v "^\s*$" $OUT/r5h_skew.log | grep -v "^    \|^E  \|^_ _\|^self\|^timeout" | head -80
timeout 600 python -m pytest tests/test_gpu_step_parity.py -x -q -m gpu -k "prepacked" 2>&1 | tail -8
for e in "B2T_PREPACK=0" "B2T_PREPACK=1" "B2T_PREPACK=0" "B2T_PREPACK=1"; do
  echo "== $e"; env $e timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1
done
