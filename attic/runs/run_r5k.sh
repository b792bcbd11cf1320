set -x
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_step_parity.py -x -q -m gpu -k "prepacked or jitter or timing" 2>&1 | tail -8
for e in "B2T_WGRAD_SPLIT=0" "B2T_WGRAD_SPLIT=1" "B2T_WGRAD_SPLIT=0" "B2T_WGRAD_SPLIT=1" "B2T_PREPACK=0"; do
  echo "== $e"; env $e timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1
done
