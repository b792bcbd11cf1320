#!/bin/bash
# round 5: Z-batched packed GEMM for the day layer (B2T_ZPACK) + unrolled norm reduction: parity, then A/B in the bf16 steps
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gemm or norm or adamw or clip" 2>&1 | tail -4
timeout 900 python -m pytest tests/test_gpu_step_parity.py tests/test_gpu_trainer.py -q -m gpu -x 2>&1 | tail -4
for e in "B2T_ZPACK=0" "B2T_ZPACK=1" "B2T_ZPACK=0" "B2T_ZPACK=1"; do
  echo "== $e"; env $e timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r5t_ab.log
