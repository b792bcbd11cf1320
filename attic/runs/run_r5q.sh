#!/bin/bash
# round 5: the 256 x 256 packed GEMM kernel: parity, then the bf16 step A/B by B2T_GEMM_256 (read per call)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/r5_gemm256.py 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step_parity.py -q -m gpu -k "gemm or bf16 or amp" 2>&1 | tail -3
for e in "B2T_GEMM_256=0" "B2T_GEMM_256=1" "B2T_GEMM_256=0" "B2T_GEMM_256=1"; do
  echo "== $e"; env $e timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r5q_ab.log
