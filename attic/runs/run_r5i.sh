set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gemm_bf16_packed" 2>&1 | tail -3
B2T_GEMM_GM=0 timeout 200 python tools/r5_gemm_gm.py 2>&1 | tail -1
B2T_GEMM_GM=8 timeout 200 python tools/r5_gemm_gm.py 2>&1 | tail -1
B2T_GEMM_GM=4 timeout 200 python tools/r5_gemm_gm.py 2>&1 | tail -1
B2T_GEMM_GM=16 timeout 200 python tools/r5_gemm_gm.py 2>&1 | tail -1
for e in "B2T_GEMM_GM=0" "B2T_GEMM_GM=8" "B2T_GEMM_GM=0" "B2T_GEMM_GM=8"; do
  echo "== $e"; env $e timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1
done
timeout 400 python -m pytest tests/test_gpu_dp_procs.py -x -q -m gpu -s -k "delayed" 2>&1 | grep -E "skew|passed|failed" | tail -8
