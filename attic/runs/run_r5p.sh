#!/bin/bash
# round 5: the 256 x 256 packed bf16 GEMM kernel: parity + timing, then the kernels' own durations from a rocprofv3 trace
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 600 python tools/r5_gemm256.py > gpurun_out/r5p_gemm256.log 2>&1; echo "exit $?" >> gpurun_out/r5p_gemm256.log
rm -rf /tmp/g256; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/g256 -- python tools/r5_gemm256.py > /tmp/g256.log 2>&1
python tools/r5_gemm256_trace.py /tmp/g256 > gpurun_out/r5p_trace.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gemm_bf16_packed" 2>&1 | tail -3 > gpurun_out/r5p_pytest.log
cat gpurun_out/r5p_gemm256.log gpurun_out/r5p_trace.log; cat gpurun_out/r5p_pytest.log
