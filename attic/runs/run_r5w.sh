#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_step_parity.py -q -m gpu -x -k "day_layer_on_the_packed or 256_tile or prepacked" 2>&1 | tail -3
rm -rf /tmp/kz; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kz -- python -m pytest tests/test_gpu_step_parity.py -q -m gpu -k "day_layer_on_the_packed" > /tmp/kz.log 2>&1
grep -h "gemm_bf16\|pack_" $(find /tmp/kz -name "*kernel_stats.csv") | cut -c1-160 > $OUT/r5_zpack_in_step.txt; cat $OUT/r5_zpack_in_step.txt
for e in "B2T_ZPACK=0" "B2T_ZPACK=1" ; do
  echo "== $e"; env $e timeout 200 python tools/r4_cfgs.py c3_amp c2_amp c2 2>&1 | tail -1
done 2>&1 | tee gpurun_out/r5w_ab.log
