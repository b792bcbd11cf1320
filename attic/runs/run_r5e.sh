# Round-5 GPU call e: DP skew test + graph test, then the slow-process-mode hunt: 14 processes with alternating environments.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_dp_procs.py tests/test_gpu_fuzz.py -x -q -m gpu -s -k "delayed or graphs" 2>&1 | grep -E "skew test|graph replay|passed|failed|Error" | tail
: > $OUT/r5e_slow.log
run() { tag=$1; shift; env R5_TAG=$tag B2T_PLAN_DUMP=1 "$@" timeout 200 python tools/r5_slowmode.py 2> $OUT/r5e_err.tmp | grep R5SLOW >> $OUT/r5e_slow.log; grep "hop to the caller" $OUT/r5e_err.tmp | head -1 | sed "s/^/HOPS $tag: /" >> $OUT/r5e_slow.log; }
for rep in 1 2 3; do
  run default
  run nointerrupt HSA_ENABLE_INTERRUPT=0
  run hwq4 GPU_MAX_HW_QUEUES=4
  run hwq8 GPU_MAX_HW_QUEUES=8
done
run nodirect AMD_DIRECT_DISPATCH=0
run workers1 B2T_WORKERS=1
cat $OUT/r5e_slow.log
