# Slow-mode hypothesis: a process is slow for life when ANOTHER process holds hardware queues on the GPU while it creates its own.
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
: > $OUT/r5o_slow.log
run() { tag=$1; shift; env R5_TAG=$tag "$@" timeout 200 python tools/r5_slowmode.py 2>/dev/null | grep R5SLOW >> $OUT/r5o_slow.log; }
run alone_hwq16
python tools/r5_holder.py 16 400 & H=$!
sleep 25
run with_holder16_hwq16
run with_holder16_hwq16_again
run with_holder16_hwq8 GPU_MAX_HW_QUEUES=8
run with_holder16_hwq4 GPU_MAX_HW_QUEUES=4
kill $H; wait $H 2>/dev/null
sleep 3
run after_holder_hwq16
GPU_MAX_HW_QUEUES=4 python tools/r5_holder.py 4 300 & H=$!
sleep 25
run with_holder4_hwq16
run with_holder4_hwq8 GPU_MAX_HW_QUEUES=8
kill $H; wait $H 2>/dev/null
cat $OUT/r5o_slow.log | cut -c1-420
