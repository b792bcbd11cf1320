set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_dp_procs.py -x -q -m gpu -s -k "delayed" > $OUT/r5g_skew.log 2>&1; tail -40 $OUT/r5g_skew.log
timeout 600 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu -s -k "graphs" > $OUT/r5g_graph.log 2>&1; tail -15 $OUT/r5g_graph.log
bash tools/run_r5f.sh
