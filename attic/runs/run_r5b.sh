# Round-5 second GPU call: WFST binding regime / 5-gram streamed tests, graph replay vs eager repeated (flat and child-graph forms).
set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_wfst.py -x -q -m gpu -s -k "binds or 5gram or random_graphs" > $OUT/r5b_wfst.log 2>&1; tail -15 $OUT/r5b_wfst.log
grep -E "^max_active|5-gram graph" $OUT/r5b_wfst.log | tail -60
for i in 1 2 3 4 5 6; do
  B2T_EXEC_GRAPH=0 timeout 120 python tools/r4_graph_probe.py 12 2>&1 | grep "^graph=" >> $OUT/r5b_graph.log
  B2T_EXEC_GRAPH=1 timeout 120 python tools/r4_graph_probe.py 12 2>&1 | grep "^graph=" >> $OUT/r5b_graph.log
  B2T_EXEC_GRAPH=1 B2T_EXEC_GRAPH_FLAT=0 timeout 120 python tools/r4_graph_probe.py 12 2>&1 | grep "^graph=" | sed 's/^graph=1/graph=1child/' >> $OUT/r5b_graph.log
done
cat $OUT/r5b_graph.log
