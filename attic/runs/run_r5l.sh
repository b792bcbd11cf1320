set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
timeout 900 python -m pytest tests/test_gpu_wfst.py -x -q -m gpu -k "cluster or prune or production or overflow" 2>&1 | tail -8
B2T_WFST_FIN_CLUSTER=0 timeout 300 python tools/bench_wfst.py 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read(); d=json.loads(t[t.index('{'):]); o=d['offline']; print('FIN_CLUSTER=0', {k:o[k] for k in ('search_ms','finalize_gpu_ms','nbest100_host_ms','ms_per_utterance','pipelined_ms_per_batch','pipelined_ms_per_utterance')})"
timeout 300 python tools/bench_wfst.py 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read(); d=json.loads(t[t.index('{'):]); o=d['offline']; print('FIN_CLUSTER=1', {k:o[k] for k in ('search_ms','finalize_gpu_ms','nbest100_host_ms','ms_per_utterance','pipelined_ms_per_batch','pipelined_ms_per_utterance')})"
