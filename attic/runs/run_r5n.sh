set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_wfst.py -x -q -m gpu -k "cluster or prune or production or 5gram" 2>&1 | tail -4
for e in "B2T_WFST_PRUNE_CLUSTER=1"; do
env $e timeout 300 python tools/bench_wfst.py 2>/dev/null | python -c "
import json,sys
t=sys.stdin.read(); d=json.loads(t[t.index('{'):]); o=d['offline']; print('$e', {k:o[k] for k in ('search_ms','search_ms_prune_every_25_frames','finalize_gpu_ms','ms_per_utterance','pipelined_ms_per_utterance')}, {k:d['streaming'][k] for k in ('prune_pass_ms_between_frames','max_ms_per_frame_prune_inside')}, d['streaming_word_5gram']['prune_pass_ms_between_frames'])"
done
