#!/bin/bash
# WFST search: A/B of two libraries (the tree's and csrc/libb2t_hip_alt.so, a compile-time variant) in one call
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
ALT=$GRAFT_REPO_ROOT/nejm-brain-to-text_amd/csrc/libb2t_hip_alt.so
for i in 1 2; do
  for lib in "" "$ALT"; do
    B2T_LIB=$lib timeout 300 python - <<'PY'
import json, os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd")
import bench_wfst as B
r = B.run()
o = r["offline"]; st = r["streaming"]
print("early-link" if not os.environ.get("B2T_LIB") else "link-after-claim", "search", o["search_ms"], "with prune", o["search_ms_prune_every_25_frames"], "stream p50", st["p50_ms_per_frame"], flush=True)
PY
  done
done
timeout 900 python -m pytest tests/test_gpu_wfst.py -q -m gpu -x 2>&1 | tail -2
