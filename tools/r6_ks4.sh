#!/bin/bash
# bf16-mode step at C2: gated consumer chunks of the backward wavefront
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_ks4; mkdir -p $O
cat > /tmp/ab.py <<'PY'
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd"); sys.path.insert(0, ".")
import bench_secondary as bs
for shape, chunks in (("c2", "1,1"), ("c2", "1,2"), ("c2", "1,4"), ("c2", "1,8"), ("c2", "2,1"), ("c2", "1,1")):
    os.environ["B2T_WAVE_CHUNKS"] = chunks
    try:
        r = bs.train_ms(shape, True)
        print("R6AMP", shape, "chunks=" + chunks, r["ms_per_step"], r["window_ms"], flush=True)
    except Exception as e:
        print("R6AMP", shape, "chunks=" + chunks, "ERROR", repr(e)[:300], flush=True)
PY
timeout 1200 python /tmp/ab.py 2>$O/ab.err | grep R6AMP | tee $O/summary.txt; tail -3 $O/ab.err
