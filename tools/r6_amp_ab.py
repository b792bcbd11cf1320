import json, os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd"); sys.path.insert(0, ".")
import bench_secondary as bs
for shape in ("c3", "c2"):
    for wave in ("1", "0", "1"):
        os.environ["B2T_WAVE"] = wave
        try:
            r = bs.train_ms(shape, True)
            print("R6AMP", shape, "wave=" + wave, r["ms_per_step"], r["window_ms"], flush=True)
        except Exception as e:
            print("R6AMP", shape, "wave=" + wave, "ERROR", repr(e)[:300], flush=True)
