import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd")
import bench_secondary as b
for sh in ("c3", "c2"):
    r = b.train_ms(sh, True, steps=20, warmup=5)
    print(sh, "amp", r["ms_per_step"])
