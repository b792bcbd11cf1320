"""ms/step of the secondary training shapes under the current environment: r4_cfgs.py [c3 c2 c2_amp ...] (default: fp32 and bf16 of both)."""
import sys, os
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd")
import bench_secondary as b
shapes = sys.argv[1:] or ["c3", "c3_amp", "c2", "c2_amp"]
out = []
for sh in shapes:
    amp = sh.endswith("_amp")
    r = b.train_ms(sh[:2], amp, steps=20, warmup=5)
    out.append(f"{sh} {r['ms_per_step']:.3f}")
print(" | ".join(out))
