import json, os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd"); sys.path.insert(0, ".")
import bench_secondary as bs
CFG = {"c2": [("0", "", ""), ("1", "1,1", "f"), ("1", "2,1", "f"), ("1", "3,1", "f"), ("1", "2,1", "fb"), ("1", "2,4", "fb"), ("1", "2,6", "fb"), ("1", "2,8", "fb"), ("1", "2,12", "fb"), ("0", "", "")],
       "c3": [("0", "", ""), ("1", "1,1", "f"), ("1", "1,1", "fb"), ("1", "1,4", "fb"), ("0", "", "")]}
only = os.environ.get("R6_AB_SHAPES", "c3,c2").split(",")
for shape in only:
    for wave, chunks, dirs in CFG[shape]:
        os.environ["B2T_WAVE"] = wave
        if chunks: os.environ["B2T_WAVE_CHUNKS"] = chunks
        if dirs: os.environ["B2T_WAVE_DIRS"] = dirs
        try:
            r = bs.train_ms(shape, True)
            print("R6AMP", shape, "wave=" + wave, "chunks=" + chunks, "dirs=" + dirs, r["ms_per_step"], r["window_ms"], flush=True)
        except Exception as e:
            print("R6AMP", shape, "wave=" + wave, chunks, dirs, "ERROR", repr(e)[:300], flush=True)
