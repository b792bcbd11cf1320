import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, "nejm-brain-to-text_amd"), ROOT, os.path.join(ROOT, "tools")):
    sys.path.insert(0, p)
import bench_wfst as BW
prons, words, arpa, g, *_ = BW.make(U=1)
for r in BW.accuracy(prons, words, arpa, g, noise_levels=[float(x) for x in sys.argv[1:]]):
    print(json.dumps(r))
