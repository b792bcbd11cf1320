#!/usr/bin/env python3
"""More seeds of tests/test_gpu_wfst.py's random-graph fuzz (search + cluster prune passes + cluster finalize against the oracle) and of
the binding-regime test's graphs: a soak for the round-5 cluster kernels.  usage: r5_wfst_fuzz_more.py [first_seed=1] [n=10]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "nejm-brain-to-text_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import test_gpu_wfst as TW
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bad = []
for seed in range(first, first + n):
    try:
        TW.test_wfst_search_matches_oracle_on_random_graphs_and_options(seed * 7 + 1000)
        print(f"seed {seed}: ok", flush=True)
    except AssertionError as e:
        bad.append(seed); print(f"seed {seed}: MISMATCH {str(e)[:300]}", flush=True)
print("RESULT:", "clean" if not bad else f"mismatches at seeds {bad}")
sys.exit(1 if bad else 0)
