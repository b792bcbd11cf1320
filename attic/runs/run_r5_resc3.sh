#!/bin/bash
# round 5, last session: final library -- Rescore() with 1 / 4 threads for the large lattices, the GPU tests that reach it, smoke
mkdir -p gpurun_out
for k in 1 4 1 4; do
  f=gpurun_out/r5_resc3_t${k}_$RANDOM.json
  B2T_RESCORE_BIG_THREADS=$k timeout 100 python tools/bench_wfst.py > $f 2> gpurun_out/r5_resc3_err.txt
  echo -n "big_threads $k: "; grep -A4 '"rescore_nbest100_ms_32_utterances"' $f | tr -d '\n '; echo
done
timeout 150 python -m pytest tests/test_gpu_wfst.py -x -q -m gpu -k "rescore or Rescore or nbest" 2>&1 | tail -2
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
