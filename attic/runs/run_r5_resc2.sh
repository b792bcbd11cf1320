#!/bin/bash
# round 5, last session: Rescore() with helper threads for the large lattices (B2T_RESCORE_BIG_THREADS = 1 / 4 / 8) on the box's host
mkdir -p gpurun_out
for k in 1 4 8 1 4 8; do
  B2T_RESCORE_BIG_THREADS=$k timeout 100 python tools/bench_wfst.py > gpurun_out/r5_resc2_t${k}_$RANDOM.json 2> gpurun_out/r5_resc2_err.txt
done
for f in gpurun_out/r5_resc2_t*.json; do echo $f; grep -A4 '"rescore_nbest100_ms_32_utterances"' $f | tr -d '\n '; echo; done
