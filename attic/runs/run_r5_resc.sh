#!/bin/bash
# round 5, last session: Rescore() A/B (library before / after the rank-numbered determinisation) on the box's host, the decode
# tests that reach it, smoke
mkdir -p gpurun_out
for i in 1 2; do
  B2T_LIB=$PWD/tools/ab_base/libb2t_base.so timeout 300 python tools/bench_wfst.py > gpurun_out/r5_resc_base_$i.json 2> gpurun_out/r5_resc_base_$i.err
  timeout 300 python tools/bench_wfst.py > gpurun_out/r5_resc_new_$i.json 2> gpurun_out/r5_resc_new_$i.err
done
grep -h -A2 '"rescore_nbest100_ms_32_utterances"' gpurun_out/r5_resc_base_*.json gpurun_out/r5_resc_new_*.json
timeout 600 python -m pytest tests/test_gpu_wfst.py tests/test_gpu_decoder.py -x -q -m gpu 2>&1 | tail -3
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
nproc; lscpu | grep "Model name"
