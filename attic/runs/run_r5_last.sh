#!/bin/bash
# round 5, last session: validation of the tree the round ends on (full GPU suite, smoke, Rescore() numbers, default bench)
mkdir -p gpurun_out
timeout 560 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r5_last_pytest.txt
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/r5_last_smoke.txt
for i in 1 2; do timeout 120 python tools/bench_wfst.py > gpurun_out/r5_last_wfst_$i.json 2> gpurun_out/r5_last_wfst_$i.err; done
grep -h -A4 '"rescore_nbest100_ms_32_utterances"' gpurun_out/r5_last_wfst_*.json
timeout 330 python bench.py > gpurun_out/r5_last_bench.json 2> gpurun_out/r5_last_bench.err; tail -c 600 gpurun_out/r5_last_bench.json
