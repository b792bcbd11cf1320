#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_rgf2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_wave.py -x -q -s > $O/pytest_wave.txt 2>&1; echo "pytest wave rc $?" | tee $O/summary.txt
tail -5 $O/pytest_wave.txt
export R6_PROBE_ONLY_TIMING=1
B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_wtiming.so timeout 600 python tools/r6_wave_probe.py > $O/probe_timing.txt 2>&1; grep "R6WAVE" $O/probe_timing.txt | tee -a $O/summary.txt; tail -2 $O/probe_timing.txt
unset R6_PROBE_ONLY_TIMING
cat > /tmp/ab.py <<'PY'
import os, sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd"); sys.path.insert(0, ".")
import bench_secondary as bs
for shape, rgf, dirs, chunks in (("c2", "1", "f", "1,1"), ("c2", "1", "fb", "1,1"), ("c2", "0", "fb", "1,1"), ("c2", "1", "fb", "1,4"), ("c2", "1", "b", "1,1"), ("c2", "1", "f", "1,1")):
    os.environ["B2T_WAVE_RGF"] = rgf; os.environ["B2T_WAVE_DIRS"] = dirs; os.environ["B2T_WAVE_CHUNKS"] = chunks
    r = bs.train_ms(shape, True)
    print("R6AMP", shape, "rgf=" + rgf, "dirs=" + dirs, "chunks=" + chunks, r["ms_per_step"], r["window_ms"], flush=True)
PY
timeout 900 python /tmp/ab.py 2>$O/ab.err | grep R6AMP | tee -a $O/summary.txt; tail -3 $O/ab.err
export B2T_BENCH_NO_RESTART=1 B2T_STEP_HOST_TIMING=1
for i in 1 2 3 4 5 6; do python bench.py --no-secondary --no-cpu-baseline --steps 40 --warmup 6 > $O/b.json 2> $O/b.err; python - <<PY | tee -a $O/summary.txt
import json
try:
    d=json.loads([l for l in open('$O/b.json') if l.startswith('{')][-1]); print('HOSTSEG', d['ms_per_step'], 'enq', d['config']['host_enqueue_ms_per_step'], d['config'].get('step_host_ms'))
except Exception as e: print('HOSTSEG ERR', e, open('$O/b.err').read()[-300:])
PY
done
