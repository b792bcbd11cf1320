#!/bin/bash
# Round 6: validation of the tree -- GPU suite with durations, smoke, default bench (with secondary + cpu baseline), profiles.
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_full; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q --durations=25 > $O/pytest_gpu.txt 2>&1; echo "pytest gpu rc $?" | tee $O/summary.txt
tail -40 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; echo "smoke rc $?" | tee -a $O/summary.txt; tail -2 $O/smoke.txt
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?" | tee -a $O/summary.txt
python - <<'PY' | tee -a $O/summary.txt
import json
try:
    d=json.loads([l for l in open('gpurun_out/r6_full/bench_default.json') if l.startswith('{')][-1])
    print('headline', d['ms_per_step'], d['value'], 'enq', d['config']['host_enqueue_ms_per_step'], 'roof', d['roofline']['frac'], d['roofline'].get('bf16'))
    s=d.get('secondary',{})
    for k in ('c3_f32','c3_amp','c2_amp','c2_f32_shipped_dropout','trainer_loop_c2_f32','trainer_loop_c3_amp'):
        print(k, (s.get(k) or {}).get('ms_per_step'), (s.get(k) or {}).get('error'))
    dp=s.get('dp_forced_one_rank',{})
    for k,v in dp.items():
        if isinstance(v,dict): print('dp', k, v.get('ms_per_step'), v.get('host_enqueue_ms_per_step'))
    print('cpu', d.get('cpu_baseline'))
    for k in ('configs3_decode_wfst_3gram','configs4_stream_wfst'):
        print(k, json.dumps(s.get(k))[:300])
except Exception as e: print('ERR', e)
PY
