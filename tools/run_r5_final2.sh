set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 1700 python -m pytest tests/ -x -q -m gpu > $OUT/r5_suite3.log 2>&1; tail -2 $OUT/r5_suite3.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
/usr/bin/time -v timeout 1200 python bench.py > $OUT/bench_default_r5c.json 2> $OUT/bench_default_r5c.err; grep -E "Elapsed|headline" $OUT/bench_default_r5c.err | cut -c1-200
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default_r5c.json"))
print({k:d[k] for k in ("value","ms_per_step","process_runs","best_process_ms")}, d["box"]["host_api_us"])
s=d["secondary"]
print({k:(v.get("ms_per_step"), v.get("host_enqueue_ms_per_step")) for k,v in s["dp_forced_one_rank"].items() if isinstance(v, dict)})
PY
