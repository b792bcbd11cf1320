set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
T0=$SECONDS
timeout 1200 python bench.py > $OUT/bench_default_r5c.json 2> $OUT/bench_default_r5c.err; echo "bench.py wall: $((SECONDS - T0)) s"
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default_r5c.json"))
print({k:d[k] for k in ("value","ms_per_step","process_runs","best_process_ms")}, d["box"]["host_api_us"])
s=d["secondary"]
print({k:(v.get("ms_per_step"), v.get("host_enqueue_ms_per_step")) for k,v in s["dp_forced_one_rank"].items() if isinstance(v, dict)})
print({k: s[k].get("ms_per_step") for k in ("c3_f32","c3_amp","c2_amp","c2_f32_shipped_dropout","trainer_loop_c2_f32")})
PY
