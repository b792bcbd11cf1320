set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 1500 python bench.py > $OUT/bench_default_r5a.json 2> $OUT/bench_default_r5a.err; tail -3 $OUT/bench_default_r5a.err; python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default_r5a.json"))
print({k:d[k] for k in ("value","ms_per_step","process_runs","best_process_ms")})
print(d["roofline"]["kernel"], d["roofline"]["frac"], d["cpu_baseline"])
s=d["secondary"]
for k in ("c3_f32","c3_amp","c2_amp","c2_f32_shipped_dropout","trainer_loop_c2_f32"):
    print(k, s.get(k,{}).get("ms_per_step"), s.get(k,{}).get("error"))
print(json.dumps(s.get("dp_forced_one_rank"))[:1500])
print(json.dumps(s.get("configs3_decode_wfst_3gram"))[:600])
print(json.dumps(s.get("decode_wfst_tlg",{}).get("roofline")))
PY
