set -x
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
timeout 1700 python -m pytest tests/ -x -q -m gpu > $OUT/r5_suite2.log 2>&1; tail -3 $OUT/r5_suite2.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python bench.py > $OUT/bench_default_r5b.json 2> $OUT/bench_default_r5b.err; tail -2 $OUT/bench_default_r5b.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default_r5b.json"))
print({k:d[k] for k in ("value","ms_per_step","process_runs","best_process_ms")})
s=d["secondary"]
for k in ("c3_f32","c3_amp","c2_amp","c2_f32_shipped_dropout","trainer_loop_c2_f32"):
    print(k, s.get(k,{}).get("ms_per_step"), s.get(k,{}).get("error"))
w=s["decode_wfst_tlg"]; print({k:w["offline"][k] for k in ("search_ms","search_ms_prune_every_25_frames","finalize_gpu_ms","ms_per_utterance","pipelined_ms_per_utterance")}, w["streaming"], w.get("rescore"))
print(json.dumps(s["dp_forced_one_rank"])[:1800])
PY
