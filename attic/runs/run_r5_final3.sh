set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
mkdir -p $OUT
# does the step test reach the 256-tile kernel?  (kernel names of one run of it)
rm -rf /tmp/k256; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/k256 -- python -m pytest tests/test_gpu_step_parity.py -q -m gpu -k "256_tile" > /tmp/k256.log 2>&1
grep -h "gemm_bf16p_kernel" $(find /tmp/k256 -name "*kernel_stats.csv") | cut -c1-200 > $OUT/r5_k256_in_step.txt; cat $OUT/r5_k256_in_step.txt
timeout 1700 python -m pytest tests/ -x -q -m gpu > $OUT/r5_suite7.log 2>&1; tail -3 $OUT/r5_suite7.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1200 python bench.py > $OUT/bench_default_r5h.json 2> $OUT/bench_default_r5h.err; tail -2 $OUT/bench_default_r5h.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench_default_r5h.json"))
print({k:d[k] for k in ("value","ms_per_step","process_runs","best_process_ms")})
s=d["secondary"]
print({k: s[k].get("ms_per_step") for k in ("c3_f32","c3_amp","c2_amp","c2_f32_shipped_dropout","trainer_loop_c2_f32")})
w=s["decode_wfst_tlg"]; print({k:w["offline"][k] for k in ("search_ms","search_ms_prune_every_25_frames","finalize_gpu_ms","ms_per_utterance","pipelined_ms_per_utterance")}, w["streaming"], w.get("rescore"))
PY
