cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python - <<'PY'
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd")
import bench_secondary as S
print(S.train_ms("c2", True, steps=30, warmup=5)["ms_per_step"])
PY
}
for r in 1 2 3; do echo "c2 amp device-scope: $(run A=1)   xcd-local fb: $(run B2T_GRU_LOCAL_AMP=fb)   b only: $(run B2T_GRU_LOCAL_AMP=b)"; done
