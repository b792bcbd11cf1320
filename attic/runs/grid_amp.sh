cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python - <<'PY'
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd")
import bench_secondary as S
print(S.train_ms("c2", True, steps=30, warmup=5)["ms_per_step"])
PY
}
for f in 6 8 10; do for b in 4 5 6 8; do echo "c2 amp chunks $f/$b: $(run B2T_CHUNKS=$f B2T_CHUNKS_BWD=$b)"; done; done
echo "default: $(run A=1)"
