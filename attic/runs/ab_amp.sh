# usage: ab_amp.sh ENV=VAL ... : bf16-operand training step (C2 and C3 shapes) with and without the given environment
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python - <<'PY'
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd")
import bench_secondary as S
print(S.train_ms("c2", True, steps=30, warmup=5)["ms_per_step"], S.train_ms("c3", True, steps=30, warmup=5)["ms_per_step"], S.train_ms("c3", False, steps=20, warmup=4)["ms_per_step"])
PY
}
for r in 1 2 3; do echo "(c2 amp, c3 amp, c3 f32) default: $(run A=1)   with $*: $(run "$@")"; done
