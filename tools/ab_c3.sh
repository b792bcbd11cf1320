cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_step_parity.py tests/test_gpu_trainer.py -m gpu -q -x 2>&1 | tail -3
python - <<'PY'
import sys, json
sys.path.insert(0, 'tools'); sys.path.insert(0, 'nejm-brain-to-text_amd')
import bench_secondary as S
for name, fn in (("c3_f32", lambda: S.train_ms("c3", False)), ("c3_amp", lambda: S.train_ms("c3", True))):
    print(name, fn()["ms_per_step"])
PY
