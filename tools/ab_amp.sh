cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 200 python - <<'PY'
import sys
sys.path.insert(0, "tools"); sys.path.insert(0, "nejm-brain-to-text_amd")
import bench_secondary as S
print(S.train_ms("c2", True, steps=20, warmup=5)["ms_per_step"])
PY
}
echo "c2 amp 6/4:  $(run A=1)"
echo "c2 amp 4/3:  $(run B2T_CHUNKS=4 B2T_CHUNKS_BWD=3)"
echo "c2 amp 8/4:  $(run B2T_CHUNKS=8 B2T_CHUNKS_BWD=4)"
echo "c2 amp 6/3:  $(run B2T_CHUNKS=6 B2T_CHUNKS_BWD=3)"
echo "c2 amp 6/6:  $(run B2T_CHUNKS=6 B2T_CHUNKS_BWD=6)"
echo "c2 amp 8/6:  $(run B2T_CHUNKS=8 B2T_CHUNKS_BWD=6)"
echo "c2 amp 10/8: $(run B2T_CHUNKS=10 B2T_CHUNKS_BWD=8)"
echo "c2 amp 6/4:  $(run A=1)"
