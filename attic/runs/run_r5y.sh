#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for rep in 1 2 3; do
python - <<'PY'
import sys; sys.path.insert(0,"tools"); sys.path.insert(0,"nejm-brain-to-text_amd")
import bench_secondary as b
out=[]
for sh,amp in (("c3",False),("c3",True),("c2",True),("c2",True)):
    out.append(f"{sh}{'_amp' if amp else ''} {b.train_ms(sh,amp)['ms_per_step']:.3f}")
print(" | ".join(out), flush=True)
PY
done
for rep in 1 2; do
python - <<'PY'
import sys; sys.path.insert(0,"tools"); sys.path.insert(0,"nejm-brain-to-text_amd")
import bench_secondary as b
print("c2_amp alone:", " ".join(f"{b.train_ms('c2',True)['ms_per_step']:.3f}" for _ in range(3)), flush=True)
PY
done
