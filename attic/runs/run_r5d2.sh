#!/bin/bash
# bf16 mode: weight gradients with few output tiles on the 256-tile kernel with slices filling one round (B2T_SPLITK256, read per pass)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for e in "B2T_SPLITK256=0" "B2T_SPLITK256=1" "B2T_SPLITK256=0" "B2T_SPLITK256=1"; do
  echo "== $e: $(env $e timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1)"
done
timeout 900 python -m pytest tests/test_gpu_step_parity.py tests/test_gpu_trainer.py -q -m gpu -x 2>&1 | tail -3
