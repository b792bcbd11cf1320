#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_step_parity.py tests/test_gpu_trainer.py tests/test_gpu_fuzz.py -q -m gpu -x 2>&1 | tail -3
for i in 1 2; do timeout 200 python tools/r4_cfgs.py c3_amp c2_amp c2 2>&1 | tail -1; done
