#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_step_parity.py -q -m gpu -k "256_tile or prepacked" 2>&1 | tail -5
