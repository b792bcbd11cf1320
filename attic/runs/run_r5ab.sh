#!/bin/bash
# A/B of two builds of the same ABI in one call (B2T_LIB): fp32 steps, previous library vs the tree's
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for i in 1 2 3; do
  echo "prev: $(B2T_LIB=$GRAFT_REPO_ROOT/nejm-brain-to-text_amd/csrc/libb2t_hip_prev.so timeout 200 python tools/r4_cfgs.py c2 c3 2>&1 | tail -1)"
  echo "head: $(timeout 200 python tools/r4_cfgs.py c2 c3 2>&1 | tail -1)"
done
