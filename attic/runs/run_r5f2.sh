#!/bin/bash
# bf16 steps: the tree's library against csrc/libb2t_hip_alt.so (a build of another gemm_bf16p.hip) in one call
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
ALT=$GRAFT_REPO_ROOT/nejm-brain-to-text_amd/csrc/libb2t_hip_alt.so
for i in 1 2 3; do
  echo "alt : $(B2T_LIB=$ALT timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1)"
  echo "tree: $(timeout 200 python tools/r4_cfgs.py c3_amp c2_amp 2>&1 | tail -1)"
done
