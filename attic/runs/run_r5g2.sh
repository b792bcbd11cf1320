#!/bin/bash
# knobs re-swept on the round's last kernels (bf16 steps)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for e in "B2T_X=0" "B2T_WGRAD_SPLIT=1" "B2T_SPLITK256=1" "B2T_GI0_CHAIN=1" "B2T_SPLITK_TARGET=384" "B2T_SPLITK_TARGET=768" "B2T_X=0"; do
  echo "c3_amp $e: $(env $e timeout 200 python tools/r4_cfgs.py c3_amp 2>&1 | tail -1)"
done
for cfg in "6 4" "6 6" "8 4" "5 4" "6 3" "4 4" "6 4"; do
  set -- $cfg
  echo "c2_amp chunks $1 / $2: $(B2T_CHUNKS=$1 B2T_CHUNKS_BWD=$2 timeout 200 python tools/r4_cfgs.py c2_amp 2>&1 | tail -1)"
done
