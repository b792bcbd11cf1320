#!/bin/bash
# shipped shape, bf16 mode: time chunks of the forward / backward plan (the 3 / 2 default dates from round 3, when the step took 7.4 ms)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for cfg in ${CFGS:-"3 2" "3 3" "4 2" "4 3" "4 4" "2 2" "5 3" "6 3" "6 4" "3 2"}; do
  set -- $cfg
  echo "chunks $1 / $2: $(B2T_CHUNKS=$1 B2T_CHUNKS_BWD=$2 timeout 200 python tools/r4_cfgs.py c3_amp 2>&1 | tail -1)"
done
echo "default: $(timeout 200 python tools/r4_cfgs.py c3_amp 2>&1 | tail -1)"
