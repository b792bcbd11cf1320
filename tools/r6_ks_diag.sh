#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r6_ks_diag; mkdir -p $O; : > $O/summary.txt
for v in _fullbar; do
  if [ -z "$v" ]; then unset B2T_LIB; else export B2T_LIB=$PWD/nejm-brain-to-text_amd/csrc/libb2t_hip_ks$v.so; fi
  timeout 600 python tools/r6_ks_diag.py > $O/diag$v.txt 2>&1; grep KSDIAG $O/diag$v.txt | cut -c1-400 | tee -a $O/summary.txt; tail -2 $O/diag$v.txt | grep -v KSDIAG
done
