#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04k; mkdir -p $O
cd $R
for cfg in "3 18 2048 64 69 72 srk" "4 17 1024 256 14 50 milstein" "4 17 1024 128 21 101 euler" "6 17 4096 128 21 201 euler"; do
  echo "== $cfg" >> $O/sweep_wgrad.txt
  for b in 2 4 8; do for w in 256 512 768 1024; do SNSDE_WGRAD_BIAS=$b SNSDE_WGRAD_WGS=$w python tools/time_param_pass.py $cfg 2>/dev/null | grep "param pass" >> $O/sweep_wgrad.txt; done; done
done
cat $O/sweep_wgrad.txt
