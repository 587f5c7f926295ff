#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04n; mkdir -p $O
cd $R
for cfg in "3 18 2048 64 69 72 srk" "4 17 1024 256 14 50 milstein" "4 17 1024 128 21 101 euler" "6 17 4096 128 21 201 euler"; do
  echo "== $cfg" >> $O/sweep_rc.txt
  for rc in 16 32 64; do
    lib=$R/stable-neural-sdes_amd/libsnsde_rc$rc.so; [ $rc = 32 ] && lib=$R/stable-neural-sdes_amd/libsnsde.so
    for w in 512 1024 1536; do echo -n "RC=$rc " >> $O/sweep_rc.txt; SNSDE_LIB=$lib SNSDE_WGRAD_WGS=$w python tools/time_param_pass.py $cfg 2>/dev/null | grep "param pass" >> $O/sweep_rc.txt; done
  done
done
cat $O/sweep_rc.txt
