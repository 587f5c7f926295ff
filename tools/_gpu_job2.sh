#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04o; mkdir -p $O
cd $R
python tools/sweep_batch_h.py 64 > $O/sweep_h.txt 2>&1
python tools/sweep_batch_h.py 32 >> $O/sweep_h.txt 2>&1
python tools/sweep_batch_h.py 128 6 17 >> $O/sweep_h.txt 2>&1
python tools/sweep_batch_h.py 256 >> $O/sweep_h.txt 2>&1
grep -v amdgpu $O/sweep_h.txt
