#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05K; mkdir -p $O
timeout 300 python tools/prof_host_step.py euler > $O/host_prof.txt 2>&1 < /dev/null
grep -v amdgpu $O/host_prof.txt | head -110
