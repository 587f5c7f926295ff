#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05N; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fields.py -x -q -m gpu > $O/fields.log 2>&1 < /dev/null; echo "rc=$?" >> $O/fields.log
tail -8 $O/fields.log | cut -c1-220
timeout 600 python tools/time_fields.py > $O/time_fields.txt 2>&1 < /dev/null
grep -v amdgpu $O/time_fields.txt | tail -40
