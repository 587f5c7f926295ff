#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05J; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_w4.py -x -q -m gpu -k "shortest" > $O/w4_tests.log 2>&1 < /dev/null; echo "rc=$?" >> $O/w4_tests.log
tail -25 $O/w4_tests.log
