#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05Q; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fields.py -q -m gpu -k "training_step_fused_vs_fp64_autograd and nsde" > $O/fields.log 2>&1 < /dev/null; echo "rc=$?" >> $O/fields.log
tail -30 $O/fields.log | cut -c1-260
