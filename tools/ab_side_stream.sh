#!/bin/bash
# A/B of the weight-gradient pass's side stream (diffusion-side reductions beside the GEMMs): training-step times with and without
R=$GRAFT_REPO_ROOT
for k in 0 1 0 1; do
  echo "SNSDE_NO_SIDE_STREAM=$k"
  SNSDE_NO_SIDE_STREAM=$k python $R/tools/time_train.py 2>/dev/null | grep -E "forward\+backward|native snsde_param"
  SNSDE_NO_SIDE_STREAM=$k python $R/bench.py --steps 20 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
e=d['extra']
print('  K2_train fwd+bwd', e['K2_train']['forward_backward']['median_ms'], '| K5_strong_train ms/step', e['K5_strong_train']['ms_per_step'], '| srk (1,18) fwd+bwd', e['NSDE_1_18_srk_H128']['forward_backward']['median_ms'])"
done
