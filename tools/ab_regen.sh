#!/bin/bash
# A/B of the adjoint's increment regeneration at K2: kernel stats of 20 training steps with and without SNSDE_KEEP_INCREMENTS
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for k in 0 1; do
  SNSDE_KEEP_INCREMENTS=$k rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_regen_$k -o run -- python $R/tools/train_steps.py 0 20 > /dev/null 2>&1
  echo "SNSDE_KEEP_INCREMENTS=$k"
  grep -E "m4_kernel|reverse_kernel|wgrad_kernel\(" $(find $R/gpurun_out/ab_regen_$k -name '*kernel_stats.csv' | head -1) | awk -F'",' '{print $1}' | cut -c1-90 | paste - <(grep -E "m4_kernel|reverse_kernel|wgrad_kernel\(" $(find $R/gpurun_out/ab_regen_$k -name '*kernel_stats.csv' | head -1) | awk -F'",' '{print $2}' | cut -d, -f1-3)
done
