#!/bin/bash
# PMC counters of the K2 training step's kernels (training-mode forward, adjoint, weight-gradient GEMMs): one rocprofv3 --pmc pass per
# counter over tools/train_steps.py.  usage: pmc_train_kernels.sh <counters...>
R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
  out=$R/gpurun_out/pmc_train/$c; mkdir -p $out
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $out -o run -- python $R/tools/train_steps.py 0 6 > $out/log.txt 2>&1
  f=$(find $out -name '*counter_collection.csv' | head -1)
  python - "$f" "$c" <<'PY'
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']
    for key in ('snsde_m4_kernel', 'snsde_mfma_reverse_kernel', 'snsde_wgrad_kernel'):
        if key in k and r['Counter_Name'] == sys.argv[2]:
            agg[key].append(float(r['Counter_Value']))
print(sys.argv[2], ' | '.join(f'{k}: {sum(v) / len(v):.4g} (n={len(v)})' for k, v in agg.items()))
PY
done
