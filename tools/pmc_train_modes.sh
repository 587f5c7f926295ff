#!/bin/bash
# HBM traffic of a K2 training step (forward + adjoint + weight gradients) in saved-activation and recompute mode:
# FETCH_SIZE / WRITE_SIZE summed over every kernel of 10 steps (separate rocprofv3 --pmc passes).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_train; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
MODES=${MODES:-0 25 10}
for mode in $MODES; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/m${mode}_$c -o run -- python $R/tools/train_steps.py $mode 10 > /dev/null 2>&1
  done
done
python - <<PY
import csv, glob
print('HBM traffic per K2 training step (B=1024, H=128, N=100; sdeint forward + backward), rocprofv3 PMC, 10 steps per pass;')
print('FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md), WRITE_SIZE as reported; KB counters -> MB')
for mode in [int(m) for m in '$MODES'.split()]:
    tot = {}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = glob.glob('$O/m%d_%s/*counter_collection.csv' % (mode, c))
        s = 0.0
        if f:
            for r in csv.DictReader(open(f[0])):
                if r['Counter_Name'] == c: s += float(r['Counter_Value'])
        tot[c] = s / 10 / 1024
    name = 'saved activations' if mode == 0 else 'recompute, %d steps per chunk' % mode
    print('%-32s read %8.1f MB  write %8.1f MB  total %8.1f MB per step' % (name, 2 * tot['FETCH_SIZE'], tot['WRITE_SIZE'], 2 * tot['FETCH_SIZE'] + tot['WRITE_SIZE']))
PY
