#!/bin/bash
# PMC counters of the H = 256 forward kernels (two-tile resident snsde_m4s2_kernel vs fully streamed snsde_m4s_kernel) at the K5 shape:
# one rocprofv3 --pmc pass per counter over tools/k5_forward_steps.py.  usage: pmc_k5.sh <tag>
tag=${1:-r06}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_k5_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SALU FETCH_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o run -- python $R/tools/k5_forward_steps.py 5 > $O/$c.log 2>&1
done
python - "$O" "$tag" <<'PY'
import csv, glob, sys, collections, os
O, tag = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(dict)
meta = {}
for d in sorted(glob.glob(O + '/*/')):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        continue
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        k = r['Kernel_Name']
        if 'snsde_m4s' in k:
            key = k[:90]
            vals[(key, r['Counter_Name'])].append(float(r['Counter_Value']))
            meta[key] = (r['Grid_Size'], r['Workgroup_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'])
    for (key, c), v in vals.items():
        agg[key][c] = (sum(v) / len(v), len(v))
with open(os.path.join(O, tag + '_pmc_k5.txt'), 'w') as out:
    print('PMC counters of the H = 256 forward kernels at the K5 shape (1024 rows, 49 Milstein steps; one rocprofv3 --pmc pass per counter', file=out)
    print('over tools/k5_forward_steps.py 5), mean per dispatch.  SQ_* cycle counters are in units of 4 cycles summed over waves / SIMDs.', file=out)
    for key, cs in agg.items():
        g, wg, vg, ag, lds = meta[key]
        print(f'\n{key}\n  grid {g} wg {wg} vgpr {vg} agpr {ag} lds {lds}', file=out)
        for c, (m, n) in sorted(cs.items()):
            print(f'  {c:28s} {m:16.1f}   n={n}', file=out)
print(open(os.path.join(O, tag + '_pmc_k5.txt')).read())
PY
