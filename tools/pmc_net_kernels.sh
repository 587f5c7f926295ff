#!/bin/bash
# PMC counters of the diffusion-net kernels (snsde_m4n_kernel, snsde_m4n_srk_reverse_kernel, snsde_m4n_mil_reverse_kernel) at the
# K4 shape and the (1,18) H = 128 shape: one rocprofv3 --pmc pass per counter over tools/net_steps.py (same counter set as
# profiles/r03_pmc_counters.txt), then kernel stats of the same run and the per-call host share.
# usage: pmc_net_kernels.sh <out tag> [counters...]
tag=${1:-r04}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_net_$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CNT=${@:-GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE FETCH_SIZE WRITE_SIZE}
for c in $CNT; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -o run -- python $R/tools/net_steps.py 3 > $O/$c.log 2>&1
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python $R/tools/net_steps.py 10 > $O/stats.log 2>&1
cp $(find $O/stats -name '*kernel_stats.csv' | head -1) $O/${tag}_net_kernel_stats.csv
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
        if 'snsde_m4n' in k or 'snsde_w4' in k:
            key = (k[:120], r['Grid_Size'])
            vals[(key, r['Counter_Name'])].append(float(r['Counter_Value']))
            meta[key] = (r['Workgroup_Size'], r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'])
    for (key, c), v in vals.items():
        agg[key][c] = (sum(v) / len(v), len(v))
with open(os.path.join(O, tag + '_pmc_net_kernels.txt'), 'w') as out:
    print('PMC counters of the diffusion-net kernels (one rocprofv3 --pmc pass per counter over tools/net_steps.py 3), mean per dispatch.', file=out)
    print('FETCH_SIZE / WRITE_SIZE are the raw KB counters (FETCH_SIZE x 2 on gfx950 per MI355X_MICROARCH.md for bytes).', file=out)
    for key, cs in agg.items():
        wg, vg, ag, lds, scr = meta[key]
        waves = int(key[1]) // 64
        print(f'\n{key[0]}\n  grid {key[1]} wg {wg} vgpr {vg} agpr {ag} lds {lds} scratch {scr} waves {waves}', file=out)
        for c, (m, n) in sorted(cs.items()):
            print(f'  {c:28s} {m:16.1f}  per-wave {m / waves:12.1f}  (n={n})', file=out)
        g = cs.get('GRBM_GUI_ACTIVE', (0, 0))[0]; wc = cs.get('SQ_WAVE_CYCLES', (0, 0))[0]
        if wc:
            f = lambda c: cs.get(c, (0, 0))[0]
            # SQ_*_CYCLES counters tick in units of 4 cycles per the guide; ratios between SQ counters are unit-free
            print(f'  -> WAIT_ANY / WAVE_CYCLES {f("SQ_WAIT_ANY") / wc:.3f}; WAIT_INST_ANY / WAVE_CYCLES {f("SQ_WAIT_INST_ANY") / wc:.3f}; '
                  f'WAIT_INST_LDS / WAVE_CYCLES {f("SQ_WAIT_INST_LDS") / wc:.3f}; ACTIVE_INST_VALU / WAVE_CYCLES {f("SQ_ACTIVE_INST_VALU") / wc:.3f}; '
                  f'MFMA_BUSY / BUSY_CYCLES {f("SQ_VALU_MFMA_BUSY_CYCLES") / max(f("SQ_BUSY_CYCLES"), 1):.3f}; '
                  f'VALU (non-MFMA) per MFMA {(f("SQ_INSTS_VALU") - f("SQ_INSTS_MFMA")) / max(f("SQ_INSTS_MFMA"), 1):.2f}; '
                  f'LDS_BANK_CONFLICT / LDS_IDX_ACTIVE {f("SQ_LDS_BANK_CONFLICT") / max(f("SQ_LDS_IDX_ACTIVE"), 1):.3f}', file=out)
print(open(os.path.join(O, tag + '_pmc_net_kernels.txt')).read())
PY
python $R/tools/net_steps.py time > $O/${tag}_net_host_share.txt 2>&1
cat $O/${tag}_net_host_share.txt
grep -E 'm4n|snsde_w4' $O/${tag}_net_kernel_stats.csv | cut -c1-200
