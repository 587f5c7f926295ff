#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: per-kernel mean of each counter (per dispatch) and
per-wave values.  usage: pmc_summary.py counter_collection.csv [kernel-substring]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
key = sys.argv[2] if len(sys.argv) > 2 else 'snsde'
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for r in rows:
    if key in r['Kernel_Name']:
        name = r['Kernel_Name'][:110]
        agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
        meta[name] = r
for name, v in agg.items():
    m = meta[name]
    waves = int(m['Grid_Size']) // 64
    print(name)
    print('  grid', m['Grid_Size'], 'wg', m['Workgroup_Size'], 'vgpr', m['VGPR_Count'], 'agpr', m['Accum_VGPR_Count'],
          'sgpr', m['SGPR_Count'], 'lds', m['LDS_Block_Size'], 'scratch', m['Scratch_Size'], 'waves', waves)
    for c, vals in sorted(v.items()):
        mean = sum(vals) / len(vals)
        print(f'  {c:32s} {mean:16.1f}  per-wave {mean / waves:12.1f}  (n={len(vals)})')
