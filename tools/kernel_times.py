#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel trace CSV: per (kernel, grid) count / average / max duration in microseconds."""
import collections, csv, glob, re, sys
pat = sys.argv[2] if len(sys.argv) > 2 else 'snsde'
f = sys.argv[1]
if not f.endswith('.csv'):
    f = glob.glob(f + '/*/*_kernel_trace.csv')[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if pat in n:
        m = re.search(r'(snsde_\w*kernel)(<[^>]*>+)?', n)
        key = (m.group(0) if m else n[:60]) + ' grid=' + r['Grid_Size_X'] + 'x' + r['Grid_Size_Y'] + 'x' + r['Grid_Size_Z']
        agg[key].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(f'{k:86s} n={len(v):4d} avg={sum(v) / len(v):9.1f} max={max(v):9.1f}')
