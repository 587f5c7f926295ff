#!/usr/bin/env python3
"""Timeline of the LAST training step in a rocprofv3 kernel trace of tools/train_steps.py: every kernel between the last two
forward solve kernels (start offset, duration, gap to the previous kernel's end), in microseconds.
usage: step_timeline.py <trace dir or csv> [forward kernel name pattern]"""
import csv, glob, re, sys
f = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'snsde_m4_kernel'
if not f.endswith('.csv'):
    f = (glob.glob(f + '/*/*_kernel_trace.csv') + glob.glob(f + '/*_kernel_trace.csv'))[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))))
fw = [i for i, r in enumerate(rows) if pat in r[2]]
# the step's first kernel: the launch after the previous step's last backward kernel = first kernel after the previous forward's tail;
# print from the second-to-last forward kernel's successor chain: previous forward .. this forward exclusive gives one full period
a, b = fw[-2], fw[-1]
t0 = rows[a][0]
prev_end = None
tot = 0
for s, e, n in rows[a:b]:
    m = re.search(r'(snsde_\w+|at::native::\w+|__amd_\w+)', n)
    name = m.group(0) if m else n[:50]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f'{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:8.1f} us  gap {gap:7.1f}  {name}')
    prev_end = max(e, prev_end or e)
    tot += (e - s) / 1e3
print(f'period {(rows[b][0] - t0) / 1e3:.1f} us, kernel time {tot:.1f} us, {b - a} launches')
