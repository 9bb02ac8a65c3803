#!/usr/bin/env python3
"""Per-kernel duration statistics from a rocprofv3 --kernel-trace --output-format csv directory:
count, total, min / p10 / median / p90 / max per launch, the idle gaps between kernels, and the durations of a few
consecutive trace and shade launches in launch order (one sample pass is four trace and five shade launches).
Usage: python tools/ktrace_summary.py <output dir of rocprofv3>"""
import csv, sys, glob, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
by = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0]
    for k in ('k_wave_trace', 'k_wave_shade', 'k_wave_resolve', 'k_raygen', 'k_finalize', 'fill'):
        if k in name: name = k
    by[name].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    v2 = sorted(v)
    print(f"{k[:40]:40s} n {len(v):5d} sum {sum(v)/1e3:8.2f} ms  min {v2[0]:7.1f} p10 {v2[len(v)//10]:7.1f} med {v2[len(v)//2]:7.1f} p90 {v2[len(v)*9//10]:7.1f} max {v2[-1]:7.1f} us")
# gaps
rows.sort(key=lambda r: int(r['Start_Timestamp']))
gap = 0; prev = None
for r in rows:
    if prev is not None:
        g = int(r['Start_Timestamp']) - prev
        if 0 < g < 1e6: gap += g
    prev = max(prev or 0, int(r['End_Timestamp']))
print("idle gaps between kernels (<1ms each): %.2f ms over %d kernels" % (gap / 1e6, len(rows)))
tr = [r for r in rows if 'k_wave_trace' in r['Kernel_Name']]
tr.sort(key=lambda r: int(r['Start_Timestamp']))
print("trace launches 200..216 (us):", [round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in tr[200:216]])
sh = [r for r in rows if 'k_wave_shade' in r['Kernel_Name']]
sh.sort(key=lambda r: int(r['Start_Timestamp']))
print("shade launches 200..215 (us):", [round((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3) for r in sh[200:215]])
