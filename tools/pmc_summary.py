"""Summarise rocprofv3 --pmc CSVs: usage pmc_summary.py <kernel-substring> <trace_dir> <pmc_dir>..."""
import csv, collections, re, sys
sub, trace, pmcs = sys.argv[1], sys.argv[2], sys.argv[3:]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f'{trace}/r_kernel_trace.csv')):
    if sub in r['Kernel_Name']:
        k = (re.sub(r'.*mst\d*', '', r['Kernel_Name'])[:30], r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Workgroup_Size_X', ''))
        d[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in d.items():
    print(k, len(v), f"min {min(v):.1f} med {sorted(v)[len(v)//2]:.1f} us")
res = collections.defaultdict(dict)
for p in pmcs:
    for r in csv.DictReader(open(f'{p}/r_counter_collection.csv')):
        if sub not in r['Kernel_Name']:
            continue
        k = (re.sub(r'.*mst\d*', '', r['Kernel_Name'])[:30], r['Grid_Size'], r.get('Workgroup_Size', ''))
        res[k][r['Counter_Name']] = float(r['Counter_Value'])
for k, v in res.items():
    wc = v.get('SQ_WAVE_CYCLES', 1.0)
    print(k, ' '.join(f"{n[3:]}={x/wc:.3f}" for n, x in v.items() if n not in ('SQ_WAVES', 'SQ_WAVE_CYCLES')),
          f"wavecyc/wave={wc/v.get('SQ_WAVES',1):.0f} valu/wave={v.get('SQ_INSTS_VALU',0)/v.get('SQ_WAVES',1):.0f}")
