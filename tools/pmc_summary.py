#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel of this library.
Usage: tools/pmc_summary.py DIR  -> prints a table, also as CSV to stdout."""
import csv, collections, glob, re, sys
d = sys.argv[1]
KERN = ('k_normal_fused', 'k_intr_normal', 'k_cross_schur', 'k_peak_read4', 'k_peak_read', 'k_finalize', 'k_pose_refine')
print('kernel,counter,launches,mean,min,max,mean_duration_us')
for f in sorted(glob.glob(d + '/*counter_collection.csv')):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        m = re.search(r'(k_[a-z_0-9]+)(<[^>]*>)?', k)  # the template arguments tell the instantiations apart
        if 'anonymous namespace' not in k or not m or m.group(1) not in KERN:
            continue
        name = m.group(1) + (m.group(2) or '')
        dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        agg.setdefault((name, r['Counter_Name']), []).append((float(r['Counter_Value']), dur))
    for (k, c), v in agg.items():
        vals = [x[0] for x in v]
        print(f'"{k}",{c},{len(v)},{sum(vals)/len(vals):.6e},{min(vals):.6e},{max(vals):.6e},{sum(x[1] for x in v)/len(v):.1f}')
