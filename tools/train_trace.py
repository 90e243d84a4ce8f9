#!/usr/bin/env python3
"""Timeline of one training step from a rocprofv3 kernel trace:  python tools/train_trace.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_qsample' in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b]:
    s = int(r['Start_Timestamp']) - t0; e = int(r['End_Timestamp']) - t0
    print(f"{s/1e3:9.1f} {e/1e3:9.1f} {(e-s)/1e3:7.1f} q{r['Queue_Id']} {r['Kernel_Name'][:70]}")
