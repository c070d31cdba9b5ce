#!/usr/bin/env python3
"""Kernel timeline of one step from a rocprofv3 kernel trace:  python tools/timeline.py gpurun_out/<tag>/run_kernel_trace.csv"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
cams = [i for i, r in enumerate(rows) if 'project_kernel' in r['Kernel_Name']]      # the first kernel of every step
a, b = cams[-3], cams[-2]
t0 = int(rows[a]['Start_Timestamp'])
prev = t0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')[:52]
    print(f"{name:52s} start {(s - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f} gap {(s - prev) / 1e3:6.1f}")
    prev = e
print(f"step total {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
