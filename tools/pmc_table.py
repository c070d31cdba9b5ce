#!/usr/bin/env python3
"""Per-kernel table of rocprofv3 --pmc counters: python tools/pmc_table.py <dir> [name-filter]"""
import collections
import csv
import glob
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from profile_summary import short

f = glob.glob(sys.argv[1] + "/**/*_counter_collection.csv", recursive=True)[0]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    k = short(r["Kernel_Name"])
    if flt and flt not in k:
        continue
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[(k, r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for k, cs in agg.items():
    d = [v for (kk, _), v in dur.items() if kk == k]
    print(f"{k}: launches {len(d)} avg_us {sum(d) / len(d):.1f}")
    for c, v in cs.items():
        print(f"    {c:28s} {sum(v) / len(v):16.1f}")
