#!/usr/bin/env python3
"""Condense rocprofv3 output (copied back under gpurun_out/) into small tracked files under profiles/.

  python tools/profile_summary.py stats  <dir with *_kernel_stats.csv>  profiles/<name>_kernel_stats.csv
  python tools/profile_summary.py pmc    <fetch dir> <write dir>        profiles/pmc_traffic.json [tag]

`pmc` follows /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE and WRITE_SIZE come from SEPARATE --pmc passes, are
in KiB, and on gfx950 FETCH_SIZE counts 128-B requests as 64 B (exactly half for wide coalesced reads), so
hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch.  (16-B gathers are not calibrated by the guide; the x2 was
checked here on project_kernel, whose reads are a known streamed byte count.)
"""
import collections
import csv
import glob
import json
import re
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"(?:void )?([A-Za-z0-9_:]+)", name)
    base = m.group(1) if m else name
    t = re.match(r"(?:void )?[A-Za-z0-9_:]+(<[0-9a-z, ]+>)\(", name)       # keep simple template arguments: list_sort_kernel<512, 16, 13>
    if t:
        base += t.group(1).replace(", ", ";").replace(",", ";")
    if "rocprim" in name:
        k = re.search(r"(radix_sort\w*|onesweep\w*|scan\w*|lookback\w*|histogram\w*|block_sort\w*|merge\w*)", name)
        base = "rocprim::" + (k.group(1) if k else "kernel")
    return base


def stats(src, dst):
    f = glob.glob(src + "/**/*_kernel_stats.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    with open(dst, "w") as o:
        o.write("kernel,calls,total_ms,avg_us,min_us,max_us,pct\n")
        for r in rows:
            o.write(f"{short(r['Name'])},{r['Calls']},{float(r['TotalDurationNs']) / 1e6:.3f},{float(r['AverageNs']) / 1e3:.1f},"
                    f"{float(r['MinNs']) / 1e3:.1f},{float(r['MaxNs']) / 1e3:.1f},{r['Percentage']}\n")
    print("wrote", dst)


def pmc_mean(d, counter):
    f = glob.glob(d + "/**/*_counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}


def pmc(fetch_dir, write_dir, dst, tag=""):
    fe, wr = pmc_mean(fetch_dir, "FETCH_SIZE"), pmc_mean(write_dir, "WRITE_SIZE")
    out = {"_note": "per launch; hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024, separate --pmc passes (MI355X_MICROARCH.md §HBM)",
           "_tag": tag}
    for k in sorted(set(fe) | set(wr)):
        f, w = fe.get(k, 0.0), wr.get(k, 0.0)
        out[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
