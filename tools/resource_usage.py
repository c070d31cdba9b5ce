#!/usr/bin/env python3
"""Registers, scratch, occupancy and LDS of every kernel:  make -C 3d-..._amd/csrc resource-usage"""
import re
import sys

cur, d = None, {}
for l in sys.stdin:
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur, d = m.group(1), {}
    for key, pat in (("vgpr", r" VGPRs: (\d+)"), ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                     ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
        m = re.search(pat, l)
        if m:
            d[key] = int(m.group(1))
    if "LDS Size" in l and cur:
        name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", cur)[:48]
        print(f"{name:50s} vgpr {d.get('vgpr', -1):4d}  scratch {d.get('scratch', -1):4d}  waves/SIMD {d.get('occ', -1)}  lds {d.get('lds', -1)}")
