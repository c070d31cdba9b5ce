#!/usr/bin/env python3
"""Instruction mix per kernel from a hipcc -save-temps gfx950 .s file:  python tools/isa_stats.py file.s [name-filter]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
# kernels: label line "<mangled>:" ... until ".Lfunc_end"
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if flt not in name or "rocprim" in name:
        continue
    lines = [l.strip() for l in body.split("\n")]
    ins = [l.split()[0] for l in lines if l and not l.startswith((";", ".", "_")) and not l.endswith(":")]
    c = collections.Counter(ins)
    grp = collections.Counter()
    for k, v in c.items():
        key = ("v_pk" if k.startswith("v_pk") else "v_exp/rcp/sqrt" if re.match(r"v_(exp|rcp|sqrt|rsq|log)", k) else
               "valu" if k.startswith("v_") else "salu" if k.startswith("s_") else "lds" if k.startswith("ds_") else
               "vmem" if re.match(r"(global|buffer|flat|scratch)_", k) else "other")
        grp[key] += v
    dpp = sum(1 for l in lines if " dpp" in l or "row_" in l or "quad_perm" in l)
    meta = re.search(r"\.name:\s+%s\b(.*?)\.wavefront_size" % re.escape(name), txt, re.S)
    regs = dict(re.findall(r"\.(vgpr_count|sgpr_count|group_segment_fixed_size|private_segment_fixed_size):\s+(\d+)", meta.group(1))) if meta else {}
    print(f"{name[:70]}\n   total {len(ins)}  {dict(grp)}  dpp {dpp}  regs {regs}")
    print("   top:", c.most_common(14))
