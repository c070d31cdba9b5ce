#!/usr/bin/env python3
"""profiles/<tag>_sq_counters.txt (tools/sq_counters.sh) -> profiles/valu_config<N>.json: per kernel, VALU wave-instructions per launch and
the VALU pipeline's busy fraction  SQ_ACTIVE_INST_VALU * 4 / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)  (MI355X_MICROARCH.md: the SQ
counters count quad-cycles; rocprofv3 sums GRBM_GUI_ACTIVE over the 8 XCDs).
    python tools/valu_summary.py profiles/r03_v3_sq_counters.txt profiles/valu_config3.json"""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
out, cur = {"_source": src, "_note": "per launch; busy = SQ_ACTIVE_INST_VALU * 4 / 1024 / (GRBM_GUI_ACTIVE / 8)"}, None
for line in open(src):
    m = re.match(r"^(\S.*?): launches (\d+) avg_us ([0-9.]+)", line)
    if m:
        cur = out.setdefault(m.group(1), {"avg_us": float(m.group(3))})
        continue
    m = re.match(r"^\s+(\w+)\s+([0-9.]+)", line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
for k, v in list(out.items()):
    if not isinstance(v, dict):
        continue
    if "SQ_INSTS_VALU" in v and v.get("GRBM_GUI_ACTIVE"):
        v["insts_valu"] = v["SQ_INSTS_VALU"]
        v["busy"] = v["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / (v["GRBM_GUI_ACTIVE"] / 8)
        v["clock_ghz"] = v["GRBM_GUI_ACTIVE"] / 8 / (v["avg_us"] * 1e3)
    else:
        del out[k]
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: {kk: v[kk] for kk in ("avg_us", "insts_valu", "busy", "clock_ghz")} for k, v in out.items() if isinstance(v, dict)}, indent=1))
