#!/bin/bash
# Kernel-time profile of a command on the GPU box:  tools/prof.sh <tag> <python script and args...>
# Writes gpurun_out/<tag>/ (rocprofv3 csv) and prints the per-kernel averages.
tag=$1; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -o run -- python3 $root/"$@" > $root/gpurun_out/$tag.log 2>&1
cd $root
python3 - <<EOF
import csv, glob
f = glob.glob("gpurun_out/$tag/**/*kernel_stats.csv", recursive=True)
if not f:
    print(open("gpurun_out/$tag.log").read()[-2000:]); raise SystemExit(1)
for r in list(csv.DictReader(open(f[0])))[:22]:
    print(f'{r["Name"][:90]:90s} {r["Calls"]:>5s} {float(r["AverageNs"]) / 1e3:9.1f} us {r["Percentage"]:>6s}%')
EOF
