#!/bin/bash
# kernel times of one variant:  kt.sh <tag> <lib or ""> 
tag=$1; lib=$2
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
GSPLAT_MI355X_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/kt_$tag -o run -- python3 $root/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 $KT_ARGS > $root/gpurun_out/kt_$tag.log 2>&1
cd $root
python3 tools/profile_summary.py stats gpurun_out/kt_$tag gpurun_out/kt_$tag.csv > /dev/null
echo "== $tag"; head -${KT_ROWS:-8} gpurun_out/kt_$tag.csv | cut -d, -f1,2,4
rm -rf gpurun_out/kt_$tag
