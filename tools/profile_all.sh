#!/bin/bash
# The three rocprofv3 passes behind profiles/: kernel trace + stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE (separate passes,
# /opt/skills/guides/MI355X_MICROARCH.md §HBM), all on the command bench.py itself runs (no extras: one scene per profile).
#   tools/profile_all.sh <tag> [stats|all] [extra bench.py arguments, e.g. --config 5 --steps 4]
tag=${1:-r02}
what=${2:-all}
shift; shift
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cmd="python3 $root/bench.py --no-cpu-baseline --no-extras --steps 10 --warmup 3 $*"
rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/${tag}_stats -o run -- $cmd > $root/gpurun_out/${tag}_stats.log 2>&1
rc=$?
if [ "$what" = "all" ] && [ $rc -eq 0 ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $root/gpurun_out/${tag}_fetch -o run -- $cmd > $root/gpurun_out/${tag}_fetch.log 2>&1 &&
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $root/gpurun_out/${tag}_write -o run -- $cmd > $root/gpurun_out/${tag}_write.log 2>&1
  rc=$?
fi
cd $root
grep '^{' gpurun_out/${tag}_stats.log > gpurun_out/${tag}_bench_under_rocprof.json
python3 tools/profile_summary.py stats gpurun_out/${tag}_stats gpurun_out/${tag}_kernel_stats.csv
if [ "$what" = "all" ]; then python3 tools/profile_summary.py pmc gpurun_out/${tag}_fetch gpurun_out/${tag}_write gpurun_out/${tag}_pmc_traffic.json $tag; fi
exit $rc
