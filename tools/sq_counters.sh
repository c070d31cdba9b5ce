#!/bin/bash
# SQ counters of every kernel of a bench step (separate rocprofv3 --pmc passes, counters only):  tools/sq_counters.sh <tag>
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
cmd="python3 $root/bench.py --no-cpu-baseline --no-extras --steps 6 --warmup 2 $SQ_ARGS"
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $root/gpurun_out/${tag}_sq1 -o run -- $cmd > $root/gpurun_out/${tag}_sq1.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS --output-format csv -d $root/gpurun_out/${tag}_sq2 -o run -- $cmd > $root/gpurun_out/${tag}_sq2.log 2>&1
rc=$?
cd $root
{ echo "# rocprofv3 --pmc (two passes) on \`$cmd\`, per-launch averages, counters summed over the chip"; 
  python3 tools/pmc_table.py gpurun_out/${tag}_sq1; python3 tools/pmc_table.py gpurun_out/${tag}_sq2; } > gpurun_out/${tag}_sq_counters.txt
rm -rf gpurun_out/${tag}_sq1 gpurun_out/${tag}_sq2
exit $rc
