set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03_a_bench.json 2> gpurun_out/r03_a_bench.err
python tools/host_queue_time.py 3 200 profile > gpurun_out/r03_a_hostq.log 2>&1
KT_ROWS=16 bash tools/kernel_times.sh base "" > gpurun_out/r03_a_kt_base.log 2>&1
KT_ROWS=16 KT_ARGS=--deterministic bash tools/kernel_times.sh det "" > gpurun_out/r03_a_kt_det.log 2>&1
python tools/big_gaussians.py > gpurun_out/r03_a_big.log 2>&1
