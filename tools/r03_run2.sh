set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py tests/test_gpu_training.py -x -q -m gpu -k "composite or deferred or capacity or offscreen or sh_tensors or trainer or data_parallel or fused_vs_reference or factored" > gpurun_out/r03_b_tests.log 2>&1 || { tail -40 gpurun_out/r03_b_tests.log; exit 1; }
tail -3 gpurun_out/r03_b_tests.log
python tools/host_queue_time.py 3 100 profile > gpurun_out/r03_b_hostq.log 2>&1 || { tail -30 gpurun_out/r03_b_hostq.log; exit 1; }
head -4 gpurun_out/r03_b_hostq.log
python bench.py --no-cpu-baseline --steps 100 --warmup 10 > gpurun_out/r03_b_bench.json 2> gpurun_out/r03_b_bench.err || { tail -30 gpurun_out/r03_b_bench.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/r03_b_bench.json'))
print(d['value'], d['ms_per_step'], d.get('host_queue_ms'), d['sustained']['ms_per_step'], d['train_step'], d['forward_only']['frame_by_frame_ms'], d['forward_only']['render_frames_ms'])"
