#!/usr/bin/env python3
"""One-off stress: tests/test_gpu_parity.py::test_random_scenes_vs_oracle over many more seeds (needs a GPU).
    python tests/stress_random_scenes.py [first_seed] [count]
A failure whose message is a tolerance overshoot by a few percent is usually the reference's own fp32 arithmetic
(tests/debug_seed.py <seed> shows the oracle-in-float32 errors beside the HIP ones); anything else is a bug."""
import importlib
import sys

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import tests.test_gpu_parity as T

gs = importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd")
first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 80
bad = 0
for seed in range(first, first + count):
    try:
        T.test_random_scenes_vs_oracle(gs, seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED:", type(e).__name__, str(e)[:300], flush=True)
    if (seed - first) % 50 == 49:
        print("...", seed - first + 1, "done", flush=True)
print("done, failures:", bad, "of", count)
