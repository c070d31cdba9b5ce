import sys, importlib, traceback
sys.path.insert(0, '/root/repo')
import tests.test_gpu_parity as T
gs = importlib.import_module("3d-gaussian-splatting-for-novel-view-synthesis_amd")
bad = 0
for seed in range(100, 180):
    try:
        T.test_random_scenes_vs_oracle(gs, seed)
    except Exception as e:
        bad += 1
        print("seed", seed, "FAILED:", type(e).__name__, str(e)[:300])
print("done, failures:", bad)
