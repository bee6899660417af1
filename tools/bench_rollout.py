"""Roll-out only (BASELINE.json configs[3]: Tennis-main 256x256, S=4, batch 1): python tools/bench_rollout.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda", 0)
for rep in range(3):
    t0 = time.time()
    r = bench.rollout_fps(dev, frames)
    print(f"run {rep}: {r['value']:.1f} frames/s  ({r['ms_per_frame'] * 1e3:.0f} us/frame; setup+run {time.time() - t0:.2f} s)", flush=True)
