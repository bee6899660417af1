"""TEST ONLY: what `tools/launch_dp.sh N` starts on an N-GPU node, on the host functional simulator -- the same launcher line (python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P <script> --gpus N --steps K --warmup W), the same bench.run() control flow (hooks, bucketed gradient all-reduce,
profiled steps on every rank, barriers, max-over-ranks timing, one JSON line from rank 0), but gloo instead of RCCL, the simulator build instead of libcaddy_hip.so and a tiny
workload.  Selected with CADDY_DP_SCRIPT=tests/dp_sim_bench.py (tests/test_cabi_and_dp.py); never part of the product."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    a = ap.parse_args()
    import torch.distributed as dist
    from tests.emu.loader import load_emu
    from playablevideogeneration_amd import configs
    import bench
    torch.set_num_threads(1)
    dist.init_process_group("gloo")      # RANK / WORLD_SIZE / MASTER_* from the launcher, as on the GPU node
    configs.WORKLOADS["tiny"] = dict(configs.BREAKOUT, batch=1, seq_len=3, height=32, width=32, gt_init=2, tau=0.8)
    ns = argparse.Namespace(gpus=a.gpus, steps=a.steps, warmup=a.warmup, workload="tiny", no_cpu_baseline=True, profile_steps=1, no_rollout=True)
    res = bench.run(ns, torch.device("cpu"), lib=load_emu(), backend="gloo")
    if int(os.environ.get("RANK", "0")) == 0:
        out = os.environ.get("CADDY_DP_SIM_OUT")
        if out:
            with open(out, "w") as f:
                json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
