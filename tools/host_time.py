"""Is the training step host-bound?  Host enqueue time of one E/R/A/D step (no device wait inside) vs its device time:  python tools/host_time.py [workload]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from playablevideogeneration_amd import configs
from playablevideogeneration_amd.engine import Engine
from playablevideogeneration_amd.init import init_parameters

name = sys.argv[1] if len(sys.argv) > 1 else "bair256_t16_b8"
wl = configs.WORKLOADS[name]
print(name, flush=True)
B, T, H, W, S, K, Da = wl["batch"], wl["seq_len"], wl["height"], wl["width"], wl["stacking"], wl["actions"], wl["action_dim"]
dev = torch.device("cuda")
eng = Engine(variant=wl["variant"], batch=B, seq_len=T, height=H, width=W, stacking=S, actions=K, action_dim=Da, hidden=wl["hidden"], device=dev)
init_parameters(eng, 0)
gen = torch.Generator(device=dev).manual_seed(1)
obs = torch.rand(B, T, 3 * S, H, W, device=dev, generator=gen) * 2 - 1
noise = bench.make_noise(B, T, K, Da, dev, gen)
w = dict(configs.LOSS_WEIGHTS, perceptual=0.0)
def step():
    eng.forward_full(obs, wl["gt_init"], wl["tau"], noise, training=True, fetch_outputs=False)
    eng.loss_backward(w, deferred=True)
for _ in range(3):
    step()
torch.cuda.synchronize()
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.forward_full(obs, wl["gt_init"], wl["tau"], noise, training=True, fetch_outputs=False)
    t1 = time.perf_counter()
    eng.loss_backward(w, deferred=True)
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"host: forward enqueue {1e3 * (t1 - t0):.1f} ms, backward enqueue {1e3 * (t2 - t1):.1f} ms, device done after {1e3 * (t3 - t0):.1f} ms", flush=True)
