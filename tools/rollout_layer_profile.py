"""Per-launch conv timing of ONE roll-out frame (Tennis-main 256x256, S=4, batch 1; eager launches, HIP events around every conv):
    CADDY_ROLLOUT_GRAPH=0 python tools/rollout_layer_profile.py"""
import os, sys
os.environ.setdefault("CADDY_ROLLOUT_GRAPH", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from playablevideogeneration_amd import configs
from playablevideogeneration_amd.engine import Engine
from playablevideogeneration_amd.init import init_parameters

c = dict(configs.TENNIS)
dev = torch.device("cuda")
eng = Engine(variant=c["variant"], batch=1, seq_len=2, height=256, width=256, stacking=c["stacking"], actions=c["actions"], action_dim=c["action_dim"], hidden=c["hidden"], device=dev)
init_parameters(eng, seed=0)
obs = torch.rand(3 * c["stacking"], 256, 256, device=dev) * 2 - 1
eng.start_inference()
for i in range(3):
    _, obs = eng.generate_next(obs, i % c["actions"])
eng.profile_begin()
_, obs = eng.generate_next(obs, 1)
recs = eng.profile_records(); fam = eng.profile_end()
tot = 0.0
for kind, P, Kc, Cout, KS, fl, ms in recs:
    tot += ms
    print(f"P={int(P):7d} K={int(Kc):5d} Cout={int(Cout):5d} k{int(KS)}  {ms * 1e3:7.1f} us  {fl / ms / 1e9 if ms else 0:6.1f} TF")
print(f"{len(recs)} conv launches, {tot * 1e3:.0f} us of conv time in the frame")
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(64):
    _, obs = eng.generate_next(obs, i % c["actions"])
torch.cuda.synchronize(); print(f"eager: {(time.perf_counter() - t0) / 64 * 1e6:.0f} us / frame")
