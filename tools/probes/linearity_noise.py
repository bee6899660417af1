"""Run-to-run noise of the backward at the BASELINE geometry: repeatability (same weights twice) and linearity (weights x2) of the flat gradient,
several times, so that the bound in tests/model_cases.py:property_case can be set from a distribution and not from one sample."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from playablevideogeneration_amd import _lib
from tests import model_cases as M, helpers as H
from playablevideogeneration_amd.init import init_parameters

c = dict(variant="main", K=7, Da=2, Ch=128, S=1, B=8, T=16, H=256, W=256, gt=6, tau=1.0)
lib = _lib.load()
eng = M.make_engine(c, lib, "cuda")
init_parameters(eng, 3)
g = torch.Generator(device="cuda").manual_seed(3)
B, T, K, Da = c["B"], c["T"], c["K"], c["Da"]
obs = torch.rand(B, T, 3, 256, 256, device="cuda", generator=g) * 2 - 1
n = T - 1
noise = {"eps_states": torch.randn(B * T, Da, device="cuda", generator=g), "eps_dirs": torch.randn(B * n, Da, device="cuda", generator=g),
         "gumbel_uniform": torch.rand(B * n, K, device="cuda", generator=g),
         "eps_states_rec": torch.randn(B * T, Da, device="cuda", generator=g), "eps_dirs_rec": torch.randn(B * n, Da, device="cuda", generator=g)}
eng.forward_full(obs, c["gt"], c["tau"], noise, training=True)
w1 = dict(H.LOSS_W)
w2 = {k: 2 * v for k, v in w1.items() if k != "mi_entropy"}
rel = lambda a, b: ((a - b).double().norm() / b.double().norm()).item()
for it in range(4):
    eng.loss_backward(w1, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False); g1 = eng.grads.clone()
    eng.loss_backward(w1, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False); g1b = eng.grads.clone()
    eng.loss_backward(w2, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False); g2 = eng.grads.clone()
    print(f"repeat {rel(g1b, g1):.2e}  linear {rel(g2, 2 * g1):.2e}", flush=True)
