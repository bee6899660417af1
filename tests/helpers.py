"""Shared helpers for the parity tests: golden-case loading and input re-derivation (seeds, not stored data)."""
import ast
import os

import numpy as np
import torch

from oracle import caddy_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
PARAM_SEED, OBS_SEED, NOISE_SEED = 7, 1, 5            # must match tools/gen_golden.py
LOSS_W = dict(O.DEFAULT_LOSS_WEIGHTS, state_kl=1e-5, entropy=0.01)


def load_case(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    c = ast.literal_eval(str(z["case"]))
    return c, z


def dims_of(c):
    return O.Dims(variant=c["variant"], actions=c["K"], action_dim=c["Da"], hidden=c["Ch"], stacking=c["S"],
                  state_res=(c["H"] // 8, c["W"] // 8), hard_gumbel=c.get("hard", False), use_gumbel=c.get("use_gumbel", True),
                  use_variations=c.get("use_variations", True), ensemble=c.get("ens", 1))


def inputs_of(c):
    d = dims_of(c)
    P = O.make_params(d, seed=PARAM_SEED)
    obs = torch.rand(c["B"], c["T"], 3 * c["S"], c["H"], c["W"], generator=torch.Generator().manual_seed(OBS_SEED)) * 2 - 1
    return d, P, obs


def golden_outputs(z):
    out = []
    for i in range(20):
        if f"out{i}" in z:
            out.append(torch.from_numpy(z[f"out{i}"]))
        else:
            out.append([torch.from_numpy(z[f"out{i}_{j}"]) for j in range(3)])
    return out


INTERP = [(1, 2, 0.3), (0, 2, 0.8)]       # must match tools/gen_golden.py


def sampler_inputs(c):
    """ground-truth action tensor and the evaluation samplers of a SAMPLER_CASES fixture (tools/gen_golden.py)"""
    from playablevideogeneration_amd import action_samplers as AS
    acts = (torch.arange(c["B"] * c["T"]).reshape(c["B"], c["T"]) % c["K"]).to(torch.int32)
    sampler = AS.OneHotActionSampler() if c["sampler"] == "onehot" else AS.GroundTruthActionSampler({i: (i + 1) % c["K"] for i in range(c["K"])})
    return acts, sampler, (AS.ZeroActionVariationSampler() if c["zero_var"] else None)
