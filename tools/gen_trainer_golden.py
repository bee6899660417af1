"""Golden of the REAL reference training step (build container only): SmoothMITrainer.compute_losses + optimizer.zero_grad / backward /
step on one seeded batch (training/trainer.py:400-550,575-587; training/smooth_mi_trainer.py), once with perceptual_loss_lambda = 0 (the
seeded VGG19 of tools/ref_harness.py still runs and is logged) and once with 1.0.  Stored: the scalar loss_info entries, the schedule values,
per-parameter post-step summaries and the MI estimator state -> tests/golden/trainer_[perc_][pre_]reduced_s1.npz.     python tools/gen_trainer_golden.py"""
import os
import random
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness as rh  # noqa: E402
from oracle import caddy_oracle as O  # noqa: E402
from tests.test_host_api_emu import _config  # noqa: E402   (the mirror's test config: the golden must be produced with the same keys)

PARAM_SEED, OBS_SEED, STEP_SEED, GLOBAL_STEP = 7, 1, 11, 5000


PRE_W = {"reconstruction_loss_lambda_pretraining": 1.0, "perceptual_loss_lambda_pretraining": 0.0, "hidden_states_rec_lambda_pretraining": 1.0,
         "states_rec_lambda_pretraining": 0.2, "entropy_lambda_pretraining": 0.0, "action_directions_kl_lambda_pretraining": 1e-4,
         "action_mutual_information_lambda_pretraining": 0.15, "action_state_distribution_kl_lambda_pretraining": 0.0}


def main():
    only = set(sys.argv[1:])
    for perc in (0.0, 1.0):
        for pre in (False, True):
            if not only or "smooth" in only:
                one(pre, perc)
    if not only or "plain" in only:
        one(False, 0.0, plain=True)          # training.trainer (03_tennis.yaml): plain MutualInformationLoss, no estimator state in the checkpoint
    if not only or "ensemble" in only:
        ensemble()
    if not only or "zerofill" in only:
        ensemble(zero_fill=True)


ENS_RSEEDS = [7, 1, 5]      # Python `random` seed of each step: random.choice(self.action_network) (model.py:152) then draws members 1, 0, 1


ZF_KINDS = ["pre", "full", "full"]      # zero_fill golden: a pretraining pass first, so that state_to_hidden_state_layer has had a gradient before the full-model passes


def ensemble(zero_fill=False):
    """zero_fill: the same three steps under torch < 2.0's optimizer.zero_grad() (the reference pins pytorch 1.4.0, env.yml: gradients are zero-FILLED, not dropped) --
    `optimizer.zero_grad(set_to_none=False)` on this image's torch -- with a PRETRAINING pass as the first step: afterwards every parameter that has had a gradient once (the
    member drawn earlier, state_to_hidden_state_layer) is stepped by Adam at every later step with g = 0 (weight decay, ageing moments, its own step count)
    -> tests/golden/trainer_ens2_reduced_s1_zerofill.npz.  Default:
    model.action_network.ensamble_size = 2: three training steps of the real reference trainer.  Each step draws ONE action network (model.py:152); the other one has
    `.grad is None` after optimizer.zero_grad() and torch.optim.Adam neither updates nor decays it, and Adam's per-parameter step counts diverge (member 1: 2 steps, member 0: 1)."""
    rh.install()
    cfg = _config(res=(8, 8))
    cfg["model"]["architecture"] = "model.reduced_model.model"
    cfg["model"]["action_network"]["use_variations"] = True
    cfg["model"]["action_network"]["ensamble_size"] = 2
    tr = cfg["training"]
    tr["trainer"] = "training.smooth_mi_trainer"
    tr["batching"].update(batch_size=2, num_workers=0)
    tr.update(motion_weights_bias=0.1, use_motion_weights=False, action_mutual_information_entropy_lambda=1.0, action_direction_plotting_freq=10 ** 9, max_steps=10 ** 6)
    tr["loss_weights"]["perceptual_loss_lambda"] = 0.0
    tr["loss_weights"].update(PRE_W)
    cfg["logging"] = {"save_root_directory": "/tmp", "output_images_directory": "/tmp"}
    d = O.Dims.from_config(cfg)
    P = O.make_params(d, seed=PARAM_SEED)
    ref = nn.DataParallel(rh.build_reference_model(cfg, P))
    import importlib
    logger = types.SimpleNamespace(print=lambda *a, **k: None, get_wandb=lambda: types.SimpleNamespace(log=lambda *a, **k: None))
    trainer = importlib.import_module(tr["trainer"]).trainer(cfg, ref, [0] * 8, logger)
    trainer.global_step = GLOBAL_STEP
    obs = torch.rand(2, 4, 3, 64, 64, generator=torch.Generator().manual_seed(OBS_SEED)) * 2 - 1
    acts = torch.zeros(2, 4, dtype=torch.int32)
    batch = types.SimpleNamespace(observations=obs, actions=acts, size=4, to_tuple=lambda cuda=True: (obs, acts, None, None))
    ref.train()
    losses, members = [], []
    for i, rs in enumerate(ENS_RSEEDS):
        torch.manual_seed(STEP_SEED + i)
        random.seed(rs)
        st_ = random.getstate(); members.append(random.choice(range(2))); random.setstate(st_)
        pre = zero_fill and ZF_KINDS[i] == "pre"
        loss, info, _ = (trainer.compute_losses_pretraining if pre else trainer.compute_losses)(ref, batch, 4)
        if zero_fill:
            trainer.optimizer.zero_grad(set_to_none=False)      # torch 1.4.0: `p.grad.detach_(); p.grad.zero_()`
        else:
            trainer.optimizer.zero_grad()
        loss.backward()
        trainer.optimizer.step()
        trainer.lr_scheduler.step()
        losses.append(loss.item())
    data = {"losses": np.array(losses), "members": np.array(members), "rseeds": np.array(ENS_RSEEDS), "global_step": np.array(GLOBAL_STEP), "step_seed": np.array(STEP_SEED)}
    if zero_fill:
        data["kinds"] = np.array(ZF_KINDS)
    names, psum, pabs, first, steps = [], [], [], [], []
    for n, p in ref.module.named_parameters():
        names.append(n); psum.append(p.detach().double().sum().item()); pabs.append(p.detach().double().abs().sum().item())
        first.append(p.detach().flatten()[:4].tolist() + [0.0] * max(0, 4 - p.numel()))
        st = trainer.optimizer.state.get(p)
        steps.append(int(float(st["step"])) if st else 0)
    data["param_names"], data["param_sum"], data["param_abs"], data["param_first4"] = np.array(names), np.array(psum), np.array(pabs), np.array(first, dtype=np.float32)
    data["adam_steps"] = np.array(steps)
    data["mi_ema"] = trainer.mutual_information_loss.matrix_estimator.estimated_matrix.detach().numpy()
    data["lr"] = np.array(trainer._get_current_lr())
    out = os.path.join(ROOT, "tests", "golden", "trainer_ens2_reduced_s1_zerofill.npz" if zero_fill else "trainer_ens2_reduced_s1.npz")
    np.savez_compressed(out, **data)
    print("written", out, "members", members, "losses", losses, "adam steps", sorted(set(steps)), {n: st for n, st in zip(names, steps) if "mean_fc.bias" in n or "state_to_hidden" in n})


def one(pretraining, perc=0.0, plain=False):
    rh.install()
    cfg = _config(res=(8, 8))          # 64 x 64 frames: the (stub) VGG19 of the reference's perceptual loss needs >= 16 pixels at the quarter resolution
    cfg["model"]["architecture"] = "model.reduced_model.model"
    cfg["model"]["action_network"]["use_variations"] = True
    tr = cfg["training"]
    tr["trainer"] = "training.trainer" if plain else "training.smooth_mi_trainer"
    tr["batching"].update(batch_size=2, num_workers=0)
    tr.update(motion_weights_bias=0.1, use_motion_weights=False, action_mutual_information_entropy_lambda=1.0, action_direction_plotting_freq=10 ** 9,
              max_steps=10 ** 6)
    tr["loss_weights"]["perceptual_loss_lambda"] = perc
    tr["loss_weights"].update(PRE_W)
    tr["loss_weights"]["perceptual_loss_lambda_pretraining"] = perc
    cfg["logging"] = {"save_root_directory": "/tmp", "output_images_directory": "/tmp"}
    d = O.Dims.from_config(cfg)
    P = O.make_params(d, seed=PARAM_SEED)
    ref = nn.DataParallel(rh.build_reference_model(cfg, P))            # CPU: pass-through, but provides `.module` as the trainer expects
    import importlib
    logger = types.SimpleNamespace(print=lambda *a, **k: None, get_wandb=lambda: types.SimpleNamespace(log=lambda *a, **k: None))
    trainer = importlib.import_module(tr["trainer"]).trainer(cfg, ref, [0] * 8, logger)          # train.py:54
    trainer.global_step = GLOBAL_STEP
    obs = torch.rand(2, 4, 3, 64, 64, generator=torch.Generator().manual_seed(OBS_SEED)) * 2 - 1
    acts = torch.zeros(2, 4, dtype=torch.int32)
    batch = types.SimpleNamespace(observations=obs, actions=acts, size=4, to_tuple=lambda cuda=True: (obs, acts, None, None))
    ref.train()
    torch.manual_seed(STEP_SEED)
    random.seed(STEP_SEED)
    loss, info, _ = (trainer.compute_losses_pretraining if pretraining else trainer.compute_losses)(ref, batch, 4)
    trainer.optimizer.zero_grad()
    loss.backward()
    trainer.optimizer.step()
    trainer.lr_scheduler.step()
    data = {"loss": np.array(loss.item()), "global_step": np.array(GLOBAL_STEP), "step_seed": np.array(STEP_SEED)}
    for k, v in info.items():
        if isinstance(v, (int, float)):
            data["info:" + k] = np.array(float(v))
    names, psum, pabs, first = [], [], [], []
    for n, p in ref.module.named_parameters():
        names.append(n); psum.append(p.detach().double().sum().item()); pabs.append(p.detach().double().abs().sum().item())
        first.append(p.detach().flatten()[:4].tolist() + [0.0] * max(0, 4 - p.numel()))
    data["param_names"], data["param_sum"], data["param_abs"], data["param_first4"] = np.array(names), np.array(psum), np.array(pabs), np.array(first, dtype=np.float32)
    if not plain:
        data["mi_ema"] = trainer.mutual_information_loss.matrix_estimator.estimated_matrix.detach().numpy()
    data["trainer"] = np.array(tr["trainer"])
    data["lr"] = np.array(trainer._get_current_lr())
    data["perceptual_lambda"] = np.array(perc)
    out = os.path.join(ROOT, "tests", "golden", ("trainer_plain_" if plain else ("trainer_perc_" if perc else "trainer_")) + ("pre_reduced_s1.npz" if pretraining else "reduced_s1.npz"))
    np.savez_compressed(out, **data)
    print("written", out, {k: float(v) for k, v in data.items() if k.startswith("info:")})


if __name__ == "__main__":
    main()
