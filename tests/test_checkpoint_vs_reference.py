"""Checkpoint interchange with the REAL reference trainer (build container only: needs /root/reference; skipped elsewhere).
A checkpoint written by training/smooth_mi_trainer.py resumes in the trainer mirror and vice versa: after loading, one more training step
on the same seeded batch must give the same weights on both sides (model state, Adam moments + step count, LR schedule, MI estimator)."""
import os
import random
import sys
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import caddy_oracle as O
from tests.test_host_api_emu import PRE_W, _config, _make_model

pytestmark = [pytest.mark.emu, pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="reference tree only exists in the build container")]


def _cfg(tmp):
    cfg = _config(res=(8, 8))
    tr = cfg["training"]
    tr["batching"].update(batch_size=2, num_workers=0)
    tr.update(motion_weights_bias=0.1, use_motion_weights=False, action_mutual_information_entropy_lambda=1.0, action_direction_plotting_freq=10 ** 9, max_steps=10 ** 6)
    tr["loss_weights"]["perceptual_loss_lambda"] = 0.0
    tr["loss_weights"].update(PRE_W)
    cfg["logging"] = {"save_root_directory": str(tmp), "output_images_directory": str(tmp)}
    return cfg


def _reference_side(cfg, P):
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ref_harness as rh
    rh.install()
    rcfg = dict(cfg, model=dict(cfg["model"], architecture="model.reduced_model.model"))
    rcfg["model"]["action_network"] = dict(cfg["model"]["action_network"], use_variations=True)
    ref = nn.DataParallel(rh.build_reference_model(rcfg, P))
    import training.smooth_mi_trainer as SM
    logger = types.SimpleNamespace(print=lambda *a, **k: None, get_wandb=lambda: types.SimpleNamespace(log=lambda *a, **k: None))
    tr = SM.SmoothMITrainer(rcfg, ref, [0] * 8, logger)
    ref.train()
    return ref, tr


def _ref_step(ref, tr, obs, seed):
    acts = torch.zeros(obs.shape[0], obs.shape[1], dtype=torch.int32)
    batch = types.SimpleNamespace(observations=obs, actions=acts, size=obs.shape[1], to_tuple=lambda cuda=True: (obs, acts, None, None))
    torch.manual_seed(seed); random.seed(seed)
    tr.global_step += 1
    loss, info, _ = tr.compute_losses(ref, batch, obs.shape[1])
    tr.optimizer.zero_grad(); loss.backward(); tr.optimizer.step(); tr.lr_scheduler.step()
    return loss.item()


def _mirror_step(m, tr, obs, seed):
    torch.manual_seed(seed)
    tr.global_step += 1
    loss, _, _ = tr.compute_losses(m, (obs, torch.zeros(obs.shape[0], obs.shape[1], dtype=torch.int32), None, None), obs.shape[1])
    tr.optimizer_step(m)
    return loss


def _close(m, ref, lr):
    """weights equal up to a few elements per tensor whose ~0 gradient moved the other way (Adam: 2 * lr per such element)"""
    sd = dict(m.named_parameters())
    for n, p in ref.module.named_parameters():
        a, b = sd[n].detach().double(), p.detach().double()
        assert (a - b).abs().max().item() <= 2.5 * lr + 1e-6, (n, (a - b).abs().max().item())
        # elements whose gradient is pure round-off (|g| ~ 1e-12) take +-lr steps at random in BOTH implementations: allow 3 % of a tensor
        assert ((a - b).abs() > 1e-5).sum().item() <= max(4, 0.03 * a.numel()), (n, ((a - b).abs() > 1e-5).sum().item(), a.numel())


@pytest.mark.parametrize("direction", ["reference_to_mirror", "mirror_to_reference"])
def test_checkpoint_interchange(tmp_path, direction):
    from playablevideogeneration_amd import smooth_mi_trainer
    cfg = _cfg(tmp_path)
    d = O.Dims.from_config(dict(cfg, model=dict(cfg["model"], architecture="model.reduced_model.model")))
    P = O.make_params(d, seed=7)
    obs = torch.rand(2, 3, 3, 64, 64, generator=torch.Generator().manual_seed(1)) * 2 - 1      # small clip; 64x64 is the smallest frame the reference trainer's VGG19 accepts
    ref, rtr = _reference_side(cfg, P)
    m = _make_model(cfg)
    m.load_state_dict(P)
    m.train()
    mtr = smooth_mi_trainer.trainer(cfg, m, dataset=None, logger=None)
    rtr.global_step = mtr.global_step = 5000
    if direction == "reference_to_mirror":
        _ref_step(ref, rtr, obs, 20)
        rtr.save_checkpoint(ref)                                   # <root>/latest.pth.tar, written by the REAL reference
        mtr.load_checkpoint(m)
        assert mtr.global_step == rtr.global_step and mtr.opt_steps == 1
        assert np.allclose(mtr.mi_ema.numpy(), rtr.mutual_information_loss.matrix_estimator.estimated_matrix.detach().numpy(), atol=1e-7)
    else:
        _mirror_step(m, mtr, obs, 20)
        mtr.save_checkpoint(m)
        rtr.load_checkpoint(ref.module)                            # the REAL reference loads the mirror's file (model, optimizer, lr_scheduler, mi_estimator, step); unwrapped module: its state_dict keys carry no "module." prefix
        assert rtr.global_step == mtr.global_step
    la, lb = _ref_step(ref, rtr, obs, 31), _mirror_step(m, mtr, obs, 31)
    assert abs(la - lb) < 5e-5 * max(1.0, abs(la)), (la, lb)
    _close(m, ref, 4e-4)
