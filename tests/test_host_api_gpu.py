"""The reference's plugin seam on the MI355X: `model(config)` / `trainer(config, model, dataset, logger)` factories resolved exactly as
train.py:38-39,54 does (importlib on the dotted paths of the YAML), `.cuda()`, forward vs the oracle, and a short optimisation run."""
import importlib

import pytest
import torch

from oracle import caddy_oracle as O
from tests.test_host_api_emu import _config, trainer_ensemble_case, trainer_golden_case

pytestmark = pytest.mark.gpu


def _build(cfg):
    model = getattr(importlib.import_module(cfg["model"]["architecture"]), "model")(cfg)          # train.py:38-39
    return model.cuda()


def test_plugin_factories_forward_and_training_progress(tmp_path):
    assert torch.cuda.is_available()
    cfg = _config()
    cfg["logging"] = {"save_root_directory": str(tmp_path)}
    m = _build(cfg)
    d = O.Dims.from_config(dict(cfg, model=dict(cfg["model"], architecture="model.reduced_model.model")))
    P = O.make_params(d, seed=7)
    m.load_state_dict(P)
    obs = torch.rand(2, 4, 3, 32, 32, generator=torch.Generator().manual_seed(1)) * 2 - 1
    m.train()
    torch.manual_seed(5)
    out = m((obs, None, None, None), 2, gumbel_temperature=0.7)
    torch.manual_seed(5)
    with torch.no_grad():
        ref = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True).forward_full(obs, 2, tau=0.7)
    assert torch.equal(out[5].cpu(), ref[5]) and (out[0].cpu() - ref[0]).abs().max() < 2e-4
    assert all(t.is_cuda for t in out if torch.is_tensor(t))
    # optimisation on one fixed batch: the reconstruction loss must go down (trainer mirror: schedules, fused losses, Adam)
    trainer = getattr(importlib.import_module(cfg["training"]["trainer"]), "trainer")(cfg, m, dataset=None, logger=None)   # train.py:54
    trainer.global_step = 20000
    first = last = None
    for i in range(12):
        torch.manual_seed(100 + i)
        loss, info, _ = trainer.compute_losses(m, (obs, None, None, None), 4)
        trainer.optimizer_step(m)
        rec = info["avg_observations_rec_loss"]
        first = rec if first is None else first
        last = rec
    assert last < 0.9 * first, (first, last)
    trainer.save_checkpoint(m)
    m2 = _build(cfg)
    tr2 = getattr(importlib.import_module(cfg["training"]["trainer"]), "trainer")(cfg, m2, dataset=None, logger=None)
    tr2.load_checkpoint(m2)
    assert torch.equal(m2._flat.cpu(), m._flat.cpu())


@pytest.mark.parametrize("name", ["trainer_reduced_s1", "trainer_pre_reduced_s1", "trainer_perc_reduced_s1", "trainer_perc_pre_reduced_s1", "trainer_plain_reduced_s1"])
def test_trainer_mirror_matches_reference_trainer_golden_on_gpu(name):
    """the REAL reference's training step (loss_info, MI estimator, post-Adam parameters), perceptual weight 0 and 1, on the real library"""
    trainer_golden_case(name, _build, with_vgg=True)


def test_trainer_mirror_ensemble_of_action_networks_on_gpu(tmp_path):
    """model.action_network.ensamble_size = 2: three steps of the real reference trainer (members 1, 0, 1) vs the mirror on the real library -- losses, parameters, the untouched
    member of every step, Adam's per-parameter step counts, checkpoint round trip"""
    trainer_ensemble_case(_build, tmp_path)


def test_trainer_mirror_zero_fill_semantics_of_torch_1_4_on_gpu(tmp_path):
    """training.zero_grad_semantics: zero_fill (torch < 2.0's optimizer.zero_grad(); the reference pins pytorch 1.4.0) against the golden of the reference trainer under
    zero_grad(set_to_none=False): the undrawn member and state_to_hidden_state_layer keep being stepped with g = 0 once they have had a gradient (caddy_adam_step_ex)"""
    trainer_ensemble_case(_build, tmp_path, zero_fill=True)


def test_checkpoint_loaded_before_cuda_then_step(tmp_path):
    """train.py loads the checkpoint BEFORE model.cuda() (train.py:61-68): Adam moments and the MI estimator restored on the CPU must be on the
    GPU before the kernels get their raw pointers -- the resumed step equals the uninterrupted one: bit for bit, with the plugin's `training.deterministic` switch
    (caddy_set_deterministic: no arrival-order atomics in the backward)."""
    cfg = _config()
    cfg["training"]["deterministic"] = True
    cfg["logging"] = {"save_root_directory": str(tmp_path)}
    obs = torch.rand(2, 4, 3, 32, 32, generator=torch.Generator().manual_seed(1)) * 2 - 1
    d = O.Dims.from_config(dict(cfg, model=dict(cfg["model"], architecture="model.reduced_model.model")))
    P = O.make_params(d, seed=7)
    mk = lambda m: getattr(importlib.import_module(cfg["training"]["trainer"]), "trainer")(cfg, m, dataset=None, logger=None)

    def step(m, tr, seed):
        torch.manual_seed(seed)
        loss, _, _ = tr.compute_losses(m, (obs, None, None, None), 4)
        tr.optimizer_step(m)
        return loss
    m = _build(cfg); m.load_state_dict(P); m.train()
    tr = mk(m); tr.global_step = 20000
    step(m, tr, 100)
    tr.save_checkpoint(m)
    want = step(m, tr, 101)
    m2 = getattr(importlib.import_module(cfg["model"]["architecture"]), "model")(cfg)      # still on the CPU, as in train.py:38-39
    tr2 = mk(m2); tr2.global_step = 20000
    tr2.load_checkpoint(m2)                                                                  # train.py:61-65
    assert not tr2.mi_ema.is_cuda and not tr2.adam_m.is_cuda
    m2 = m2.cuda(); m2.train()                                                               # train.py:67-68
    got = step(m2, tr2, 101)
    assert tr2.mi_ema.is_cuda and tr2.adam_m.is_cuda
    assert abs(got - want) <= 1e-14 * abs(want), (got, want)      # (the loss scalars are fp64 atomic sums over blocks: the last bit depends on the arrival order -- seen once in ~10 suite runs; the parameters below are what must be bit-identical)
    assert torch.equal(m2._flat, m._flat), ("parameters after the resumed step", (m2._flat - m._flat).abs().max().item())      # (the default, atomic backward: +-2 lr on ~0 gradients)


def test_evaluator_mirror_on_gpu():
    """headless Evaluator (per-position losses, entropies, MI, Hungarian accuracy) on the real kernels, vs the oracle"""
    from tests import test_host_api_emu as TE
    TE.headless_evaluator_case(_build, "cuda")


def test_evaluator_per_position_perceptual_loss_on_gpu():
    """the evaluator's per-position perceptual / observation / state losses from the HIP loss network and loss kernels, vs the oracle"""
    from tests import test_host_api_emu as TE
    TE.headless_evaluator_perceptual_case(_build, "cuda")


def test_headless_drivers_cli_train_play_interpolate(tmp_path):
    """`python -m playablevideogeneration_amd.drivers train|play|interpolate --config ...` (train.py:76-108, play.py:115-207, interpolate.py:102-158) end to end on
    the real library: factories from the YAML, model.cuda(), DataLoader over the on-disk video format, checkpoints, evaluation, roll-out frames as PNGs"""
    import os
    from playablevideogeneration_amd import drivers as D
    from tests.test_drivers_emu import _yaml_config
    path = _yaml_config(tmp_path)
    assert D.main(["train", "--config", path, "--max-steps", "2"]) == 0
    cfg = D.load_configuration(path)
    assert os.path.isfile(os.path.join(cfg["logging"]["save_root_directory"], "latest.pth.tar"))
    out = str(tmp_path / "play_results")
    assert D.main(["play", "--config", path, "--actions", "1,2,3,3", "--out", out, "--sample", "1:0"]) == 0
    assert sorted(os.listdir(os.path.join(out, "0"))) == ["0.png", "1.png", "2.png", "3.png", "4.png", "play_metadata.pkl"]
    assert D.main(["interpolate", "--config", path, "--first", "0", "--second", "1", "--steps", "2", "--frames", "3"]) == 0
    seqs = cfg["logging"]["interpolated_sequences"]
    assert sorted(os.listdir(seqs)) == ["0", "1", "2"] and len(os.listdir(os.path.join(seqs, "0"))) == 4
    assert D.main(["build-dataset", "--config", path]) == 0                                   # build_evaluation_dataset.py:17-77
    out_ds = cfg["logging"]["evaluation_dataset_directory"]
    assert len(os.listdir(out_ds)) > 0 and os.path.isfile(os.path.join(out_ds, "00000", "00000.png")) and os.path.isfile(os.path.join(out_ds, "00000", "actions.pkl"))


def test_train_epoch_deferred_loss_readback_matches_step_by_step(tmp_path):
    """train_epoch reads the loss values of step i through an asynchronous copy AFTER step i + 1 has been enqueued (caddy_loss_cfg.no_sync + pinned buffers): the
    logged values and the parameters must be the ones of the plain compute_losses / optimizer_step sequence -- exactly, in the deterministic mode"""
    cfg = _config()
    cfg["training"]["deterministic"] = True
    cfg["logging"] = {"save_root_directory": str(tmp_path)}
    d = O.Dims.from_config(dict(cfg, model=dict(cfg["model"], architecture="model.reduced_model.model")))
    P = O.make_params(d, seed=7)
    g = torch.Generator().manual_seed(3)
    batches = [(torch.rand(2, 4, 3, 32, 32, generator=g) * 2 - 1, None, None, None) for _ in range(4)]
    mk = lambda m, lg: getattr(importlib.import_module(cfg["training"]["trainer"]), "trainer")(cfg, m, dataset=None, logger=lg)

    class Log:
        def __init__(self): self.lines = []
        def print(self, *a, **k): self.lines.append(" ".join(str(x) for x in a))
    ma = _build(cfg); ma.load_state_dict(P); ma.train()
    la = Log(); ta = mk(ma, la); ta.global_step = 20000
    torch.manual_seed(11)
    assert ta.train_epoch(ma, batches) == 4
    mb = _build(cfg); mb.load_state_dict(P); mb.train()
    tb = mk(mb, None); tb.global_step = 20000
    torch.manual_seed(11)
    want = []
    for b in batches:
        tb.global_step += 1
        loss, info, _ = tb.compute_losses(mb, b, 4)
        tb.optimizer_step(mb)
        want.append((tb.global_step, loss, info["avg_observations_rec_loss"]))
    assert len(la.lines) == 4
    import re
    for i, (line, (step, loss, rec)) in enumerate(zip(la.lines, want)):      # every line carries ITS step's values (printed with 3 decimals)
        assert line.startswith(f"step: {step} "), (line, step)
        got_rec = float(re.search(r"avg_observations_rec_loss:([-0-9.]+)", line).group(1)); got_loss = float(re.search(r" loss:([-0-9.]+) lr:", line).group(1))
        assert abs(got_rec - rec) < 6e-4 and abs(got_loss - loss) < 6e-4, (i, line, loss, rec)
    assert ta.last_loss_info["loss"] == want[-1][1]
    assert torch.equal(ma._flat, mb._flat), ("four steps through train_epoch vs by hand", (ma._flat - mb._flat).abs().max().item())
    # one step through train_epoch == one step by hand (parameters; later steps amplify the run-to-run round-off of the gradients through Adam's normalisation)
    mc = _build(cfg); mc.load_state_dict(P); mc.train()
    tc = mk(mc, None); tc.global_step = 20000
    torch.manual_seed(11)
    assert tc.train_epoch(mc, batches[:1]) == 1
    md = _build(cfg); md.load_state_dict(P); md.train()
    td = mk(md, None); td.global_step = 20001
    torch.manual_seed(11)
    td.compute_losses(md, batches[0], 4); td.optimizer_step(md)
    assert torch.equal(mc._flat, md._flat), (mc._flat - md._flat).abs().max().item()
