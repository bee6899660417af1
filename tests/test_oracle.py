"""CPU tests of the oracle: against golden vectors produced by the reference (always), against the imported
reference itself (only where /root/reference exists) and against the known answers of SURVEY.md section 4."""
import os
import random
import sys

import numpy as np
import pytest
import torch

from oracle import caddy_oracle as O
from tests import helpers as H

FULL_CASES = ["full_reduced_s1", "full_main_s1", "full_main_s4_hard", "full_reduced_s1_ens2"]      # ..._ens2: model.action_network.ensamble_size 2, member drawn with random.choice


def _cmp(a, b, tol, what):
    if isinstance(a, (list, tuple)):
        for i, (x, y) in enumerate(zip(a, b)):
            _cmp(x, y, tol, f"{what}[{i}]")
        return
    if a.dtype == torch.int64:
        assert torch.equal(a, b), what
    else:
        assert (a - b).abs().max().item() <= tol, (what, (a - b).abs().max().item())


@pytest.mark.parametrize("name", FULL_CASES)
def test_oracle_full_model_matches_reference_golden(name):
    c, z = H.load_case(name)
    d, P, obs = H.inputs_of(c)
    P = {k: v.clone().requires_grad_(O.is_trainable(k)) for k, v in P.items()}
    orc = O.Oracle(d, P, training=True)
    torch.manual_seed(H.NOISE_SEED)
    random.seed(c.get("rseed", H.NOISE_SEED))      # model.py:152: the ensemble member is drawn from Python's global `random` state
    out = orc.forward_full(obs, c["gt"], tau=c["tau"])
    if c.get("ens", 1) > 1:
        assert orc.member == int(z["member"]) == 1      # the same draw as the reference's, and not the default member
        assert all(P[k].grad is None for k in P if k.startswith("action_network.0."))
    _cmp(out, H.golden_outputs(z), 1e-6, name)
    total, comp, ema = O.full_model_loss(out, obs, H.LOSS_W, mi_ema=torch.full((d.K, d.K), 1.0 / (d.K * d.K)), mi_alpha=0.2)
    assert abs(total.item() - float(z["loss_total"])) < 1e-6
    for k, v in comp.items():
        assert abs(v.item() - float(z["loss_" + k])) < 1e-6, k
    assert np.allclose(ema.numpy(), z["mi_ema"], atol=1e-7)
    total.backward()
    for n, gs, ga, g4 in zip(z["grad_names"], z["grad_sum"], z["grad_abs"], z["grad_first4"]):
        g = P[str(n)].grad
        assert abs(g.double().abs().sum().item() - ga) <= 1e-5 * max(1.0, abs(ga)), n
        assert abs(g.double().sum().item() - gs) <= 1e-5 * max(1.0, abs(ga)), n
        k = min(4, g.numel())
        assert np.allclose(g.flatten()[:k].numpy(), g4[:k], rtol=1e-4, atol=1e-7), n
    for k in z.files:
        if k.startswith("buf:"):
            assert np.allclose(P[k[4:]].detach().numpy(), z[k], atol=1e-6), k
    assert np.allclose(P["centroid_estimator.estimated_centroids"].numpy(), z["centroids"], atol=1e-6)


def test_oracle_pretraining_matches_reference_golden():
    c, z = H.load_case("pre_main_s4")
    d, P, obs = H.inputs_of(c)
    orc = O.Oracle(d, P, training=True)
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        out = orc.forward_pretraining(obs, tau=c["tau"])
    _cmp(out, H.golden_outputs(z), 1e-6, "pre")


@pytest.mark.parametrize("name", ["perc_main_s1", "perc_pre_reduced_s1"])
def test_oracle_perceptual_loss_matches_reference_golden(name):
    """VGG19 perceptual term: the reference's ParallelPerceptualLoss + Trainer.sum_loss_components on the seeded VGG weights
    (tools/gen_golden.py) vs oracle.perceptual_terms -- every level x resolution, the total, d(total)/d(rec_r) and parameter gradients."""
    c, z = H.load_case(name)
    d, P, obs = H.inputs_of(c)
    P = {k: v.clone().requires_grad_(O.is_trainable(k)) for k, v in P.items()}
    V = O.make_vgg_params()
    orc = O.Oracle(d, P, training=True)
    torch.manual_seed(H.NOISE_SEED)
    out = orc.forward_pretraining(obs, tau=c["tau"]) if c["pre"] else orc.forward_full(obs, c["gt"], tau=c["tau"])
    _cmp(out, H.golden_outputs(z), 1e-6, name)
    for m in out[1]:
        m.retain_grad()
    w = dict(H.LOSS_W, perceptual=c["perc"])
    fn = O.pretraining_loss if c["pre"] else O.full_model_loss
    total, comp, _ = fn(out, obs, w, mi_ema=torch.full((d.K, d.K), 1.0 / (d.K * d.K)), mi_alpha=0.2, vgg=V)
    assert abs(total.item() - float(z["loss_total"])) < 1e-6
    for r in range(3):
        assert abs(comp[f"perceptual_loss_r{r}"].item() - float(z[f"perceptual_loss_r{r}"])) < 1e-6
        for l in range(5):
            assert abs(comp[f"perceptual_loss_r{r}_l{l}"].item() - float(z[f"perceptual_loss_r{r}_l{l}"])) < 1e-6, (r, l)
    assert abs(comp["perceptual"].item() - float(z["loss_perceptual"])) < 1e-6
    total.backward()
    for r, m in enumerate(out[1]):
        g = torch.from_numpy(z[f"dout1_{r}"])
        assert (m.grad - g).abs().max().item() <= 1e-6 * max(1.0, g.abs().max().item()) + 1e-9, r
    for n, gs, ga in zip(z["grad_names"], z["grad_sum"], z["grad_abs"]):
        g = P[str(n)].grad
        assert abs(g.double().abs().sum().item() - ga) <= 1e-5 * max(1.0, abs(ga)), n


def test_vgg_slicing_matches_torchvision_layout():
    """13 convolutions up to conv5_1, taps after features[1], [6], [11], [20], [29] (model/layers/vgg.py:25-34)"""
    idx = [l[0] for l in O.VGG_LAYERS if l != "M"]
    assert idx == [0, 2, 5, 7, 10, 12, 14, 16, 19, 21, 23, 25, 28] and len(O.vgg_table()) == 26
    f = O.vgg_features(torch.zeros(1, 3, 32, 32), O.make_vgg_params())
    assert [tuple(t.shape[1:]) for t in f] == [(64, 32, 32), (128, 16, 16), (256, 8, 8), (512, 4, 4), (512, 2, 2)]


@pytest.mark.parametrize("name", ["rollout_main_s4", "rollout_reduced_s1"])
def test_oracle_rollout_matches_reference_golden(name):
    c, z = H.load_case(name)
    d, P, obs = H.inputs_of(c)
    orc = O.Oracle(d, P, training=False)
    o = obs[0, 0]
    with torch.no_grad():
        orc.start_inference()
        for i in range(c["steps"]):
            f, o = orc.generate_next(o, i % c["K"])
            assert np.allclose(f.numpy(), z["frames"][i], atol=1e-6), i
    assert np.allclose(o.numpy(), z["last_obs"], atol=1e-6)
    with torch.no_grad():                       # generate_next_interpolation (model.py:609-655) continues the same sequence
        for j, (a1, a2, al) in enumerate(H.INTERP):
            f, _ = orc.generate_next_interpolation(o, a1 % c["K"], a2 % c["K"], al)
            assert np.allclose(f.numpy(), z["interp_frames"][j], atol=1e-6), j
        torch.manual_seed(H.NOISE_SEED + 1)                 # generate_next(noise=True): variation ~ N(0, 1), drawn before the (unused) dynamics noise
        v = torch.randn((1, d.Da))
        f, _ = orc.generate_next(o, 1, v[0])
        assert np.allclose(f.numpy(), z["noise_frame"], atol=1e-6)


@pytest.mark.parametrize("name", ["eval_main_s1_onehot_zero", "eval_reduced_s1_gt"])
def test_oracle_eval_samplers_match_reference_golden(name):
    """eval-mode forward_full_model driven by the reference's evaluation samplers (evaluation/action_sampler.py)"""
    c, z = H.load_case(name)
    d, P, obs = H.inputs_of(c)
    acts, sampler, vsampler = H.sampler_inputs(c)
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        out = O.Oracle(d, P, training=False).forward_full(obs, c["gt"], tau=c["tau"], action_sampler=sampler, variation_sampler=vsampler, gt_actions=acts)
    _cmp(out, H.golden_outputs(z), 1e-6, name)


def test_loss_known_answers():
    """Known-answer values captured from the reference's loss classes."""
    assert abs(O.kl_gaussian_loss(torch.tensor([[[1.0, 1.0], [1.0, 0.005]]])).item() - 3.151658773422241) < 1e-6
    p1 = torch.tensor([[.7, .2, .1], [.1, .8, .1], [.2, .2, .6], [.6, .3, .1]])
    p2 = torch.tensor([[.6, .3, .1], [.2, .7, .1], [.1, .2, .7], [.5, .4, .1]])
    j = O.joint_matrix(p1, p2)
    assert np.allclose(j.numpy(), [[.1900, .1262, .0588], [.1262, .1950, .0662], [.0588, .0662, .1125]], atol=1e-4)
    assert abs(O.mutual_information_loss(p1, p2)[0].item() - (-0.05785660073161125)) < 1e-6
    assert abs(O.mutual_information_loss(p1, p2, lamb=2.0)[0].item() - (-2.2110631465911865)) < 1e-5
    ema = torch.full((3, 3), 1.0 / 9)
    l1, ema = O.mutual_information_loss(p1, p2, ema=ema, alpha=0.2)
    l2, ema = O.mutual_information_loss(p2, p1, ema=ema, alpha=0.2)
    assert abs(l1.item() - (-0.002445726655423641)) < 1e-7 and abs(l2.item() - (-0.007775790058076382)) < 1e-7
    assert abs(O.entropy_logit_loss(torch.tensor([[1., 2., 3.], [0., 0., 0.]])).item() - 0.965503990650177) < 1e-6
    # KLGeneralGaussianDivergenceLoss self-test of training/losses.py:716-725
    a = torch.tensor([[[0.0, 0.0], [1.0, 1.0]]])
    b = torch.tensor([[[1.0, 1.0], [0.01, 0.01]]])
    v = O.kl_general_gaussian_loss(a, b, eps=0.05).item()
    assert np.isfinite(v)


def test_param_table_counts():
    d = O.Dims(variant="main", actions=7, action_dim=2, hidden=128, stacking=1, state_res=(32, 32))
    n = sum(int(np.prod(s)) for k, s in O.param_table(d) if O.is_trainable(k))
    assert n == 9856353                      # trainable scalars of BAIR-main
    assert len(O.param_table(d)) == 246      # state_dict tensors


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="reference tree only exists in the build container")
def test_oracle_bitwise_vs_imported_reference():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ref_harness as rh
    cfg = rh.make_config(variant="reduced", actions=3, action_dim=1, hidden=64, stacking=2, state_res=(4, 6))
    d = O.Dims.from_config(cfg)
    P = O.make_params(d, seed=11)
    ref = rh.build_reference_model(cfg, P)
    ref.train()
    obs = torch.rand(2, 4, 6, 32, 48) * 2 - 1
    torch.manual_seed(3); random.seed(3)
    rout = ref((obs, torch.zeros(2, 4, dtype=torch.int32), None, None), 2, gumbel_temperature=0.6)
    orc = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True)
    torch.manual_seed(3)
    oout = orc.forward_full(obs, 2, tau=0.6)
    _cmp(oout, list(rout), 0.0, "bitwise")
    # ensemble of three action networks (model.py:28,47,152): over several passes the oracle draws the members the reference draws (same `random` state) and matches it bit for bit
    cfg3 = rh.make_config(variant="reduced", actions=3, action_dim=1, hidden=64, stacking=1, state_res=(4, 4), ensamble_size=3)
    d3 = O.Dims.from_config(cfg3)
    P3 = O.make_params(d3, seed=12)
    ref3 = rh.build_reference_model(cfg3, P3)
    ref3.train()
    orc3 = O.Oracle(d3, {k: v.clone() for k, v in P3.items()}, training=True)
    obs3 = torch.rand(2, 3, 3, 32, 32) * 2 - 1
    drawn = set()
    for it in range(4):
        torch.manual_seed(20 + it); random.seed(40 + it)
        with torch.no_grad():
            r3 = ref3((obs3, torch.zeros(2, 3, dtype=torch.int32), None, None), 1, gumbel_temperature=0.8)
        torch.manual_seed(20 + it); random.seed(40 + it)
        with torch.no_grad():
            o3 = orc3.forward_full(obs3, 1, tau=0.8)
        _cmp(o3, list(r3), 0.0, f"bitwise, ensemble pass {it}")
        drawn.add(orc3.member)
    assert len(drawn) > 1      # (the passes exercised different members)


@pytest.mark.skipif(not os.path.isdir("/root/reference/model"), reason="reference tree only exists in the build container")
def test_reference_motion_weighted_losses_cannot_run():
    """Why `use_motion_weights` raises in the trainer mirror instead of being implemented: the reference's own weighted paths are dead code.
    The mask calculator works (losses.py:591-649), but ObservationsLoss' weighted branch calls TensorFolder.fold without its second argument
    (losses.py:105) and raises TypeError for any input -- there is no reference behaviour to be on par with (default False, configuration.py:66)."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import ref_harness as rh
    rh.install()
    from training.losses import MotionLossWeightMaskCalculator, ObservationsLoss
    g = torch.Generator().manual_seed(0)
    obs, rec = torch.rand(2, 4, 3, 16, 24, generator=g) * 2 - 1, torch.rand(2, 3, 3, 16, 24, generator=g) * 2 - 1
    mask = MotionLossWeightMaskCalculator(0.1).compute_weight_mask(obs, rec)
    assert mask.shape == (2, 4, 1, 16, 24) and torch.all(mask[:, 0] == 1.0)
    with pytest.raises(TypeError):
        ObservationsLoss()(obs, rec, mask)
