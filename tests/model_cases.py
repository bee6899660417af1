"""Whole-path parity cases (HIP engine vs oracle and vs reference golden vectors), shared by simulator and GPU tests."""
import ctypes as C
import os

import numpy as np
import torch

from oracle import caddy_oracle as O
from playablevideogeneration_amd.engine import Engine
from tests import helpers as H


def noise_dict(rec, B, T, K, Da, gumbel=True):
    """Map the oracle's recorded RNG draws (reference order) to the engine's noise arguments.
    gumbel=False: an explicit action sampler replaced the Gumbel draw (model.py:171-176), so the record is one entry shorter."""
    n = T - 1
    g = 1 if gumbel else 0
    return {"eps_states": rec[0], "eps_dirs": rec[1].reshape(B * n, Da), "gumbel_uniform": rec[2] if gumbel else torch.full((B * n, K), 0.5),
            "eps_states_rec": rec[2 + g + n], "eps_dirs_rec": rec[3 + g + n].reshape(B * n, Da)}


# full_case's bounds on the bit-reproducible backward (the default): worst per-parameter max error relative to that parameter's largest fp64 gradient, and the relative
# deviation of sum |grad| from the reference's own summaries (z["grad_abs"]).  Written for the arrival-order backward as 1e-1 / 5e-2 (rounds 1 - 4); round 6: set to what the
# deterministic backward shows on the MI355X (two boxes, bit-identical values) with 2 x head-room:
#   worst per-parameter error   reduced_s1 1.37e-2, main_s1 9.8e-4, main_s4_hard 1.21e-2, plainmi 1.25e-2, nogumbel 5.6e-3 | novar 3.42e-2, ens2 3.96e-2
#   worst sum |grad| deviation  2.2e-4, 1.3e-4, 9.98e-3, 2.3e-4, 2.7e-3 | novar 3.45e-2, ens2 5.7e-3
# The three goldens right of the bar keep a looser pair: novar carries one documented LeakyReLU slope decision (test_gradient_offset_of_the_2e2_floor_goldens_is_a_slope_decision),
# nogumbel had one in rounds 3 - 5 and can get it back with any change of summation order, ens2's undrawn member leaves the action head's tiny gradients dominated by round-off.
# The host simulator (other kernel variants and summation orders of the same goldens) keeps the old pair, see full_case.
WORST_PARAM_FLOOR = 3e-2
GRAD_ABS_TOL = 2e-2
LOOSE_CASES = {"full_reduced_s1_novar": (7e-2, 7e-2), "full_main_s1_nogumbel": (7e-2, 7e-2), "full_reduced_s1_ens2": (8e-2, 2e-2)}      # (per-parameter floor, sum |grad| tolerance)
SIM_SPLIT = False      # simulator runs: exact-fp32 convolutions by default (the split-operand kernels are 4x slower to simulate; they have their own cases)


def make_engine(c, lib, dev, perceptual=False):
    eng = Engine(variant=c["variant"], batch=c["B"], seq_len=c["T"], height=c["H"], width=c["W"], stacking=c["S"], actions=c["K"],
                 action_dim=c["Da"], hidden=c["Ch"], hard_gumbel=c.get("hard", False), use_gumbel=c.get("use_gumbel", True),
                 use_variations=c.get("use_variations", True), device=dev, lib=lib, perceptual=perceptual, ensemble=c.get("ens", 1))
    if dev == "cpu":
        eng.set_precision(*((16, 17) if SIM_SPLIT else (0, 0)))
        eng.set_vgg_precision(*((16, 17) if SIM_SPLIT else (0, 0)))
    # every test runs with the first-touch part of the gradient arena NaN-filled before each backward: a gradient that is read before
    # its single writer assigned it cannot go unnoticed (fresh workspaces are often zero pages, which would mask it)
    lib.caddy_debug_set_poison.argtypes = [C.c_void_p, C.c_int]
    lib.caddy_debug_set_poison(eng.ctx, 1)
    return eng


def _cmp(a, b, tol, what):
    if isinstance(a, (list, tuple)):
        for i, (x, y) in enumerate(zip(a, b)):
            _cmp(x, y, tol, f"{what}[{i}]")
        return
    a, b = a.detach().cpu(), b.detach().cpu()
    if a.dtype == torch.int64:
        assert torch.equal(a, b), what          # action indices: bit-exact
    else:
        err = (a - b).abs().max().item()
        assert err <= tol * max(1.0, b.abs().max().item()), (what, err)


def _oracle_run(c, d, P, obs, dtype, record):
    """Oracle forward + loss + backward in `dtype` replaying the recorded fp32 noise.  c["mi"] == "plain": MutualInformationLoss (training.trainer)
    instead of the smooth variant -- no estimator state."""
    Po = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in P.items()}
    for k in Po:
        if O.is_trainable(k):
            Po[k].requires_grad_(True)
    orc = O.Oracle(d, Po, training=True, member=c.get("_member"))      # (ensemble cases: the member full_case drew)
    out = orc.forward_full(obs.to(dtype), c["gt"], tau=c["tau"], noise=O.Noise(replay=[t.to(dtype) for t in record]))
    ema0 = None if c.get("mi") == "plain" else torch.full((d.K, d.K), 1.0 / (d.K * d.K), dtype=dtype)
    total, comp, ema = O.full_model_loss(out, obs.to(dtype), H.LOSS_W, mi_ema=ema0, mi_alpha=0.2)
    total.backward()
    return Po, out


def full_case(name, lib, dev, fwd_tol=2e-4, prep=None, deterministic=True, grad_floor=5e-3):
    """forward_full_model + losses + backward + BN buffers + centroids + MI-EMA.

    Forward / losses / buffers: against the REFERENCE's golden vectors (and the oracle).  Gradients are compared with an
    fp64 run of the oracle.  Two effects bound what "equal" can mean in fp32: (1) train-mode BatchNorm over tiny batches x
    BPTT amplifies round-off (the fp32 oracle itself is up to ~1e-2 off the fp64 one on the reduced case); (2) LeakyReLU's
    slope is discontinuous at 0, so pre-activations within the ~1e-5 forward round-off of zero take the other slope
    (0.2 <-> 1) and shift the BatchNorm-backward means -- O(1e-3) relative, data dependent (verified element by element
    with the caddy_debug_* introspection API; single-step graphs, where no flip occurs, agree to 2e-5).  Criterion:
    relative L2 error <= max(2 x fp32-oracle error, 5e-3) and per-parameter max error <= max(5 x, 3e-2; three goldens looser: LOOSE_CASES) on the library's default, bit-reproducible backward (round 5; no
    arrival-order atomics, so no run-to-run noise on top of the arithmetic's own error); a missing or wrong term in the backward graph shows up as O(0.1 - 1).
    deterministic=False (caddy_set_deterministic(0): fp32 atomics in arrival order): relative L2 error <= max(5 x fp32-oracle error, 3e-2).
    """
    c, z = H.load_case(name)
    d, P, obs = H.inputs_of(c)
    orc = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True)
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    import random
    random.seed(c.get("rseed", H.NOISE_SEED))      # model.py:152: the ensemble member comes from Python's global `random` state
    with torch.no_grad():
        oout = orc.forward_full(obs, c["gt"], tau=c["tau"], noise=nz)
    eng = make_engine(c, lib, dev)
    eng.load_state_dict(P)
    if c.get("ens", 1) > 1:
        assert orc.member == int(z["member"]) == 1, (orc.member, int(z["member"]))      # the oracle draws what the reference drew (and not the default member)
        eng.set_action_member(orc.member)
        c = dict(c, _member=orc.member)
    if prep is not None:
        prep(eng)
    smooth = c.get("mi") != "plain"
    out = eng.forward_full(obs, c["gt"], c["tau"], noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"], gumbel=c.get("use_gumbel", True)), training=True)
    _cmp(out, list(oout), fwd_tol, name + " vs oracle")
    _cmp(out, H.golden_outputs(z), fwd_tol, name + " vs reference golden")
    if not c.get("use_gumbel", True):
        assert torch.equal(out[7].cpu(), torch.softmax(out[6].cpu(), -1)) or (out[7].cpu() - torch.softmax(out[6].cpu(), -1)).abs().max() < 1e-6      # model.py:177-179
    if not c.get("use_variations", True):
        assert out[14].abs().max().item() == 0.0                                                                                                  # model.py:188-189
    # frame MSE criterion of the north star (evaluation/metrics/mse.py:21): mean over C,H,W per (b,t), within 1e-5
    mse = ((out[0].cpu() - torch.from_numpy(z["out0"])) ** 2).mean(dim=(2, 3, 4))
    assert mse.max().item() < 1e-5
    eng.set_deterministic(bool(deterministic))
    losses = eng.loss_backward(H.LOSS_W, smooth_mi=smooth, mi_alpha=0.2)
    assert abs(losses["total"] - float(z["loss_total"])) < 2e-5, (losses["total"], float(z["loss_total"]))
    for k in ("rec", "states", "entropy", "dir_kl", "mi", "state_kl"):
        tol = 1e-3 if k.endswith("_kl") else 1e-4      # the KL terms contain log(variance) of a 2-sample BatchNorm'ed head: ill-conditioned (the trainer goldens use 2e-3)
        assert abs(losses[k] - float(z["loss_" + k])) < tol * max(1.0, abs(float(z["loss_" + k]))), (k, losses[k], float(z["loss_" + k]))
    if smooth:
        assert np.allclose(eng.mi_ema.cpu().numpy(), z["mi_ema"], atol=1e-6)
    else:
        assert eng.mi_ema is None and "mi_ema" not in z.files              # the plain loss keeps no estimator (losses.py:238-302)
    P64, _ = _oracle_run(c, d, P, obs, torch.float64, nz.record)
    P32, _ = _oracle_run(c, d, P, obs, torch.float32, nz.record)
    num_h = num_o = den = 0.0
    worst_h = worst_o = 0.0
    for n, _ in O.param_table(d):
        if not O.is_trainable(n):
            continue
        g64 = P64[n].grad if P64[n].grad is not None else torch.zeros_like(P64[n])
        g32 = (P32[n].grad if P32[n].grad is not None else torch.zeros_like(P32[n])).double()
        g = eng.grad_view(n).cpu().double()
        s = max(g64.abs().max().item(), 1e-9)
        worst_h, worst_o = max(worst_h, (g - g64).abs().max().item() / s), max(worst_o, (g32 - g64).abs().max().item() / s)
        if os.environ.get("CADDY_TEST_VERBOSE") and (g - g64).abs().max().item() / s > 0.02:
            print(f"  grad {n}: err {(g - g64).abs().max().item() / s:.4f} (oracle32 {(g32 - g64).abs().max().item() / s:.4f}), scale {s:.3e}")
        num_h += ((g - g64) ** 2).sum().item(); num_o += ((g32 - g64) ** 2).sum().item(); den += (g64 ** 2).sum().item()
    rel_h, rel_o = (num_h / den) ** 0.5, (num_o / den) ** 0.5
    assert rel_h <= (max(2 * rel_o, grad_floor) if deterministic else max(5 * rel_o, 3e-2)), ("relative L2 gradient error vs fp64", rel_h, rel_o)
    # (the tight pair is what the MI355X run of the default kernels shows; the host simulator runs other kernel variants / summation orders of the same goldens -- e.g. the
    #  fused-BatchNorm paths on the split-operand tiles move one LeakyReLU decision of full_reduced_s1 and its action head's sum |grad| by 2.5 % -- and keeps rounds 1 - 5's pair)
    param_floor, abs_tol = LOOSE_CASES.get(name, (WORST_PARAM_FLOOR, GRAD_ABS_TOL)) if (deterministic and dev != "cpu") else (1e-1, 5e-2)
    assert worst_h <= max(5 * worst_o, param_floor), ("worst per-parameter gradient error vs fp64", worst_h, worst_o)
    # reference's own gradient summaries (loose: same conditioning caveat)
    worst_abs = 0.0
    for n, ga in zip(z["grad_names"], z["grad_abs"]):
        got = eng.grad_view(str(n)).double().abs().sum().item()
        worst_abs = max(worst_abs, abs(got - ga) / max(ga, 1e-4))
        assert abs(got - ga) <= abs_tol * max(ga, 1e-4), (n, got, ga)
    sd = eng.state_dict()
    for k in z.files:
        if k.startswith("buf:"):
            assert np.allclose(sd[k[4:]].cpu().numpy(), z[k], atol=1e-4), k
    assert np.allclose(sd["centroid_estimator.estimated_centroids"].cpu().numpy(), z["centroids"], atol=1e-5)
    return eng, dict(rel_l2_hip=rel_h, rel_l2_oracle32=rel_o, worst_hip=worst_h, worst_oracle32=worst_o, worst_grad_abs=worst_abs)


def slope_flip_record(name, lib, dev, max_flips=4, candidates=24, done=5e-3):
    """Element-level record of a gradient offset caused by LeakyReLU slope decisions.  LeakyReLU'(x) jumps from 0.2 to 1 at x = 0; a pre-activation that
    the fp64 oracle puts within the forward round-off (~1e-6 of the tensor's scale) of zero can land on the other side in an fp32 forward, and through the train-mode BatchNorms of
    a tiny batch that one slope shifts every upstream gradient by O(1e-2).  Procedure: (1) HIP gradients g of the golden case; (2) fp64 oracle with every LeakyReLU input
    recorded; the `candidates` elements closest to zero relative to their tensor's rms; (3) greedy search: re-run the fp64 oracle with one candidate at a time forced onto the
    OTHER slope (the forward changes by < 1e-6), keep the flip that brings the oracle's gradients closest to g, repeat up to max_flips times or until the relative L2 distance is
    below `done`.  Returns the flips (call index of oracle._lrelu, tensor shape, flat index, fp64 value, tensor rms) and the distance before / after; asserts that the flips
    explain the offset (after <= done and after <= before / 4) -- otherwise the offset would be a kernel error, not a slope decision."""
    import torch.nn.functional as F
    c, z = H.load_case(name)
    d, P, obs = H.inputs_of(c)
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True).forward_full(obs, c["gt"], tau=c["tau"], noise=nz)
    eng = make_engine(c, lib, dev)
    eng.load_state_dict(P)
    eng.forward_full(obs, c["gt"], c["tau"], noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"], gumbel=c.get("use_gumbel", True)), training=True, fetch_outputs=False)
    eng.loss_backward(H.LOSS_W, smooth_mi=c.get("mi") != "plain", mi_alpha=0.2)
    names = [n for n, _ in O.param_table(d) if O.is_trainable(n)]
    g_hip = {n: eng.grad_view(n).cpu().double() for n in names}
    orig = O._lrelu
    seen = []

    def run(flips, record=False):
        calls = [0]

        def lrelu(x):
            k = calls[0]; calls[0] += 1
            if record:
                seen.append(x.detach())
            y = F.leaky_relu(x, O.LRELU)
            idx = [f for kk, f in flips if kk == k]
            if idx:
                m = torch.zeros(x.numel(), dtype=torch.bool); m[idx] = True
                y = torch.where(m.view_as(x), torch.where(x > 0, O.LRELU * x, x), y)
            return y
        O._lrelu = lrelu
        try:
            P64, _ = _oracle_run(c, d, P, obs, torch.float64, nz.record)
        finally:
            O._lrelu = orig
        num = den = 0.0
        for n in names:
            g64 = P64[n].grad if P64[n].grad is not None else torch.zeros_like(P64[n])
            num += ((g_hip[n] - g64) ** 2).sum().item(); den += (g64 ** 2).sum().item()
        return (num / den) ** 0.5
    before = run([], record=True)
    cand = []
    for k, x in enumerate(seen):
        rms = x.pow(2).mean().sqrt().item() + 1e-30
        r = (x.abs().flatten() / rms)
        v, i = torch.topk(r, min(8, r.numel()), largest=False)
        cand += [(v[j].item(), k, int(i[j]), x.flatten()[int(i[j])].item(), rms, tuple(x.shape)) for j in range(len(v))]
    cand.sort()
    cand = cand[:candidates]
    flips, cur, chosen = [], before, []
    for _ in range(max_flips):
        if cur <= done:
            break
        best = None
        for rel, k, i, val, rms, shape in cand:
            if (k, i) in flips:
                continue
            e = run(flips + [(k, i)])
            if best is None or e < best[0]:
                best = (e, k, i, val, rms, shape, rel)
        if best is None or best[0] >= cur:
            break
        cur = best[0]
        flips.append((best[1], best[2]))
        chosen.append(dict(lrelu_call=best[1], shape=best[5], flat_index=best[2], value_fp64=best[3], tensor_rms=best[4], value_over_rms=best[6], rel_l2_after=best[0]))
    rec = dict(rel_l2_before=before, rel_l2_after=cur, flips=chosen)
    assert cur <= done and (not chosen or cur <= before / 4), rec      # (no flip needed when this build's forward round-off happens to fall on the oracle's side everywhere)
    return rec


def perceptual_case(name, lib, dev, fwd_tol=2e-4, grad_tol=5e-3, prep=None):
    """VGG19 perceptual loss end to end (training/losses.py:379-491, trainer.py:442-466,494-500) against the golden written by the reference's own
    ParallelPerceptualLoss on the seeded VGG weights: every level x resolution, avg / term / total, d(total)/d(rec_r) for the three
    resolutions (= the L1 seed + the VGG dgrad chain) and the parameter-gradient summaries."""
    c, z = H.load_case(name)
    d, P, obs = H.inputs_of(c)
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    orc = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True)
    with torch.no_grad():
        oout = orc.forward_pretraining(obs, tau=c["tau"], noise=nz) if c["pre"] else orc.forward_full(obs, c["gt"], tau=c["tau"], noise=nz)
    eng = make_engine(c, lib, dev, perceptual=True)
    eng.load_state_dict(P)
    eng.load_vgg(O.make_vgg_params())
    if prep is not None:
        prep(eng)
    nd = noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"])
    out = eng.forward_pretraining(obs, c["tau"], nd, training=True) if c["pre"] else eng.forward_full(obs, c["gt"], c["tau"], nd, training=True)
    _cmp(out, list(oout), fwd_tol, name + " vs oracle")
    _cmp(out, H.golden_outputs(z), fwd_tol, name + " vs reference golden")
    w = dict(H.LOSS_W, perceptual=c["perc"])
    losses = eng.loss_backward(w, smooth_mi=True, mi_alpha=0.2)
    for r in range(3):
        ref = float(z[f"perceptual_loss_r{r}"])
        assert abs(losses[f"perceptual_loss_r{r}"] - ref) < 1e-4 * ref, (r, losses[f"perceptual_loss_r{r}"], ref)
        for l in range(5):
            ref = float(z[f"perceptual_loss_r{r}_l{l}"])
            assert abs(losses[f"perceptual_loss_r{r}_l{l}"] - ref) < 1e-4 * ref, (r, l, losses[f"perceptual_loss_r{r}_l{l}"], ref)
    assert abs(losses["perceptual"] - float(z["loss_perceptual"])) < 1e-4 * float(z["loss_perceptual"])
    assert abs(losses["perceptual_term"] - float(z["loss_perceptual_term"])) < 1e-4 * float(z["loss_perceptual_term"])
    assert abs(losses["total"] - float(z["loss_total"])) < 1e-4 * abs(float(z["loss_total"])), (losses["total"], float(z["loss_total"]))
    # parameter gradients after the full backward: the reference's own summaries (sum |g| per tensor).  Same conditioning caveat as full_case.
    worst_abs = 0.0
    for n, ga in zip(z["grad_names"], z["grad_abs"]):
        got = eng.grad_view(str(n)).double().abs().sum().item()
        worst_abs = max(worst_abs, abs(got - ga) / max(ga, 1e-4))
        assert abs(got - ga) <= 5e-2 * max(ga, 1e-4), (n, got, ga)
    # Gradients w.r.t. the reconstructions.  The perceptual gradient is DISCONTINUOUS in its input (sign() of the feature L1, ReLU masks,
    # max-pool arg-max): fp32 round-off flips a handful of those decisions, each flip moving one image's gradient by O(1e-2) while all
    # other images agree to 1e-6.  Criteria: (a) per image, against the ORACLE evaluated at the engine's own reconstructions (identical
    # inputs): median relative error <= max(2e-3, s) and worst image <= max(1e-1, 4 s), s = the oracle's own per-image response to a 1e-5
    # perturbation of the reconstruction (larger frames: more decisions per image, every image has a few flips); (b) against the reference golden (inputs differ by the forward
    # tolerance): relative L2 <= 3 x the reference arithmetic's own response to a 1e-5 input perturbation (measured with the oracle), >= 2e-2.
    V = O.make_vgg_params()

    def oracle_seed(m0):
        m = m0.clone().requires_grad_(True)
        tot, comps = O.perceptual_loss(obs, m, V)
        term = comps[0] * 0.0
        for cc in comps:
            term = term + cc * c["perc"]
        (O.observations_loss(obs, m) * (H.LOSS_W["rec"] / 3) + term / 3).backward()
        return m.grad
    info = dict(per_image_median=[], per_image_worst=[], vs_golden=[], reference_sensitivity_1e5=[])
    if c["pre"]:      # the reference's `.grad` of the folded pretraining outputs includes what E sends back through the re-stacked frames: compare the full backward
        for r in range(3):
            ref = torch.from_numpy(z[f"dout1_{r}"])
            g = eng.output_grad(100 + r, ref.to(dev)).cpu()
            rel = ((g - ref).double().norm() / ref.double().norm()).item()
            info["vs_golden"].append(rel)
            assert rel < 3e-2, ("d(total)/d(rec_r)", r, rel)
    lib.caddy_debug_set_seeds_only.argtypes = [C.c_void_p, C.c_int]
    lib.caddy_debug_set_seeds_only(eng.ctx, 1)      # direct loss terms only (what `.grad` of the reference's stacked full-model outputs holds)
    try:
        eng.loss_backward(w, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
        gen = torch.Generator().manual_seed(3)
        for r in range(3):
            mine = out[1][r].cpu()
            g = eng.output_grad(100 + r, out[1][r]).cpu().flatten(0, 1)
            go = oracle_seed(mine).flatten(0, 1)
            per = torch.tensor([((g[i] - go[i]).double().norm() / go[i].double().norm()).item() for i in range(g.shape[0])])
            info["per_image_median"].append(per.median().item()); info["per_image_worst"].append(per.max().item())
            g1p = oracle_seed(mine + 1e-5 * torch.randn(mine.shape, generator=gen)).flatten(0, 1)
            sens_img = torch.tensor([((g1p[i] - go[i]).double().norm() / go[i].double().norm()).item() for i in range(g.shape[0])]).median().item()
            assert per.median().item() < max(2e-3, sens_img) and per.max().item() < max(1e-1, 4 * sens_img), ("seed d(total)/d(rec_r) per image", r, per.tolist(), sens_img)
            if not c["pre"]:
                ref = torch.from_numpy(z[f"dout1_{r}"]).flatten(0, 1)
                o_r = oout[1][r]
                g0 = oracle_seed(o_r).flatten(0, 1)
                # (that the oracle reproduces the reference's gradient is pinned in tests/test_oracle.py; on another host CPU its own forward
                # differs by round-off, which this discontinuous gradient amplifies -- hence no tight assertion here)
                g1 = oracle_seed(o_r + 1e-5 * torch.randn(o_r.shape, generator=gen)).flatten(0, 1)
                sens = ((g1 - g0).double().norm() / g0.double().norm()).item()
                rel = ((g - ref).double().norm() / ref.double().norm()).item()
                info["vs_golden"].append(rel); info["reference_sensitivity_1e5"].append(sens)
                assert rel < max(3 * sens, 2e-2), ("seed d(total)/d(rec_r) vs golden", r, rel, sens)
    finally:
        lib.caddy_debug_set_seeds_only(eng.ctx, 0)
    return eng, info


def perceptual_oracle_case(lib, dev, c, lam=1.0):
    """perceptual term vs the oracle on an arbitrary geometry (no golden): losses per level, total, seed gradients per image"""
    d, P, obs = H.inputs_of(c)
    V = O.make_vgg_params()
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    orc = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True)
    with torch.no_grad():
        oout = orc.forward_full(obs, c["gt"], tau=c["tau"], noise=nz)
    eng = make_engine(c, lib, dev, perceptual=True)
    eng.load_state_dict(P)
    eng.load_vgg(V)
    out = eng.forward_full(obs, c["gt"], c["tau"], noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"]), training=True)
    _cmp(out, list(oout), 2e-4, "vs oracle")
    w = dict(H.LOSS_W, perceptual=lam)
    total, comp, _ = O.full_model_loss(oout, obs, w, mi_ema=torch.full((d.K, d.K), 1.0 / (d.K * d.K)), mi_alpha=0.2, vgg=V)
    lib.caddy_debug_set_seeds_only.argtypes = [C.c_void_p, C.c_int]
    lib.caddy_debug_set_seeds_only(eng.ctx, 1)
    try:
        losses = eng.loss_backward(w, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
        assert abs(losses["total"] - total.item()) < 1e-4 * abs(total.item()), (losses["total"], total.item())
        for r in range(3):
            for k in [f"perceptual_loss_r{r}"] + [f"perceptual_loss_r{r}_l{l}" for l in range(5)]:
                assert abs(losses[k] - comp[k].item()) < 1e-4 * comp[k].item(), (k, losses[k], comp[k].item())
            m = out[1][r].cpu().clone().requires_grad_(True)
            tot, comps = O.perceptual_loss(obs, m, V)
            term = comps[0] * 0.0
            for cc in comps:
                term = term + cc * lam
            (O.observations_loss(obs, m) * (w["rec"] / 3) + term / 3).backward()
            g, go = eng.output_grad(100 + r, out[1][r]).cpu().flatten(0, 1), m.grad.flatten(0, 1)
            per = torch.tensor([((g[i] - go[i]).double().norm() / go[i].double().norm()).item() for i in range(g.shape[0])])
            # yardstick: the reference arithmetic's own response to a 1e-5 perturbation of the reconstruction (discontinuous gradient, see perceptual_case)
            m2 = (out[1][r].cpu() + 1e-5 * torch.randn(out[1][r].shape, generator=torch.Generator().manual_seed(r))).requires_grad_(True)
            tot2, comps2 = O.perceptual_loss(obs, m2, V)
            term2 = comps2[0] * 0.0
            for cc in comps2:
                term2 = term2 + cc * lam
            (O.observations_loss(obs, m2) * (w["rec"] / 3) + term2 / 3).backward()
            g2 = m2.grad.flatten(0, 1)
            sens = torch.tensor([((g2[i] - go[i]).double().norm() / go[i].double().norm()).item() for i in range(g.shape[0])]).median().item()
            assert per.median().item() < max(2e-3, sens) and per.max().item() < max(1e-1, 4 * sens), (r, per.tolist(), sens)
    finally:
        lib.caddy_debug_set_seeds_only(eng.ctx, 0)
    return eng


def single_step_grad_case(lib, dev, variant="main", tol_median=5e-3, tol_worst=5e-2, size=32):
    """T=2 (one R/D step + D->E feedback + both A calls): gradient check of every op's backward vs the fp64 oracle.
    Without a LeakyReLU slope flip every parameter agrees to ~2e-5; one flip in D shifts everything upstream by ~1e-3
    (see full_case), hence median / worst bounds instead of a uniform tight one.  size = 128 reaches the kernels only larger feature maps
    select (k_conv_narrow, k_conv_c4<7,*>, the tile-resident / split-bf16 weight gradients, 16x16-pixel conv_hx tiles)."""
    K_, Da, Ch = (7, 2, 128) if variant == "main" else (3, 1, 64)
    c = dict(variant=variant, K=K_, Da=Da, Ch=Ch, S=1, B=2, T=2, H=size, W=size, gt=1, tau=0.7, hard=False)
    d, P, obs = H.inputs_of(c)
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True).forward_full(obs, 1, tau=0.7, noise=nz)
    P64, _ = _oracle_run(c, d, P, obs, torch.float64, nz.record)
    eng = make_engine(c, lib, dev)
    eng.load_state_dict(P)
    eng.forward_full(obs, 1, 0.7, noise_dict(nz.record, 2, 2, K_, Da), training=True)
    eng.loss_backward(H.LOSS_W)
    errs = []
    for n, _ in O.param_table(d):
        if not O.is_trainable(n) or P64[n].grad is None:
            continue
        g64 = P64[n].grad
        s = g64.abs().max().item()
        if s < 1e-7:
            continue
        errs.append(((eng.grad_view(n).cpu().double() - g64).abs().max().item() / s, n))
    errs.sort(reverse=True)
    assert errs[0][0] < tol_worst, errs[:3]
    assert errs[len(errs) // 2][0] < tol_median, errs[len(errs) // 2]


def rollout_oracle_case(lib, dev, c, steps):
    """play.py path at its real size against the CPU oracle: start_inference + `steps` x generate_next (actions i mod K, zero variation),
    every frame (max error, evaluation/metrics/mse.py:21 frame MSE) and the final stacked observation.  This is the code the roll-out
    benchmark times: batch-1 launches (split-K slabs, thin kernels with compact tables, eval-mode BatchNorm) replayed from a captured graph."""
    d, P, obs = H.inputs_of(dict(c, B=1, T=1))
    orc = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=False)
    eng = make_engine(dict(c, B=1, T=2), lib, dev)
    eng.load_state_dict(P)
    o_ref = o = obs[0, 0]
    worst = worst_mse = 0.0
    with torch.no_grad():
        orc.start_inference()
        eng.start_inference()
        for i in range(steps):
            fr, o_ref = orc.generate_next(o_ref, i % c["K"])
            f, o = eng.generate_next(o, i % c["K"])
            err = f.cpu() - fr
            worst, worst_mse = max(worst, err.abs().max().item()), max(worst_mse, (err ** 2).mean().item())
            assert err.abs().max().item() < 2e-3 and (err ** 2).mean().item() < 1e-6, (i, err.abs().max().item(), (err ** 2).mean().item())
    assert (o.cpu() - o_ref).abs().max().item() < 2e-3
    # a second roll-out from the same start must reproduce the first one bit for bit (graph replay, persistent ConvLSTM state re-initialised)
    eng.start_inference()
    o2 = obs[0, 0]
    for i in range(steps):
        f2, o2 = eng.generate_next(o2, i % c["K"])
    assert torch.equal(o2, o) and torch.equal(f2, f)
    return dict(worst_abs=worst, worst_mse=worst_mse)


def oracle_case(lib, dev, c, fwd_tol=3e-4):
    """Forward + losses vs the fp32 oracle on an ad-hoc geometry (no golden): outputs, exact action indices, frame MSE."""
    d, P, obs = H.inputs_of(c)
    orc = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True)
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        oout = orc.forward_full(obs, c["gt"], tau=c["tau"], noise=nz)
        total, comp, _ = O.full_model_loss(oout, obs, H.LOSS_W, mi_ema=torch.full((d.K, d.K), 1.0 / (d.K * d.K)), mi_alpha=0.2)
    eng = make_engine(c, lib, dev)
    eng.load_state_dict(P)
    out = eng.forward_full(obs, c["gt"], c["tau"], noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"]), training=True)
    _cmp(out, list(oout), fwd_tol, "vs oracle")
    assert ((out[0].cpu() - oout[0]) ** 2).mean(dim=(2, 3, 4)).max().item() < 1e-5
    losses = eng.loss_backward(H.LOSS_W)
    assert abs(losses["total"] - total.item()) < 1e-4 * max(1.0, abs(total.item()))
    assert torch.isfinite(eng.grads).all()


def oracle_grad_case(lib, dev, c, fwd_tol=3e-4, grad_floor=5e-3, plain_mi=False, factor=2.0):
    """An ad-hoc geometry end to end WITHOUT a golden: all 20 outputs (action indices bit-exact), frame MSE, every loss term, and the gradients against an fp64 run of the oracle
    with full_case's criterion -- relative L2 error <= max(factor x the fp32 oracle's own error, grad_floor), worst per-parameter error <= max(5 x the oracle's, 1e-1).  plain_mi: the
    MutualInformationLoss of `training.trainer` (configs/03_tennis.yaml:61) instead of the smooth estimator."""
    if plain_mi:
        c = dict(c, mi="plain")
    d, P, obs = H.inputs_of(c)
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        oout = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True).forward_full(obs, c["gt"], tau=c["tau"], noise=nz)
        ema0 = None if plain_mi else torch.full((d.K, d.K), 1.0 / (d.K * d.K))
        total, comp, _ = O.full_model_loss(oout, obs, H.LOSS_W, mi_ema=ema0, mi_alpha=0.2)
    eng = make_engine(c, lib, dev)
    eng.load_state_dict(P)
    out = eng.forward_full(obs, c["gt"], c["tau"], noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"]), training=True)
    _cmp(out, list(oout), fwd_tol, "vs oracle")
    mse = ((out[0].cpu() - oout[0]) ** 2).mean(dim=(2, 3, 4)).max().item()
    assert mse < 1e-5, mse
    losses = eng.loss_backward(H.LOSS_W, smooth_mi=not plain_mi, mi_alpha=0.2)
    assert abs(losses["total"] - total.item()) < 1e-4 * max(1.0, abs(total.item())), (losses["total"], total.item())
    for k in ("rec", "states", "entropy", "dir_kl", "mi", "state_kl"):
        tol = 1e-3 if k.endswith("_kl") else 1e-4
        assert abs(losses[k] - comp[k].item()) < tol * max(1.0, abs(comp[k].item())), (k, losses[k], comp[k].item())
    P64, _ = _oracle_run(c, d, P, obs, torch.float64, nz.record)
    P32, _ = _oracle_run(c, d, P, obs, torch.float32, nz.record)
    num_h = num_o = den = worst_h = worst_o = 0.0
    for n, _ in O.param_table(d):
        if not O.is_trainable(n):
            continue
        g64 = P64[n].grad if P64[n].grad is not None else torch.zeros_like(P64[n])
        g32 = (P32[n].grad if P32[n].grad is not None else torch.zeros_like(P32[n])).double()
        g = eng.grad_view(n).cpu().double()
        s = max(g64.abs().max().item(), 1e-9)
        worst_h, worst_o = max(worst_h, (g - g64).abs().max().item() / s), max(worst_o, (g32 - g64).abs().max().item() / s)
        num_h += ((g - g64) ** 2).sum().item(); num_o += ((g32 - g64) ** 2).sum().item(); den += (g64 ** 2).sum().item()
    rel_h, rel_o = (num_h / den) ** 0.5, (num_o / den) ** 0.5
    assert rel_h <= max(factor * rel_o, grad_floor), ("relative L2 gradient error vs fp64", rel_h, rel_o)
    assert worst_h <= max(5 * worst_o, 1e-1), ("worst per-parameter gradient error vs fp64", worst_h, worst_o)
    return eng, dict(frame_mse=mse, rel_l2_hip=rel_h, rel_l2_oracle32=rel_o, worst_hip=worst_h, worst_oracle32=worst_o)


def pretraining_case(lib, dev, name="pre_main_s4", fwd_tol=2e-4):
    """forward_pretraining vs the reference's golden vector (S=4 re-stacking included) and the oracle; loss terms incl. the
    hidden-states loss; gradients vs the fp64 oracle (same robust criterion as full_case)."""
    c, z = H.load_case(name)
    d, P, obs = H.inputs_of(c)
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        oout = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True).forward_pretraining(obs, tau=c["tau"], noise=nz)
    eng = make_engine(c, lib, dev)
    eng.load_state_dict(P)
    out = eng.forward_pretraining(obs, c["tau"], noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"]), training=True)
    _cmp(out, list(oout), fwd_tol, name + " vs oracle")
    _cmp(out, H.golden_outputs(z), fwd_tol, name + " vs reference golden")
    w = dict(H.LOSS_W, hidden=1.0)
    losses = eng.loss_backward(w, smooth_mi=True, mi_alpha=0.2)

    def orun(dtype):
        Po = {k: (v.clone().to(dtype) if v.dtype.is_floating_point else v.clone()) for k, v in P.items()}
        for k in Po:
            if O.is_trainable(k):
                Po[k].requires_grad_(True)
        o = O.Oracle(d, Po, training=True).forward_pretraining(obs.to(dtype), tau=c["tau"], noise=O.Noise(replay=[t.to(dtype) for t in nz.record]))
        total, comp, _ = O.pretraining_loss(o, obs.to(dtype), w, mi_ema=torch.full((d.K, d.K), 1.0 / (d.K * d.K), dtype=dtype), mi_alpha=0.2)
        total.backward()
        return Po, total, comp
    P64, t64, c64 = orun(torch.float64)
    P32, _, _ = orun(torch.float32)
    assert abs(losses["total"] - t64.item()) < 1e-4 * max(1.0, abs(t64.item())), (losses["total"], t64.item())
    assert abs(losses["hidden"] - c64["hidden"].item()) < 1e-4 * max(1.0, abs(c64["hidden"].item()))
    num_h = num_o = den = 0.0
    for n, _ in O.param_table(d):
        if not O.is_trainable(n):
            continue
        g64 = P64[n].grad if P64[n].grad is not None else torch.zeros_like(P64[n])
        g32 = (P32[n].grad if P32[n].grad is not None else torch.zeros_like(P32[n])).double()
        g = eng.grad_view(n).cpu().double()
        num_h += ((g - g64) ** 2).sum().item(); num_o += ((g32 - g64) ** 2).sum().item(); den += (g64 ** 2).sum().item()
    rel_h, rel_o = (num_h / den) ** 0.5, (num_o / den) ** 0.5
    assert rel_h <= max(2 * rel_o, 5e-3), ("relative L2 gradient error vs fp64", rel_h, rel_o)      # (the default backward is the bit-reproducible one: round 5)


def rollout_case(name, lib, dev, tol=2e-4, fold=True):
    """start_inference + generate_next vs the reference golden frames.  fold: eval-mode BatchNorms folded into the convolutions (the default
    roll-out graph) or launched separately."""
    c, z = H.load_case(name)
    d, P, obs = H.inputs_of(c)
    cc = dict(c, B=1, T=2)
    eng = make_engine(cc, lib, dev)
    eng.load_state_dict(P)
    eng.set_rollout_fold(fold)
    o = obs[0, 0]
    eng.start_inference()
    for i in range(c["steps"]):
        f, o = eng.generate_next(o, i % c["K"])
        err = (f.cpu().numpy() - z["frames"][i])
        assert np.abs(err).max() < tol and (err ** 2).mean() < 1e-5, (i, np.abs(err).max())
    assert np.abs(o.cpu().numpy() - z["last_obs"]).max() < tol
    cen = P["centroid_estimator.estimated_centroids"]
    for j, (a1, a2, al) in enumerate(H.INTERP):      # generate_next_interpolation: nearer centroid + offset as the variation
        a1, a2 = a1 % c["K"], a2 % c["K"]
        sel = a2 if al > 0.5 else a1
        f, _ = eng.generate_next(o, sel, (cen[a2] - cen[a1]) * al + cen[a1] - cen[sel])
        assert np.abs(f.cpu().numpy() - z["interp_frames"][j]).max() < tol, ("interpolation", j)


def sampler_case(name, lib, dev, tol=2e-4):
    """eval-mode forward_full_model with the evaluation samplers called mid-forward through caddy_set_sampler_hook, vs the
    reference golden (reference samplers) and the oracle."""
    c, z = H.load_case(name)
    d, P, obs = H.inputs_of(c)
    acts, sampler, vsampler = H.sampler_inputs(c)
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        oout = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=False).forward_full(
            obs, c["gt"], tau=c["tau"], noise=nz, action_sampler=sampler, variation_sampler=vsampler, gt_actions=acts)
    eng = make_engine(c, lib, dev)
    eng.load_state_dict(P)
    eng.set_samplers(sampler, vsampler, acts[:, :-1].reshape(-1).to(dev))
    nd = noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"], gumbel=False)
    out = eng.forward_full(obs, c["gt"], c["tau"], nd, training=False)
    _cmp(out, list(oout), tol, name + " vs oracle")
    _cmp(out, H.golden_outputs(z), tol, name + " vs reference golden")
    eng.set_samplers(None, None)
    out2 = eng.forward_full(obs, c["gt"], c["tau"], nd, training=False)
    assert not torch.equal(out2[7].cpu(), out[7].cpu())        # cleared: Gumbel samples again


def property_case(lib, dev, c, seed=3, perceptual=False):
    """Size-independent properties of the whole path, for geometries where the oracle is too slow (the BASELINE workload:
    BAIR 256x256, T=16, B=8; perceptual=True: with the VGG19 term inside the step, i.e. exactly the configuration bench.py times):  (1) the forward pass is bit-reproducible; (2) in eval mode clips are independent, so permuting
    the batch permutes every output bit-exactly (action indices included); (3) the reported total is the weighted sum of the
    reported components; (4) the backward pass is linear in the loss weights (all weights x2 -> all gradients x2);
    (5) frames are tanh-bounded, probabilities normalised, gradients finite and non-trivial."""
    from playablevideogeneration_amd.init import init_parameters, random_vgg19_state
    B, T, K, Da, S, Hh, W = c["B"], c["T"], c["K"], c["Da"], c["S"], c["H"], c["W"]
    eng = make_engine(c, lib, dev, perceptual=perceptual)
    init_parameters(eng, seed)
    if perceptual:
        eng.load_vgg(random_vgg19_state(0))
    g = torch.Generator(device=dev).manual_seed(seed)
    obs = torch.rand(B, T, 3 * S, Hh, W, device=dev, generator=g) * 2 - 1
    n = T - 1
    noise = {"eps_states": torch.randn(B * T, Da, device=dev, generator=g), "eps_dirs": torch.randn(B * n, Da, device=dev, generator=g),
             "gumbel_uniform": torch.rand(B * n, K, device=dev, generator=g),
             "eps_states_rec": torch.randn(B * T, Da, device=dev, generator=g), "eps_dirs_rec": torch.randn(B * n, Da, device=dev, generator=g)}
    saved = eng.params.clone()

    def flat(o):
        return [t for x in o for t in (x if isinstance(x, (list, tuple)) else [x])]

    # (1) reproducibility (train mode mutates BN running stats / centroids: restore the parameter buffer in between)
    a = flat(eng.forward_full(obs, c["gt"], c["tau"], noise, training=True))
    eng.params.copy_(saved)
    b = flat(eng.forward_full(obs, c["gt"], c["tau"], noise, training=True))
    for i, (x, y) in enumerate(zip(a, b)):
        assert torch.equal(x, y), ("forward not reproducible", i)
    # (5) ranges
    assert a[0].abs().max().item() <= 1.0 and torch.isfinite(a[0]).all()
    # (3) total = weighted sum; (4) linearity of the backward in the weights
    w1 = dict(H.LOSS_W, perceptual=1.0) if perceptual else dict(H.LOSS_W)
    l1 = eng.loss_backward(w1, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
    g1 = eng.grads.clone()
    tot = sum(w1[k] * l1[k] for k in ("rec", "states", "entropy", "dir_kl", "mi", "state_kl"))
    if perceptual:      # trainer.py:494-500: the perceptual term enters the total as lambda * (level-weighted sum) / 3 = `perceptual_term`
        tot += l1["perceptual_term"]
        assert l1["perceptual_term"] > 0 and all(l1[f"perceptual_loss_r{r}"] > 0 for r in range(3))
        l1b = eng.loss_backward(w1, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)      # the backward (incl. the side-stream VGG19 levels) is repeatable to round-off
        relb = ((eng.grads - g1).double().norm() / g1.double().norm()).item()
        # (split-K dgrads accumulate with fp32 atomics in arrival order; BPTT through the closed-loop steps amplifies that run-to-run round-off: measured 1.6e-4)
        assert abs(l1b["total"] - l1["total"]) <= 1e-9 * abs(l1["total"]) and relb < 1e-3, ("backward with the perceptual term not repeatable", relb)
    assert abs(tot - l1["total"]) <= 1e-6 * max(1.0, abs(tot)), (tot, l1["total"])
    assert torch.isfinite(g1).all() and g1.abs().max().item() > 0
    l2 = eng.loss_backward({k: 2 * v for k, v in w1.items() if k != "mi_entropy"}, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
    g2 = eng.grads.clone()
    assert abs(l2["total"] - 2 * l1["total"]) <= 1e-6 * max(1.0, abs(l1["total"]))
    rel = ((g2 - 2 * g1).double().norm() / (2 * g1).double().norm()).item()
    # bound from a distribution, not one sample: tools/probes/linearity_noise.py on the MI355X gives 2e-5 .. 3e-4 for BOTH the repeat of the same backward and the
    # x2 run at BAIR 256x256 T=16 B=8 (fp32 atomics of the split-K dgrads in arrival order, amplified through the 10 closed-loop steps); small geometries: < 1e-6
    assert rel < 1e-3, ("backward not linear in the loss weights", rel)
    # (2) eval-mode batch permutation
    if B > 1:
        eng.params.copy_(saved)
        perm = torch.arange(B - 1, -1, -1, device=dev)

        def pn(t, per):
            return t.reshape(B, per, -1)[perm].reshape(t.shape).contiguous()
        noise_p = {"eps_states": pn(noise["eps_states"], T), "eps_dirs": pn(noise["eps_dirs"], n), "gumbel_uniform": pn(noise["gumbel_uniform"], n),
                   "eps_states_rec": pn(noise["eps_states_rec"], T), "eps_dirs_rec": pn(noise["eps_dirs_rec"], n)}
        e0 = flat(eng.forward_full(obs, c["gt"], c["tau"], noise, training=False))
        e1 = flat(eng.forward_full(obs[perm].contiguous(), c["gt"], c["tau"], noise_p, training=False))
        for i, (x, y) in enumerate(zip(e0, e1)):
            if x.dim() >= 1 and x.shape[0] == B:
                assert torch.equal(x[perm], y), ("eval-mode output is not batch-permutation equivariant", i)
    return eng


def deterministic_case(lib, dev, c, seed=3, perceptual=False, tight=False, reps=3):
    """caddy_set_deterministic: two backward passes over the same forward give BIT-IDENTICAL flat gradients (slabs + fixed-order reduces instead of fp32
    atomics in arrival order); the backward is then exactly linear in the loss weights for a power-of-two factor; and the deterministic gradient agrees with the default
    (atomic) one up to its run-to-run noise.  Returns the measured distances."""
    from playablevideogeneration_amd.init import init_parameters, random_vgg19_state
    B, T, K, Da, S, Hh, W = c["B"], c["T"], c["K"], c["Da"], c["S"], c["H"], c["W"]
    eng = make_engine(c, lib, dev, perceptual=perceptual)
    init_parameters(eng, seed)
    if perceptual:
        eng.load_vgg(random_vgg19_state(0))
    g = torch.Generator(device=dev).manual_seed(seed)
    obs = torch.rand(B, T, 3 * S, Hh, W, device=dev, generator=g) * 2 - 1
    n = T - 1
    noise = {"eps_states": torch.randn(B * T, Da, device=dev, generator=g), "eps_dirs": torch.randn(B * n, Da, device=dev, generator=g),
             "gumbel_uniform": torch.rand(B * n, K, device=dev, generator=g),
             "eps_states_rec": torch.randn(B * T, Da, device=dev, generator=g), "eps_dirs_rec": torch.randn(B * n, Da, device=dev, generator=g)}
    eng.forward_full(obs, c["gt"], c["tau"], noise, training=True, fetch_outputs=False)
    w1 = dict(H.LOSS_W, perceptual=1.0) if perceptual else dict(H.LOSS_W)
    rel = lambda a, b: ((a - b).double().norm() / b.double().norm()).item()
    eng.set_deterministic(False)
    eng.loss_backward(w1, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
    g_atomic = eng.grads.clone()
    eng.set_deterministic(True)
    gs = []
    for _ in range(reps):
        eng.loss_backward(w1, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
        gs.append(eng.grads.clone())
    assert all(torch.equal(gs[0], g_) for g_ in gs[1:]), ("deterministic backward not bit-reproducible", [rel(g_, gs[0]) for g_ in gs[1:]])
    assert torch.isfinite(gs[0]).all() and gs[0].abs().max().item() > 0
    eng.loss_backward({k: 2 * v for k, v in w1.items() if k != "mi_entropy"}, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
    lin = rel(eng.grads, 2 * gs[0])
    # all weights x2: every gradient seed doubles exactly (power of two) and every kernel is linear in its gradient operand -- up to the split-bf16 rounding of the gradient
    # operands, which is scale-invariant as well: exact equality
    assert lin == 0.0 or (not tight and lin < 1e-6), ("deterministic backward not linear in the loss weights", lin)
    noise_atomic = rel(g_atomic, gs[0])
    assert noise_atomic < 1e-3, ("deterministic vs atomic gradients", noise_atomic)
    return dict(atomic_vs_deterministic=noise_atomic, linearity=lin)


def full_geometry_case(lib, dev, c):
    """The BASELINE geometry against the fp32 oracle.  For t >= gt_init the model runs closed-loop (D -> E feedback, train-mode BatchNorm
    over a batch of 2): round-off is amplified ~1.7x per step -- measured for the ORACLE ITSELF, fp32 vs fp64 at this geometry: 1.2e-5 max
    error for t < 6 growing to 3.2e-3 (MSE 1.6e-7) at t = 14.  Two fp32 trajectories therefore differ by up to ~2x that, so the bounds are:
    action indices bit-exact; teacher-forced steps and every open-loop output within 2e-4; closed-loop frames: per-(b,t) MSE < 1e-6
    (north star: 1e-5) and max error < 2e-2; closed-loop states / hidden states within 1e-2 of their range."""
    d, P, obs = H.inputs_of(c)
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        oout = O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True).forward_full(obs, c["gt"], tau=c["tau"], noise=nz)
    eng = make_engine(c, lib, dev)
    eng.load_state_dict(P)
    out = eng.forward_full(obs, c["gt"], c["tau"], noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"]), training=True)
    gt = c["gt"]
    assert torch.equal(out[5].cpu(), oout[5]), "action indices"
    f, fo = out[0].cpu(), oout[0]
    assert (f[:, :gt] - fo[:, :gt]).abs().max().item() < 2e-4                         # teacher-forced reconstructions
    mse = ((f - fo) ** 2).mean(dim=(2, 3, 4))
    assert mse.max().item() < 1e-6 and (f - fo).abs().max().item() < 2e-2, (mse.max().item(), (f - fo).abs().max().item())
    for i in (3, 6, 7, 8, 10, 11, 12, 13, 14):                                        # open-loop outputs: states, logits, samples, attention, A's distributions
        _cmp(out[i], oout[i], 2e-4, f"open-loop output {i}")
    for i in (2, 4, 9):                                                               # closed-loop: reconstructed states, hidden states, reconstructed attention
        a, b = out[i].cpu(), oout[i]
        assert (a - b).abs().max().item() < 1e-2 * max(1.0, b.abs().max().item()), (i, (a - b).abs().max().item())
    return eng


def full_geometry_grad_case(lib, dev, c):
    """Backward at the BASELINE geometry (batch 1).  BPTT through 10 closed-loop steps with train-mode BatchNorm is ill-conditioned in
    fp32: the ORACLE's own fp32 gradients are 9 % (E / R / D) and 64 % (A) away from its fp64 gradients in relative L2 at this geometry.
    A structural error in a kernel that only large maps reach (narrow / 3-channel MFMA kernels, tile-resident and time-batched wgrad)
    would show as O(1) on the affected module, so: per-module relative L2 distance to the fp64 oracle <= max(2 x the fp32 oracle's, 0.15)."""
    import collections
    d, P, obs = H.inputs_of(c)
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True).forward_full(obs, c["gt"], tau=c["tau"], noise=nz)
    P64, _ = _oracle_run(c, d, P, obs, torch.float64, nz.record)
    P32, _ = _oracle_run(c, d, P, obs, torch.float32, nz.record)
    eng = make_engine(c, lib, dev)
    eng.load_state_dict(P)
    eng.forward_full(obs, c["gt"], c["tau"], noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"]), training=True, fetch_outputs=False)
    eng.loss_backward(H.LOSS_W, smooth_mi=True, mi_alpha=0.2)
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
    for n, _ in O.param_table(d):
        if not O.is_trainable(n) or P64[n].grad is None:
            continue
        r = P64[n].grad
        g, o = eng.grad_view(n).cpu().double(), P32[n].grad.double()
        a = agg[n.split(".")[0]]
        a[0] += ((g - r) ** 2).sum().item(); a[1] += ((o - r) ** 2).sum().item(); a[2] += (r ** 2).sum().item()
    res = {k: ((a[0] / a[2]) ** 0.5, (a[1] / a[2]) ** 0.5) for k, a in agg.items()}
    for k, (hip, orc) in res.items():
        assert hip <= max(2 * orc, 0.15), (k, hip, orc, res)
    return res


def split_vs_exact_case(lib, dev, c, seed=3):
    """The arithmetic of the MI355X default (split-f16 forward / split-bf16 gradient operands on the 16-bit matrix pipe, conv_hx.hip) against the
    exact-fp32 kernels ON THE SAME DEVICE, same inputs, same geometry -- no oracle, no host arithmetic in between.  Two comparisons per module
    (E / A / R / D parameter groups), relative L2 and cosine of the flat gradient:
      (a) backward only: forward split-f16 in both runs (bit-identical activations), gradients split-bf16 vs exact fp32 -- isolates the
          8+8-bit gradient operands;
      (b) everything: default vs exact fp32 forward AND backward (the forwards differ by fp32-reordering-class round-off, which BPTT through
          the closed-loop steps and LeakyReLU slope decisions amplify -- the yardstick is the same comparison between two exact-fp32 runs whose
          summation order differs, which this library cannot produce, so (b) is reported and loosely bounded)."""
    import collections
    from playablevideogeneration_amd.init import init_parameters
    B, T, K, Da, S, Hh, W = c["B"], c["T"], c["K"], c["Da"], c["S"], c["H"], c["W"]
    eng = make_engine(c, lib, dev)
    init_parameters(eng, seed)
    g = torch.Generator(device=dev).manual_seed(seed)
    obs = torch.rand(B, T, 3 * S, Hh, W, device=dev, generator=g) * 2 - 1
    n = T - 1
    noise = {"eps_states": torch.randn(B * T, Da, device=dev, generator=g), "eps_dirs": torch.randn(B * n, Da, device=dev, generator=g),
             "gumbel_uniform": torch.rand(B * n, K, device=dev, generator=g),
             "eps_states_rec": torch.randn(B * T, Da, device=dev, generator=g), "eps_dirs_rec": torch.randn(B * n, Da, device=dev, generator=g)}
    saved = eng.params.clone()

    def run(fwd, bwd):
        eng.params.copy_(saved)
        eng.set_precision(fwd, bwd)
        out = eng.forward_full(obs, c["gt"], c["tau"], noise, training=True)
        l = eng.loss_backward(H.LOSS_W, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
        return out, l, eng.grads.clone()
    o_s, l_s, g_s = run(16, 17)          # the default
    o_b, l_b, g_b = run(16, 0)           # same forward, exact-fp32 gradients
    o_e, l_e, g_e = run(0, 0)            # exact fp32 everywhere
    eng.set_precision(16, 17)
    assert torch.equal(o_s[0], o_b[0]), ("same forward arithmetic, different frames", (o_s[0] - o_b[0]).abs().max().item())          # (a) really shares the forward
    assert abs(l_s["total"] - l_b["total"]) <= 1e-12 * abs(l_b["total"]), (l_s["total"], l_b["total"])      # (fp64 atomics of the loss sums: equal up to their order)
    assert torch.equal(o_s[5], o_e[5]), "action indices: split-f16 forward vs exact fp32"
    fmse = ((o_s[0] - o_e[0]) ** 2).mean(dim=(2, 3, 4)).max().item()
    res = {"frame_mse_split_vs_exact": fmse, "loss_split": l_s["total"], "loss_exact": l_e["total"]}
    assert fmse < 1e-6 and abs(l_s["total"] - l_e["total"]) < 1e-5 * max(1.0, abs(l_e["total"]))
    groups = collections.OrderedDict()
    for name, off, shape, kind in eng.table:
        if kind != 0:
            continue
        k = 1
        for s_ in shape:
            k *= s_
        groups.setdefault(name.split(".")[0], []).append((off, k))
    for tag, ref in (("bwd_only", g_b), ("fwd_and_bwd", g_e)):
        for mod, rng in groups.items():
            a = torch.cat([g_s[o:o + k] for o, k in rng]).double()
            b = torch.cat([ref[o:o + k] for o, k in rng]).double()
            if b.norm().item() == 0.0:
                continue
            res[f"{tag}:{mod}"] = dict(rel_l2=((a - b).norm() / b.norm()).item(), cosine=(a @ b / (a.norm() * b.norm())).item())
    return res


def random_config_sweep(lib, dev, n, seed):
    """seeded random model configurations (variant, K, Da, observation_stacking 1-4, batch, T, geometry, gt, soft/hard Gumbel) vs the oracle:
    outputs, action indices, frame MSE, loss.  Batch >= 2 and maps >= 4x4 (BatchNorm over 2 elements is not a meaningful comparison)."""
    import random
    rng = random.Random(seed)
    for i in range(n):
        variant = rng.choice(["main", "reduced"])
        T = rng.choice([2, 3, 4, 5])
        c = dict(variant=variant, K=rng.choice([2, 3, 7]), Da=rng.choice([1, 2, 5]), Ch=128 if variant == "main" else 64, S=rng.choice([1, 2, 3, 4]),
                 B=rng.choice([2, 3]), T=T, H=rng.choice([32, 48]), W=rng.choice([32, 48, 64]), gt=rng.randint(1, max(1, T - 1)),
                 tau=rng.choice([0.4, 0.9]), hard=rng.random() < 0.3)
        try:
            oracle_case(lib, dev, c)
        except AssertionError as e:
            raise AssertionError((i, c, e))


def vgg_s16_ab_case(lib, dev, c, lam=1.0, force_big=False):
    """Round 5: VGG19 feature maps / feature gradients exchanged as S16 tensors (pre-split 16-bit operand pairs, csrc/common.h) against the fp32 form of the same library
    (caddy_debug_set_vgg_s16).  The convolutions see bit-identical operands either way; the feature L1, its sign() seeds and the ReLU / max-pool routing see hi + lo instead of
    the fp32 value (2^-22 relative): losses agree to 1e-6, the gradient w.r.t. the reconstructions up to a handful of flipped sign / arg-max decisions.
    force_big: caddy_k_hx_force_big(1) puts every launch on the well-filled tile variants, so that small frames exercise the S16 path of every layer."""
    d, P, obs = H.inputs_of(c)
    V = O.make_vgg_params()
    nz = O.Noise()
    torch.manual_seed(H.NOISE_SEED)
    with torch.no_grad():
        O.Oracle(d, {k: v.clone() for k, v in P.items()}, training=True).forward_full(obs, c["gt"], tau=c["tau"], noise=nz)
    nd = noise_dict(nz.record, c["B"], c["T"], c["K"], c["Da"])
    w = dict(H.LOSS_W, perceptual=lam)
    lib.caddy_debug_set_vgg_s16.argtypes = [C.c_void_p, C.c_int]
    res = []
    if force_big:
        lib.caddy_k_hx_force_big(1)
    try:
        for s16 in (0, 1):
            eng = make_engine(c, lib, dev, perceptual=True)
            eng.load_state_dict(P)
            eng.load_vgg(V)
            lib.caddy_debug_set_vgg_s16(eng.ctx, s16)
            out = eng.forward_full(obs, c["gt"], c["tau"], nd, training=True)
            losses = eng.loss_backward(w, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
            grads = eng.grads.clone().cpu()
            lib.caddy_debug_set_seeds_only.argtypes = [C.c_void_p, C.c_int]
            lib.caddy_debug_set_seeds_only(eng.ctx, 1)
            eng.loss_backward(w, smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
            seeds = [eng.output_grad(100 + r, out[1][r]).cpu() for r in range(3)]
            lib.caddy_debug_set_seeds_only(eng.ctx, 0)
            res.append((out[0].cpu(), losses, grads, seeds))
            del eng
    finally:
        if force_big:
            lib.caddy_k_hx_force_big(-1)
    (f0, l0, g0, s0), (f1, l1, g1, s1) = res
    assert torch.equal(f0, f1)
    for k in l0:
        if k.startswith("perceptual") or k == "total":
            assert abs(l0[k] - l1[k]) <= 1e-6 * abs(l0[k]) + 1e-12, (k, l0[k], l1[k])
    info = {}
    for r in range(3):
        a, b = s0[r].flatten(0, 1).double(), s1[r].flatten(0, 1).double()
        per = torch.tensor([((a[i] - b[i]).norm() / a[i].norm()).item() for i in range(a.shape[0])])
        info[f"seed_r{r}_median"], info[f"seed_r{r}_worst"] = per.median().item(), per.max().item()
        assert per.median().item() < 1e-3 and per.max().item() < 5e-2, (r, per.tolist())
    rel = ((g0 - g1).double().norm() / g0.double().norm()).item()
    info["param_grad_rel_l2"] = rel
    assert rel < 2e-2, rel      # (BPTT amplifies the few flipped decisions; the two forms are two fp32-class evaluations of the same function)
    return info


def s16_grads_ab_case(lib, dev, c, seed=3, min_count=1, pretraining=False):
    """Round 6: the gradients of the convolution outputs written PRE-SPLIT (S16-bf16) by their point-wise producers -- BatchNorm backward, average-pool backward, ConvLSTM cell
    backward -- and copied by the dgrad / weight-gradient launches that stage them, against the fp32 exchange of the same library (caddy_debug_set_s16_grads), same inputs.  The
    matrix operands are the same bit for bit (kernel-level: s16_grad_case); what differs is what the bias / broadcast-input column sums see (hi + lo: 2^-17 relative), which
    reaches every parameter through the action network's backward: relative L2 of the flat gradient << the backward's own 16-bit-operand error (1e-5 class)."""
    from playablevideogeneration_amd.init import init_parameters
    B, T, K, Da, S, Hh, W = c["B"], c["T"], c["K"], c["Da"], c["S"], c["H"], c["W"]
    lib.caddy_debug_set_s16_grads.argtypes = [C.c_void_p, C.c_int]
    lib.caddy_debug_s16_grad_count.argtypes = [C.c_void_p]
    lib.caddy_debug_s16_grad_count.restype = C.c_long
    eng = make_engine(c, lib, dev)
    init_parameters(eng, seed)
    g = torch.Generator(device=dev).manual_seed(seed)
    obs = torch.rand(B, T, 3 * S, Hh, W, device=dev, generator=g) * 2 - 1
    n = T - 1
    noise = {"eps_states": torch.randn(B * T, Da, device=dev, generator=g), "eps_dirs": torch.randn(B * n, Da, device=dev, generator=g),
             "gumbel_uniform": torch.rand(B * n, K, device=dev, generator=g),
             "eps_states_rec": torch.randn(B * T, Da, device=dev, generator=g), "eps_dirs_rec": torch.randn(B * n, Da, device=dev, generator=g)}
    saved = eng.params.clone()
    res = []
    for on in (0, 1, 1):
        eng.params.copy_(saved)
        lib.caddy_debug_set_s16_grads(eng.ctx, on)
        if pretraining:
            out = eng.forward_pretraining(obs, c["tau"], noise, training=True)
        else:
            out = eng.forward_full(obs, c["gt"], c["tau"], noise, training=True)
        l = eng.loss_backward(dict(H.LOSS_W, **({"hidden": 1.0} if pretraining else {})), smooth_mi=True, mi_alpha=0.2, update_mi_ema=False)
        res.append((out[0].clone(), l, eng.grads.clone(), int(lib.caddy_debug_s16_grad_count(eng.ctx))))
    (f0, l0, g0, n0), (f1, l1, g1, n1), (f2, l2, g2, n2) = res
    assert n0 == 0 and n1 >= min_count and n2 == n1, (n0, n1, n2)
    assert torch.equal(f0, f1) and abs(l0["total"] - l1["total"]) <= 1e-12 * abs(l0["total"])
    assert torch.isfinite(g1).all()
    assert torch.equal(g1, g2), "bit-reproducible backward with pre-split gradients"
    rel = ((g0 - g1).double().norm() / g0.double().norm()).item()
    worst = 0.0
    for name, off, shape, kind in eng.table:
        if kind != 0:
            continue
        k = 1
        for s_ in shape:
            k *= s_
        a, b = g0[off:off + k].double(), g1[off:off + k].double()
        if a.norm().item() > 0:
            worst = max(worst, ((a - b).norm() / a.norm()).item())
    assert rel < 1e-4 and worst < 1e-2, (rel, worst)
    return {"pre_split_gradient_tensors": n1, "flat_gradient_rel_l2_vs_fp32_exchange": rel, "worst_parameter_rel_l2": worst}
