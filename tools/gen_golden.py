"""Generate tests/golden/*.npz from the REAL reference (build container only; needs /root/reference).

Each fixture stores: the case description (so inputs/weights/noise are re-derived from seeds with
oracle.caddy_oracle.make_params / torch generators -- no reference source or weights are stored), the 20 outputs of
reference Model.forward, the loss terms of the reference loss classes (training/losses.py, VGG term excluded),
per-parameter gradient summaries, BN buffers / centroids / MI-EMA after the step, and eval-mode roll-out frames
from Model.generate_next.    Usage:  python tools/gen_golden.py
"""
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import ref_harness as rh  # noqa: E402
from oracle import caddy_oracle as O  # noqa: E402

CASES = {
    # name: variant, K, Da, Ch, S, B, T, H, W, gt, tau, hard, pretraining
    "full_reduced_s1": dict(variant="reduced", K=3, Da=1, Ch=64, S=1, B=2, T=4, H=32, W=32, gt=2, tau=0.7, hard=False, pre=False),
    "full_main_s1": dict(variant="main", K=7, Da=2, Ch=128, S=1, B=2, T=5, H=32, W=32, gt=3, tau=0.4, hard=False, pre=False),
    "full_main_s4_hard": dict(variant="main", K=7, Da=5, Ch=128, S=4, B=2, T=6, H=32, W=48, gt=2, tau=0.9, hard=True, pre=False),
    "pre_main_s4": dict(variant="main", K=7, Da=5, Ch=128, S=4, B=2, T=4, H=32, W=32, gt=0, tau=1.0, hard=False, pre=True),
    # with the VGG19 perceptual term (ParallelPerceptualLoss on the seeded weights of oracle.make_vgg_params); 64x64 is the smallest frame
    # whose quarter-resolution output still reaches relu5_1 (four 2x2 max-pools)
    "perc_main_s1": dict(variant="main", K=7, Da=2, Ch=128, S=1, B=2, T=4, H=64, W=64, gt=2, tau=0.6, hard=False, pre=False, perc=1.0),
    "perc_pre_reduced_s1": dict(variant="reduced", K=3, Da=1, Ch=64, S=1, B=2, T=3, H=64, W=80, gt=0, tau=1.0, hard=False, pre=True, perc=0.5),
    # configuration branches of the reference that the cases above do not take: the PLAIN MutualInformationLoss of training.trainer (03_tennis.yaml;
    # losses.py:238-302, no estimator state), use_gumbel: False (plain softmax as the action assignment, model.py:177-179) and
    # use_variations: False (zeroed action variations, model.py:188-189)
    "full_reduced_s1_plainmi": dict(variant="reduced", K=3, Da=1, Ch=64, S=1, B=2, T=4, H=32, W=32, gt=2, tau=0.7, hard=False, pre=False, mi="plain"),
    "full_main_s1_nogumbel": dict(variant="main", K=7, Da=2, Ch=128, S=1, B=2, T=4, H=32, W=32, gt=2, tau=0.4, hard=False, pre=False, use_gumbel=False),
    "full_reduced_s1_novar": dict(variant="reduced", K=3, Da=1, Ch=64, S=1, B=2, T=4, H=32, W=32, gt=2, tau=0.7, hard=False, pre=False, use_variations=False),
    # model.action_network.ensamble_size: 2 (model.py:28,47): two action networks, `random.choice` draws the one both A calls of the pass use (model.py:152,274); rseed makes
    # Python's `random` draw member 1, so that a driver that ignores the member cannot pass
    "full_reduced_s1_ens2": dict(variant="reduced", K=3, Da=1, Ch=64, S=1, B=2, T=4, H=32, W=32, gt=2, tau=0.7, hard=False, pre=False, ens=2, rseed=7),
}
INTERP = [(1, 2, 0.3), (0, 2, 0.8)]           # (first_action, second_action, interpolation_factor) appended to the roll-out cases
SAMPLER_CASES = {
    "eval_main_s1_onehot_zero": dict(variant="main", K=7, Da=2, Ch=128, S=1, B=2, T=5, H=32, W=32, gt=1, tau=1.0, hard=False, pre=False, sampler="onehot", zero_var=True),
    "eval_reduced_s1_gt": dict(variant="reduced", K=3, Da=1, Ch=64, S=1, B=2, T=4, H=32, W=32, gt=2, tau=1.0, hard=False, pre=False, sampler="gt", zero_var=False),
}
LOSS_W = dict(O.DEFAULT_LOSS_WEIGHTS, state_kl=1e-5, entropy=0.01)
PARAM_SEED, OBS_SEED, NOISE_SEED = 7, 1, 5


def case_inputs(c):
    cfg = rh.make_config(variant=c["variant"], actions=c["K"], action_dim=c["Da"], hidden=c["Ch"], stacking=c["S"],
                         state_res=(c["H"] // 8, c["W"] // 8), hard_gumbel=c["hard"], use_gumbel=c.get("use_gumbel", True),
                         use_variations=c.get("use_variations", True), ensamble_size=c.get("ens", 1))
    d = O.Dims.from_config(cfg)
    P = O.make_params(d, seed=PARAM_SEED)
    obs = torch.rand(c["B"], c["T"], 3 * c["S"], c["H"], c["W"], generator=torch.Generator().manual_seed(OBS_SEED)) * 2 - 1
    return cfg, d, P, obs


def flat_outputs(out):
    res = {}
    for i, o in enumerate(out):
        if isinstance(o, (list, tuple)):
            for j, x in enumerate(o):
                res[f"out{i}_{j}"] = x.detach().numpy()
        else:
            res[f"out{i}"] = o.detach().numpy()
    return res


def main():
    rh.install()
    import training.losses as RL
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    only = set(sys.argv[1:])           # optional: generate only the named fixtures
    for name, c in CASES.items():
        if only and name not in only:
            continue
        cfg, d, P, obs = case_inputs(c)
        ref = rh.build_reference_model(cfg, P)
        ref.train()
        acts = torch.zeros(c["B"], c["T"], dtype=torch.int32)
        torch.manual_seed(NOISE_SEED)
        random.seed(c.get("rseed", NOISE_SEED))
        if c.get("ens", 1) > 1:      # which member will `random.choice(self.action_network)` draw?  (recorded in the fixture; the tests re-derive it from the same seed)
            st_ = random.getstate(); data_member = random.choice(range(c["ens"])); random.setstate(st_)
        out = ref((obs, acts, None, None), c["gt"], pretraining=c["pre"], gumbel_temperature=c["tau"])
        data = flat_outputs(out)
        if c.get("ens", 1) > 1:
            data["member"] = np.array(data_member)
        data["case"] = np.array(repr(c))
        if not c["pre"] or c.get("perc"):
            smi = RL.MutualInformationLoss() if c.get("mi") == "plain" else RL.SmoothMutualInformationLoss(cfg)
            rec = [RL.ObservationsLoss()(obs, m) for m in out[1]]
            li, lo = (7, 15) if c["pre"] else (6, 15)       # logits / reconstructed logits in the two tuple layouts
            comp = {
                "rec": (sum(r.double() for r in rec) / 3),
                "states": RL.StatesLoss()(out[3].detach(), out[2]),
                "entropy": RL.EntropyLogitLoss()(out[li]),
                "dir_kl": RL.KLGaussianDivergenceLoss()(out[10]),
                "mi": smi(torch.softmax(out[li], -1), torch.softmax(out[lo], -1), lamb=LOSS_W["mi_entropy"]),
                "state_kl": RL.KLGeneralGaussianDivergenceLoss()(out[18], out[12].detach()),
            }
            if c["pre"]:
                comp["hidden"] = RL.HiddenStatesLoss()(out[5], out[4].detach())      # trainer.py:313
            total = sum(LOSS_W[k] * v for k, v in comp.items())
            if c.get("perc"):
                # the perceptual part of Trainer.compute_losses (trainer.py:442-466,494-500) with the reference's own classes
                from training.trainer import Trainer
                ppl = RL.ParallelPerceptualLoss()
                p_acc = torch.zeros((1,), dtype=float)
                p_term = torch.zeros((1,), dtype=float)
                for r, m in enumerate(out[1]):
                    m.retain_grad()
                    tot, comps = ppl(obs, m, None)
                    p_acc += tot
                    p_term += Trainer.sum_loss_components(None, comps, c["perc"])
                    data[f"perceptual_loss_r{r}"] = np.array(tot.item())
                    for l, cc in enumerate(comps):
                        data[f"perceptual_loss_r{r}_l{l}"] = np.array(cc.item())
                p_acc /= 3
                p_term /= 3
                comp["perceptual"] = p_acc
                data["loss_perceptual_term"] = np.array(p_term.item())
                total = total + p_term
            total.backward()
            if c.get("perc"):
                for r, m in enumerate(out[1]):
                    data[f"dout1_{r}"] = m.grad.numpy()
            data["loss_total"] = np.array(total.item())
            for k, v in comp.items():
                data["loss_" + k] = np.array(v.item())
            if c.get("mi") != "plain":
                data["mi_ema"] = smi.matrix_estimator.estimated_matrix.detach().numpy()
            names, gsum, gabs, gfirst = [], [], [], []
            for n, p in ref.named_parameters():
                if p.grad is None:
                    continue
                names.append(n)
                gsum.append(p.grad.double().sum().item())
                gabs.append(p.grad.double().abs().sum().item())
                gfirst.append(p.grad.flatten()[:4].tolist() + [0.0] * max(0, 4 - p.grad.numel()))
            data["grad_names"] = np.array(names)
            data["grad_sum"], data["grad_abs"], data["grad_first4"] = np.array(gsum), np.array(gabs), np.array(gfirst, dtype=np.float32)
        sd = ref.state_dict()
        for k, v in sd.items():
            if O.is_buffer(k) and not k.endswith("num_batches_tracked"):
                data["buf:" + k] = v.numpy()
        data["centroids"] = sd["centroid_estimator.estimated_centroids"].numpy()
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **data)
        print(name, "written", sum(v.nbytes for v in data.values()) // 1024, "KiB raw")

    # eval-mode roll-out (play.py path): start_inference + N x generate_next, zero variation
    for name, c in {"rollout_main_s4": dict(variant="main", K=7, Da=5, Ch=128, S=4, H=32, W=32, steps=4),
                    "rollout_reduced_s1": dict(variant="reduced", K=3, Da=1, Ch=64, S=1, H=32, W=48, steps=4)}.items():
        if only and name not in only:
            continue
        cc = dict(c, B=1, T=1, hard=False)
        cfg, d, P, obs = case_inputs(cc)
        ref = rh.build_reference_model(cfg, P)
        ref.eval()
        o = obs[0, 0]
        frames = []
        torch.manual_seed(NOISE_SEED)
        with torch.no_grad():
            ref.start_inference()
            for i in range(c["steps"]):
                f, o = ref.generate_next(o, i % c["K"])
                frames.append(f.numpy())
            # generate_next_interpolation (model.py:609-655) and generate_next(noise=True) continue the same sequence
            interp = []
            for (a1, a2, al) in INTERP:
                f, o2 = ref.generate_next_interpolation(o, a1 % c["K"], a2 % c["K"], al)
                interp.append(f.numpy())
            torch.manual_seed(NOISE_SEED + 1)              # generate_next(noise=True): the action variation is drawn from N(0, 1) (model.py:590)
            fn, _ = ref.generate_next(o, 1, noise=True)
        np.savez_compressed(os.path.join(outdir, name + ".npz"), case=np.array(repr(cc | {"steps": c["steps"]})),
                            frames=np.stack(frames), last_obs=o.numpy(), interp_frames=np.stack(interp), noise_frame=fn.numpy())
        print(name, "written")

    # eval-mode forward_full_model with the evaluation samplers (evaluation/evaluator.py:126, evaluation_dataset_builder.py:57-63)
    import evaluation.action_sampler as AS
    import evaluation.action_variation_sampler as AVS
    for name, c in SAMPLER_CASES.items():
        if only and name not in only:
            continue
        cfg, d, P, obs = case_inputs(c)
        ref = rh.build_reference_model(cfg, P)
        ref.eval()
        acts = (torch.arange(c["B"] * c["T"]).reshape(c["B"], c["T"]) % c["K"]).to(torch.int32)
        sampler = AS.OneHotActionSampler() if c["sampler"] == "onehot" else AS.GroundTruthActionSampler({i: (i + 1) % c["K"] for i in range(c["K"])})
        torch.manual_seed(NOISE_SEED)
        random.seed(NOISE_SEED)
        with torch.no_grad():
            out = ref((obs, acts, None, None), c["gt"], gumbel_temperature=c["tau"], action_sampler=sampler,
                      action_variation_sampler=AVS.ZeroActionVariationSampler() if c["zero_var"] else None)
        data = flat_outputs(out)
        data["case"] = np.array(repr(c))
        np.savez_compressed(os.path.join(outdir, name + ".npz"), **data)
        print(name, "written")


if __name__ == "__main__":
    main()
