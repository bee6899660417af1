"""Counterpart of training/trainer.py for the HIP path: `config["training"]["trainer"] = "playablevideogeneration_amd.trainer"`.

Same schedules (trainer.py:124-165), same loss weighting (:494-500, including the VGG19 perceptual term of ParallelPerceptualLoss,
:442-466), same optimiser (Adam lr / weight_decay, MultiStepLR; :36-37,584-587) and checkpoint keys (:100), but losses + backward +
Adam run fused inside libcaddy_hip.so and only one small buffer of loss scalars crosses to the host per step.

VGG19 weights: the reference downloads torchvision's pretrained VGG19 at construction (model/layers/vgg.py:16).  Here they come from
`config["training"]["vgg19_weights"]` (path of a torchvision `vgg19().state_dict()` / `.features.state_dict()` file, or the dict itself) or,
with the explicit opt-in `training.vgg19_from_torchvision: true`, from an importable torchvision; they are only resolved when a perceptual lambda is
non-zero (or `training.log_perceptual: true` asks for the logged value of a zero-weight term).  With a non-zero `perceptual_loss_lambda[_pretraining]`
and no weights the constructor RAISES instead of silently training a different objective.
"""
import math
import os
from typing import Dict

import torch


class Trainer:
    SMOOTH_MI = False

    def __init__(self, config, model, dataset, logger):
        self.config, self.dataset, self.logger = config, dataset, logger
        tr = config["training"]
        self.lr, self.weight_decay = tr["learning_rate"], tr["weight_decay"]
        self.lr_schedule, self.lr_gamma = list(tr["lr_schedule"]), tr["lr_gamma"]
        self.global_step = 0
        self.action_mutual_infromation_entropy_lambda = tr.get("action_mutual_information_entropy_lambda", 1.0)   # configuration.py:79-80
        b = tr["batching"]
        self.observations_count_start, self.observations_count_end, self.observations_count_steps = b["observations_count_start"], b["observations_count"], b["observations_count_steps"]
        self.real_observations_start, self.real_observations_end, self.real_observations_steps = tr["ground_truth_observations_start"], tr["ground_truth_observations_end"], tr["ground_truth_observations_steps"]
        self.gumbel_temperature_start, self.gumbel_temperature_end, self.gumbel_temperature_steps = tr["gumbel_temperature_start"], tr["gumbel_temperature_end"], tr["gumbel_temperature_steps"]
        self.mi_alpha = tr.get("mutual_information_estimation_alpha", 0.2)
        # options of the reference this path does not implement must not be ignored silently
        if tr.get("use_motion_weights", False):
            raise Exception("training.use_motion_weights is not supported by playablevideogeneration_amd.trainer (MotionLossWeightMaskCalculator, training/losses.py:591-649)")
        # ensemble of action networks (model.py:28,47,152): torch.optim.Adam keeps `step` per parameter and skips parameters without a gradient, so every member carries its own
        # step count (= the number of passes it was drawn for)
        self.member_steps = [0] * int(config["model"]["action_network"].get("ensamble_size", 1))
        # optimizer.zero_grad() (training/trainer.py:584) under the two torch generations.  "set_to_none" (default; torch >= 2.0, what the reference does on a current install and
        # what the trainer goldens were generated under): a parameter the last backward did not reach -- an ensemble member that was not drawn, state_to_hidden_state_layer
        # after a full-model pass -- has .grad None and Adam skips it.  "zero_fill" (torch < 2.0; the reference's env.yml pins pytorch 1.4.0): once such a parameter has had a
        # gradient its .grad stays a zero tensor, so every later optimizer.step() decays it, ages its moments and counts a step.
        self.zero_grad_semantics = str(tr.get("zero_grad_semantics", "set_to_none"))
        if self.zero_grad_semantics not in ("set_to_none", "zero_fill"):
            raise Exception(f"training.zero_grad_semantics must be 'set_to_none' or 'zero_fill', got {self.zero_grad_semantics!r}")
        lw = tr["loss_weights"]
        self.perceptual_lambda = float(lw.get("perceptual_loss_lambda", 0.0))
        self.perceptual_lambda_pretraining = float(lw.get("perceptual_loss_lambda_pretraining", 0.0))
        need_vgg = self.perceptual_lambda != 0.0 or self.perceptual_lambda_pretraining != 0.0 or bool(tr.get("log_perceptual", False))
        self.vgg_state = self._find_vgg_weights(tr) if need_vgg else None      # (a zero-weight term would still cost a full VGG19 forward per step)
        if (self.perceptual_lambda != 0.0 or self.perceptual_lambda_pretraining != 0.0) and self.vgg_state is None:
            raise Exception("loss_weights.perceptual_loss_lambda is non-zero but no VGG19 weights are available.  The reference builds torchvision's pretrained vgg19 "
                            "unconditionally (model/layers/vgg.py:16, a download on first use); this plugin never downloads implicitly.  Either set "
                            "training.vgg19_weights to a torchvision vgg19 state_dict file (vgg19().state_dict() or .features.state_dict()), or set "
                            "training.vgg19_from_torchvision: true to let torchvision load its cached / downloadable weights as the reference does, "
                            "or set the perceptual lambdas to 0")
        if self.vgg_state is not None:
            model.module.enable_perceptual(self.vgg_state)
        self.dataloader = self._build_dataloader(config, dataset)
        self.adam_m = self.adam_v = None
        self.mi_ema = None
        self.opt_steps = 0
        self.s2h_steps = 0      # optimiser steps that followed a PRETRAINING pass: the only ones in which state_to_hidden_state_layer has a gradient (model.py:41-43,413) -- its own Adam step count

    @staticmethod
    def _find_vgg_weights(tr):
        src = tr.get("vgg19_weights", None)
        if isinstance(src, dict):
            return src
        if isinstance(src, str):
            sd = torch.load(src, map_location="cpu", weights_only=True)
            return sd.get("state_dict", sd) if isinstance(sd, dict) else sd
        if not tr.get("vgg19_from_torchvision", False):
            return None
        try:                                                   # explicit opt-in only: what the reference does (torchvision + its cached / downloadable weights); never implicit --
            from torchvision import models                     # on an air-gapped box the constructor would block on the download
            return models.vgg19(pretrained=True).features.state_dict()
        except Exception:
            return None

    @staticmethod
    def _build_dataloader(config, dataset):
        """training/trainer.py:39: DataLoader(dataset, batch_size, shuffle=True, collate_fn=single_batch_elements_collate_fn, num_workers, pin_memory,
        drop_last=True).  Datasets of BatchElements (anything with the reference's `dataset.batching` contract) get the mirror's collate function;
        an iterable of ready batch tuples (synthetic benches, tests) is used as it is."""
        if dataset is None or not hasattr(dataset, "__getitem__") or not hasattr(dataset, "__len__"):
            return None
        b = config["training"]["batching"]
        try:
            from .batching import single_batch_elements_collate_fn, is_batch_element
            if len(dataset) == 0 or not is_batch_element(dataset[0]):
                return None
        except Exception:
            return None
        from torch.utils.data import DataLoader
        nw = int(b.get("num_workers", 0))
        return DataLoader(dataset, batch_size=b["batch_size"], shuffle=True, collate_fn=single_batch_elements_collate_fn, num_workers=nw,
                          pin_memory=torch.cuda.is_available(), drop_last=True)

    # ---- schedules: training/trainer.py:124-165 ----
    def get_ground_truth_observations_count(self) -> int:
        v = self.real_observations_start - (self.real_observations_start - self.real_observations_end) * self.global_step / self.real_observations_steps
        return max(self.real_observations_end, math.ceil(v))

    def get_gumbel_temperature(self) -> float:
        v = self.gumbel_temperature_start - (self.gumbel_temperature_start - self.gumbel_temperature_end) * self.global_step / self.gumbel_temperature_steps
        return max(self.gumbel_temperature_end, v)

    def get_observations_count(self) -> int:
        v = self.observations_count_start + (self.observations_count_end - self.observations_count_start) * self.global_step / self.observations_count_steps
        return min(self.observations_count_end, math.floor(v))

    def _get_current_lr(self) -> float:
        return self.lr * self.lr_gamma ** sum(1 for m in self.lr_schedule if self.opt_steps >= m)     # MultiStepLR

    def loss_weights(self, pretraining=False) -> Dict[str, float]:
        lw = self.config["training"]["loss_weights"]
        sfx = "_pretraining" if pretraining else ""          # training/trainer.py:340-347 vs :494-500
        w = dict(rec=lw["reconstruction_loss_lambda" + sfx], states=lw["states_rec_lambda" + sfx], entropy=lw["entropy_lambda" + sfx],
                 dir_kl=lw["action_directions_kl_lambda" + sfx], mi=lw["action_mutual_information_lambda" + sfx],
                 state_kl=lw["action_state_distribution_kl_lambda" + sfx], mi_entropy=self.action_mutual_infromation_entropy_lambda,
                 perceptual=self.perceptual_lambda_pretraining if pretraining else self.perceptual_lambda)
        if pretraining:
            w["hidden"] = lw["hidden_states_rec_lambda_pretraining"]
        return w

    # ---- logging-only diagnostics of the reference's loss_info (trainer.py:475-491, :358-375), from the small outputs of the forward ----
    def diagnostics(self, model, eng, pretraining=False, li=None) -> Dict[str, float]:
        if not self.config["training"].get("loss_diagnostics", True):
            return {}
        if li is not None and "diagnostics" in li:      # computed on the device by caddy_loss_backward (caddy_loss_cfg.diagnostics): one host buffer per step
            return dict(li["diagnostics"])
        o = (lambda i: eng.output(i, pretraining))
        s = 1 if pretraining else 0                       # forward_pretraining's tuple has one extra entry before the logits / samples
        with torch.no_grad():
            samples = o(7 + s)

            def prob_entropy(p):                           # EntropyProbabilityLoss (losses.py:359-376): no epsilon, as in the reference
                p = p.reshape(-1, p.shape[-1])
                return (-(p * torch.log(p)).sum() / p.shape[0]).item()
            ddist, rdist, var = o(10), o(16), o(14)
            cen = model.module.centroid_estimator.get_estimated_centroids()
            kc = cen.shape[0]
            cdist = (cen.unsqueeze(0) - cen.unsqueeze(1)).pow(2).sum(2).sqrt().sum() / (kc * (kc - 1))
            rv = rdist[:, :, 1].reshape(-1, rdist.shape[-1]); rm = rdist[:, :, 0].reshape(-1, rdist.shape[-1])
            return {"samples_entropy": prob_entropy(samples), "action_distribution_entropy": prob_entropy(samples.mean(dim=(0, 1)).unsqueeze(0)),
                    "states_magnitude": o(3).abs().mean().item(), "hidden_states_magnitude": o(5 if pretraining else 4).abs().mean().item(),
                    "action_directions_mean_magnitude": ddist[:, :, 0].abs().mean().item(), "action_directions_variance_magnitude": ddist[:, :, 1].abs().mean().item(),
                    "reconstructed_action_directions_mean_magnitude": rdist[:, :, 0].abs().mean().item(),
                    "reconstructed_action_directions_variance_magnitude": rdist[:, :, 1].abs().mean().item(),
                    "action_directions_reconstruction_error": (rdist[:, :, 0] - ddist[:, :, 0]).pow(2).mean().item(),
                    "reconstructed_action_directions_kl_loss": (-0.5 * (1 + torch.log(rv) - rm.pow(2) - rv).sum(1)).mean().item(),   # losses.py:146-169
                    "centroids_mean_magnitude": cen.abs().mean().item(), "average_centroids_distance": cdist.item(),
                    "average_action_variations_norm_l2": var.pow(2).sum(-1).sqrt().mean().item(), "action_variations_mean": var.mean().item()}

    # ---- Trainer.compute_losses (trainer.py:400-550): forward + fused losses + backward ----
    def compute_losses(self, model, batch, observations_count: int):
        return self._compute_losses(model, batch, observations_count)()

    def _compute_losses(self, model, batch, observations_count: int, deferred=False):
        """enqueue forward + fused losses + backward; -> a function that returns compute_losses' triple (with `deferred` the GPU may still be running when this
        returns: the loss values arrive by an asynchronous copy and only the returned function waits for them)"""
        gt = self.get_ground_truth_observations_count()
        if gt >= observations_count:
            gt = observations_count - 1
        tau = self.get_gumbel_temperature()
        batch_tuple = batch.to_tuple() if hasattr(batch, "to_tuple") else batch
        model(batch_tuple, gt, gumbel_temperature=tau, fetch_outputs=False)
        eng = model.module.last_engine
        self._to_engine_device(eng)              # BEFORE the engine takes the raw pointer of the MI estimator (a checkpoint loaded before model.cuda() left it on the CPU)
        if self.SMOOTH_MI and self.mi_ema is not None:
            eng.mi_ema = self.mi_ema
        pending = eng.loss_backward(self.loss_weights(), smooth_mi=self.SMOOTH_MI, mi_alpha=self.mi_alpha, perceptual_log=self.vgg_state is not None,
                                    diagnostics=self.config["training"].get("loss_diagnostics", True), deferred=deferred)
        if self.SMOOTH_MI:
            self.mi_ema = eng.mi_ema
        w = self.loss_weights()
        return lambda: self._loss_info(model, eng, pending.result() if hasattr(pending, "result") else pending, w, gt, tau, observations_count)

    def _check_saturation(self, eng, li):
        """f16 range guard of the split-f16 forward (CADDY_LOSS_F16_SATURATED): a forward activation beyond +-65504 was clamped in that step.  Polling the engine moves the layers that
        reported -- only those -- onto a forward without a range limit (exact fp32 for model layers, split bf16 for VGG19 layers); the reference has no such limit (fp32 throughout).
        Every engine polls for itself (the sequence-length curriculum and the evaluators create new ones)."""
        if li.get("f16_saturated"):
            bits = eng.numerics_flags()
            if self.logger is not None:
                self.logger.print(f"warning: a forward activation exceeded the f16 range (|x| > 65504) and was clamped in this step"
                                  + (" -- a NaN among them, the total loss reads NaN" if bits & 2 else "")
                                  + f"; {eng.fallback_layers()} convolution layer(s) of this engine now run without a range limit")

    def _loss_info(self, model, eng, li, w, gt, tau, observations_count):
        self._check_saturation(eng, li)
        loss_info = {"loss_component_observations_rec": w["rec"] * li["rec"], "loss_component_states_rec": w["states"] * li["states"],
                     "loss_component_perceptual_loss": li.get("perceptual_term", 0.0), "avg_perceptual_loss": li.get("perceptual", 0.0),
                     "loss_component_entropy": w["entropy"] * li["entropy"], "loss_component_action_directions_kl_divergence": w["dir_kl"] * li["dir_kl"],
                     "loss_component_action_mutual_information": w["mi"] * li["mi"], "loss_component_action_state_distribution_kl": w["state_kl"] * li["state_kl"],
                     "avg_observations_rec_loss": li["rec"], "states_rec_loss": li["states"], "entropy_loss": li["entropy"],
                     "action_directions_kl_loss": li["dir_kl"], "action_mutual_information_loss": li["mi"], "action_state_distribution_kl_loss": li["state_kl"],
                     "observations_rec_loss_r0": li["l1_r0"], "observations_rec_loss_r1": li["l1_r1"], "observations_rec_loss_r2": li["l1_r2"],
                     "ground_truth_observations": gt, "gumbel_temperature": tau, "observations_count": observations_count}
        loss_info.update({k: v for k, v in li.items() if k.startswith("perceptual_loss_r")})      # trainer.py:459-462
        loss_info.update(self.diagnostics(model, eng, li=li))
        return li["total"], loss_info, {}

    # ---- Trainer.compute_losses_pretraining (trainer.py:241-398) ----
    def compute_losses_pretraining(self, model, batch, observations_count: int):
        return self._compute_losses_pretraining(model, batch, observations_count)()

    def _compute_losses_pretraining(self, model, batch, observations_count: int, deferred=False):
        tau = self.get_gumbel_temperature()
        batch_tuple = batch.to_tuple() if hasattr(batch, "to_tuple") else batch
        model(batch_tuple, pretraining=True, gumbel_temperature=tau, fetch_outputs=False)
        eng = model.module.last_engine
        self._to_engine_device(eng)
        if self.SMOOTH_MI and self.mi_ema is not None:
            eng.mi_ema = self.mi_ema
        w = self.loss_weights(pretraining=True)
        pending = eng.loss_backward(w, smooth_mi=self.SMOOTH_MI, mi_alpha=self.mi_alpha, perceptual_log=self.vgg_state is not None,
                                    diagnostics=self.config["training"].get("loss_diagnostics", True), deferred=deferred)
        if self.SMOOTH_MI:
            self.mi_ema = eng.mi_ema
        return lambda: self._loss_info_pretraining(model, eng, pending.result() if hasattr(pending, "result") else pending, w, tau, observations_count)

    def _loss_info_pretraining(self, model, eng, li, w, tau, observations_count):
        self._check_saturation(eng, li)
        loss_info = {"loss_component_observations_rec": w["rec"] * li["rec"], "loss_component_states_rec": w["states"] * li["states"],
                     "loss_component_perceptual_loss": li.get("perceptual_term", 0.0), "avg_perceptual_loss": li.get("perceptual", 0.0),
                     "loss_component_hidden_states_rec": w["hidden"] * li["hidden"], "loss_component_entropy": w["entropy"] * li["entropy"],
                     "loss_component_action_directions_kl_divergence": w["dir_kl"] * li["dir_kl"],
                     "loss_component_action_mutual_information": w["mi"] * li["mi"], "loss_component_action_state_distribution_kl": w["state_kl"] * li["state_kl"],
                     "avg_observations_rec_loss": li["rec"], "states_rec_loss": li["states"], "hidden_states_rec_loss": li["hidden"], "entropy_loss": li["entropy"],
                     "action_directions_kl_loss": li["dir_kl"], "action_mutual_information_loss": li["mi"], "action_state_distribution_kl_loss": li["state_kl"],
                     "gumbel_temperature": tau, "observations_count": observations_count}
        loss_info.update({k: v for k, v in li.items() if k.startswith("perceptual_loss_r")})
        loss_info.update(self.diagnostics(model, eng, pretraining=True, li=li))
        return li["total"], loss_info, {}

    def optimizer_step(self, model, world_size: int = 1):
        """optimizer.zero_grad(); loss.backward(); optimizer.step(); lr_scheduler.step() (trainer.py:584-587): backward already
        ran inside compute_losses.  Under torch.distributed (one process per GPU) the gradients of all ranks are summed here."""
        eng = model.module.last_engine
        if getattr(eng, "_dp_active", False) and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            eng.allreduce_gradients()            # R / D buckets were started during the backward; the rest + the wait happen here
            world_size = torch.distributed.get_world_size()
        if self.adam_m is None:
            self.adam_m, self.adam_v = torch.zeros_like(eng.grads), torch.zeros_like(eng.grads)
        self._to_engine_device(eng)
        eng.adam_m, eng.adam_v = self.adam_m, self.adam_v
        lr = self._get_current_lr()          # optimizer.step() runs BEFORE lr_scheduler.step() (trainer.py:586-587): step m+1 is the first at the decayed rate
        self.opt_steps += 1
        pre = bool(getattr(eng, "last_pretraining", False))
        member = getattr(model.module, "last_member", 0)
        if self.zero_grad_semantics == "zero_fill":
            # torch < 2.0: everything that has EVER had a gradient is stepped (with g = 0 when the last backward did not reach it) and counts the step
            if pre or self.s2h_steps > 0:
                self.s2h_steps += 1
            for k in range(len(self.member_steps)):
                if k == member or self.member_steps[k] > 0:
                    self.member_steps[k] += 1
            eng.adam_step(self.opt_steps, lr=lr, weight_decay=self.weight_decay, grad_scale=1.0 / world_size,
                          member_steps=self.member_steps if len(self.member_steps) > 1 else None, s2h_step=self.s2h_steps)
            return
        if pre:
            self.s2h_steps += 1
        self.member_steps[member] += 1
        extra = {"member_step": self.member_steps[member]} if len(self.member_steps) > 1 else {}
        eng.adam_step(self.opt_steps, lr=lr, weight_decay=self.weight_decay, grad_scale=1.0 / world_size, **extra)

    def _to_engine_device(self, eng):
        """optimiser / MI state loaded from a checkpoint before model.cuda() lives on the wrong device: the kernels take raw pointers"""
        dev = eng.grads.device
        if self.adam_m is not None and self.adam_m.device != dev:
            self.adam_m, self.adam_v = self.adam_m.to(dev), self.adam_v.to(dev)
        if self.mi_ema is not None and self.mi_ema.device != dev:
            self.mi_ema = self.mi_ema.to(dev)

    def train_epoch(self, model, dataloader=None):
        """training/trainer.py:552-609 without the wandb plumbing; `dataloader` yields Batch objects / batch tuples."""
        observations_count = self.get_observations_count()
        if hasattr(self.dataset, "set_observations_count"):
            self.dataset.set_observations_count(observations_count)
        performed = 0
        pending_log = None
        if dataloader is None:
            dataloader = self.dataloader if self.dataloader is not None else self.dataset
        for batch in dataloader:
            if performed > self.config["training"].get("max_steps_per_epoch", 10000):
                break
            self.global_step += 1
            performed += 1
            if self.get_observations_count() != observations_count:
                break
            # the loss values of a step are only logged: they come back through an asynchronous copy and are read AFTER the next step has been enqueued, so the GPU never
            # waits for the host between steps (the reference's .item() calls, trainer.py:503-530, stall it every step)
            if self.global_step <= self.config["training"].get("pretraining_steps", 0):
                finish = self._compute_losses_pretraining(model, batch, observations_count, deferred=True)
            else:
                finish = self._compute_losses(model, batch, observations_count, deferred=True)
            self.optimizer_step(model)
            if pending_log is not None:
                self._log_step(*pending_log)
            pending_log = (finish, self.global_step, self._get_current_lr())
        if pending_log is not None:
            self._log_step(*pending_log)
        return performed

    def _log_step(self, finish, step, lr):
        loss, loss_info, _ = finish()
        loss_info["loss"] = loss
        self.last_loss_info = loss_info
        if self.logger is not None:
            self.logger.print(f"step: {step} " + " ".join(f"{k}:{v:.3f}" for k, v in loss_info.items()) + f" lr: {lr:.4f}")

    # ---- checkpoints in the reference's format (training/trainer.py:80-122, smooth_mi_trainer.py:23-68): "model" (state_dict), "optimizer"
    # (torch.optim.Adam.state_dict layout), "lr_scheduler" (MultiStepLR.state_dict layout), "mi_estimator", "step" -- a checkpoint written
    # by the reference resumes here and vice versa ----
    @staticmethod
    def _optimizer_params(model):
        """Adam was built on model.parameters(): every nn.Parameter in registration order, trainable or not (kind 0 / 2 of the table)"""
        out = []
        for name, off, shape, kind in model.module.reference_order():
            if kind in (0, 2):
                n = 1
                for s_ in shape:
                    n *= s_
                out.append((name, off, n, tuple(shape), kind))
        return out

    def _export_optimizer(self, model):
        import collections
        params = self._optimizer_params(model)
        state = {}
        if self.adam_m is not None and self.opt_steps > 0:
            m, v = self.adam_m.cpu(), self.adam_v.cpu()
            for i, (name, off, n, shape, kind) in enumerate(params):
                if kind == 0:
                    steps = self._param_steps(name)
                    if steps == 0:
                        continue                      # (torch creates a parameter's state at its first step: an ensemble member that was never drawn has none)
                    state[i] = {"step": torch.tensor(float(steps)), "exp_avg": m[off:off + n].view(shape).clone(),
                                "exp_avg_sq": v[off:off + n].view(shape).clone()}
        group = {"lr": self._get_current_lr(), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": self.weight_decay, "amsgrad": False,
                 "initial_lr": self.lr, "params": list(range(len(params)))}
        sched = {"milestones": collections.Counter(self.lr_schedule), "gamma": self.lr_gamma, "base_lrs": [self.lr], "last_epoch": self.opt_steps,
                 "_step_count": self.opt_steps + 1, "_get_lr_called_within_step": False, "_last_lr": [self._get_current_lr()]}
        return {"state": state, "param_groups": [group]}, sched

    def _param_steps(self, name):
        """Adam step count of a parameter: the optimiser's, or its ensemble member's own (action_network.{m}.*)"""
        if len(self.member_steps) > 1 and name.startswith("action_network."):
            return self.member_steps[int(name.split(".")[1])]
        if name.startswith("state_to_hidden_state_layer."):
            return self.s2h_steps
        return self.opt_steps

    def _import_optimizer(self, model, opt, sched):
        dev = model.module._flat.device
        n_train = model.module.n_train
        if opt.get("fused_adam"):                                     # round-1 checkpoints of this mirror
            if opt.get("exp_avg") is not None:
                self.adam_m, self.adam_v, self.opt_steps = opt["exp_avg"].to(dev), opt["exp_avg_sq"].to(dev), opt["steps"]
            return
        params = self._optimizer_params(model)
        ids = opt["param_groups"][0]["params"]
        if len(ids) != len(params):
            raise Exception(f"optimizer state has {len(ids)} parameters, the model has {len(params)}")
        m, v = torch.zeros(n_train), torch.zeros(n_train)
        steps = 0
        for pid, (name, off, n, shape, kind) in zip(ids, params):
            st = opt["state"].get(pid)
            if st is None or kind != 0:
                continue
            m[off:off + n] = st["exp_avg"].reshape(-1).float()
            v[off:off + n] = st["exp_avg_sq"].reshape(-1).float()
            if len(self.member_steps) > 1 and name.startswith("action_network."):
                self.member_steps[int(name.split(".")[1])] = int(float(st["step"]))
                continue
            if name.startswith("state_to_hidden_state_layer."):
                self.s2h_steps = int(float(st["step"]))
                continue
            steps = max(steps, int(float(st["step"])))
        self.adam_m, self.adam_v, self.opt_steps = m.to(dev), v.to(dev), steps
        if sched and "last_epoch" in sched:
            self.opt_steps = max(self.opt_steps, int(sched["last_epoch"])) if steps == 0 else steps

    def save_checkpoint(self, model, name=None):
        root = self.config["logging"]["save_root_directory"]
        filename = os.path.join(root, "latest.pth.tar" if name is None else f"{name}_.pth.tar")
        sd = {k: v.detach().cpu().clone() for k, v in model.module.state_dict().items()}
        opt, sched = self._export_optimizer(model)
        state = {"model": sd, "optimizer": opt, "lr_scheduler": sched, "step": self.global_step}
        if self.SMOOTH_MI:
            k = self.config["data"]["actions_count"]
            ema = self.mi_ema.cpu() if self.mi_ema is not None else torch.full((k, k), 1.0 / (k * k))      # FixedMatrixEstimator's initial value
            state["mi_estimator"] = {"matrix_estimator.estimated_matrix": ema}
        torch.save(state, filename)

    def load_checkpoint(self, model, name=None):
        root = self.config["logging"]["save_root_directory"]
        filename = os.path.join(root, "latest.pth.tar" if name is None else f"{name}.pth.tar")
        if not os.path.isfile(filename):
            raise Exception(f"Cannot load model: no checkpoint found at '{filename}'")
        st = torch.load(filename, map_location="cpu", weights_only=False)
        model.module.load_state_dict(st["model"])
        self._import_optimizer(model, st.get("optimizer", {}) or {}, st.get("lr_scheduler", {}))
        mi = (st.get("mi_estimator") or {}).get("matrix_estimator.estimated_matrix")
        if mi is not None:
            self.mi_ema = mi.to(model.module._flat.device).float()
        self.global_step = st["step"]


def trainer(config, model, dataset, logger):
    return Trainer(config, model, dataset, logger)
