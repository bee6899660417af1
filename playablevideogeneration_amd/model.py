"""Drop-in for the reference's model plugin: `config["model"]["architecture"] = "playablevideogeneration_amd.model"`
(main) or `"playablevideogeneration_amd.reduced_model"`; the factory `model(config)` returns an nn.Module with the surface the
reference callers use (model/main_model/model.py:57-82, 561-607):

    model(batch_tuple, ground_truth_observations_init, gumbel_temperature=...) -> 20-tuple
    model.module.centroid_estimator.get_estimated_centroids(), state_dict()/load_state_dict() with the reference's keys,
    parameters() (views of ONE flat fp32 buffer), train()/eval(), cuda(), start_inference(), generate_next(obs, action).

All arithmetic runs in libcaddy_hip.so (HIP, gfx950) through `Engine`; there is no torch/CPU fallback.  Noise is drawn from
torch's global CPU generator with the reference's calls in the reference's order, so `torch.manual_seed(s)` reproduces the
reference's action samples bit-for-bit.
"""
import ctypes as C
import random
import warnings
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib
from .engine import CaddyConfig, Engine, ParamInfo, _bind
from .init import init_parameters


class _Node(nn.Module):
    """Anonymous container used to reproduce the reference's dotted state_dict names."""


class _CentroidView:
    def __init__(self, model):
        self._m = model

    def get_estimated_centroids(self) -> torch.Tensor:     # centroid_estimator.py:31-36
        return self._m._param_views["centroid_estimator.estimated_centroids"]

    @property
    def estimated_centroids(self):
        return self.get_estimated_centroids()


class Model(nn.Module):
    VARIANT = "main"

    def __init__(self, config, lib=None):
        super().__init__()
        self.config = config
        an = config["model"]["action_network"]
        if config["training"].get("use_ground_truth_actions", False):
            self._forbid_gt_actions = True
        else:
            self._forbid_gt_actions = False
        self._pretraining_detach = config["training"].get("pretraining_detach", False)
        # training.deterministic (an addition of this plugin, default TRUE since round 5): bit-reproducible backward pass (caddy_set_deterministic) -- two runs from the same
        # seeds and checkpoint give bit-identical parameters, as the reference's CPU path does; false selects fp32 atomics in arrival order (< 1 % faster)
        self._deterministic = bool(config["training"].get("deterministic", True))
        sr = config["model"]["representation_network"]["state_resolution"]
        self.dims = dict(variant=self.VARIANT, height=int(sr[0]) * 8, width=int(sr[1]) * 8,
                         stacking=config["training"]["batching"]["observation_stacking"], actions=config["data"]["actions_count"],
                         action_dim=an["action_space_dimension"], hidden=config["model"]["dynamics_network"]["hidden_state_size"],
                         use_gumbel=bool(an["use_gumbel"]), hard_gumbel=bool(an["hard_gumbel"]), use_variations=bool(an.get("use_variations", True)),
                         centroid_alpha=config["model"]["centroid_estimator"]["alpha"], ensemble=int(an.get("ensamble_size", 1)))
        if not 1 <= self.dims["ensemble"] <= 8:
            raise Exception("model.action_network.ensamble_size must be between 1 and 8")
        self.last_member = 0           # ensemble member of the last forward pass (model.py:152: random.choice(self.action_network))
        self.random_noise_size = config["model"]["dynamics_network"]["random_noise_size"]
        self.current_temperature = an["gumbel_temperature"]      # GumbelSoftmax.current_temperature (gumbel_softmax.py:21)
        self._lib = _bind(lib if lib is not None else _lib.load())
        d = self.dims
        cc = CaddyConfig(0 if d["variant"] == "main" else 1, 1, 2, d["height"], d["width"], d["stacking"], d["actions"], d["action_dim"], d["hidden"],
                         int(d["use_gumbel"]), int(d["hard_gumbel"]), int(d["use_variations"]), d["centroid_alpha"], 0, d["ensemble"])
        n = self._lib.caddy_param_floats(C.byref(cc))
        if n <= 0:
            raise Exception(self._lib.caddy_last_error().decode())
        self.n_floats, self.n_train = n, self._lib.caddy_trainable_floats(C.byref(cc))
        info = ParamInfo()
        self.table = []
        for i in range(self._lib.caddy_param_count(C.byref(cc))):
            self._lib.caddy_param_info_get(C.byref(cc), i, C.byref(info))
            self.table.append((info.name.decode(), info.offset, tuple(info.shape[:info.ndim]), info.kind))
        self._flat = torch.zeros(n, dtype=torch.float32)
        self._flat_grad = torch.zeros(self.n_train, dtype=torch.float32)
        self._param_views: Dict[str, torch.Tensor] = {}
        self._bn_counters: Dict[str, torch.Tensor] = {}
        self._register_tree()
        init_parameters(self, seed=int(torch.randint(0, 2 ** 31 - 1, (1,)).item()))
        self._engines: Dict[Tuple[int, int], Engine] = {}
        self._vgg_state = None           # VGG19 weights of the perceptual loss (set by the trainer: enable_perceptual)
        self._infer: Optional[Engine] = None
        self._bn_seen: Dict[Tuple[int, int], Dict[str, int]] = {}
        self.last_engine: Optional[Engine] = None
        self.centroid_estimator_view = _CentroidView(self)
        self._rebind()          # Parameter.grad = views of the flat gradient buffer

    # ---- parameter plumbing ---------------------------------------------------------------------------------------
    def view(self, entry) -> torch.Tensor:
        name, off, shape, kind = entry
        k = 1
        for s in shape:
            k *= s
        return self._flat[off:off + k].view(shape)

    def reference_order(self):
        """Table entries in the reference's registration order (what state_dict() / parameters() / a positional torch optimizer state see).  The
        C driver lists a ConvLSTM cell as 4 gate weights then 4 gate biases (one packed convolution); the reference module registers
        input / forget / output / cell gate as (weight, bias) pairs (convolutional_lstm_cell.py:26-45)."""
        gates = {"input_gate": 0, "forget_gate": 1, "output_gate": 2, "cell_gate": 3}

        def key(ie):
            i, e = ie
            parts = e[0].split(".")
            if len(parts) >= 3 and parts[-3] == "cell" and parts[-2] in gates:
                return (self._cell_anchor[".".join(parts[:-2])], gates[parts[-2]] * 2 + (0 if parts[-1] == "weight" else 1))
            return (i, 0)
        self._cell_anchor = {}
        for i, e in enumerate(self.table):
            parts = e[0].split(".")
            if len(parts) >= 3 and parts[-3] == "cell" and parts[-2] in gates:
                self._cell_anchor.setdefault(".".join(parts[:-2]), i)
        return [e for _, e in sorted(enumerate(self.table), key=key)]

    def _register_tree(self):
        for e in self.reference_order():
            name, off, shape, kind = e
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                if p not in node._modules:
                    node.add_module(p, _Node())
                node = node._modules[p]
            v = self.view(e)
            if kind == 1:
                node.register_buffer(parts[-1], v)
                if parts[-1] == "running_var":
                    cnt = torch.zeros((), dtype=torch.int64)
                    node.register_buffer("num_batches_tracked", cnt)
                    self._bn_counters[".".join(parts[:-1])] = cnt
            else:
                node.register_parameter(parts[-1], nn.Parameter(v, requires_grad=(kind == 0)))
            self._param_views[name] = v

    def _rebind(self):
        """Re-point every Parameter / buffer at the (moved) flat buffers; gradients are views of the flat gradient buffer."""
        for e in self.table:
            name, off, shape, kind = e
            parts = name.split(".")
            node = self
            for p in parts[:-1]:
                node = node._modules[p]
            v = self.view(e)
            self._param_views[name] = v
            if kind == 1:
                node._buffers[parts[-1]] = v
            else:
                prm = node._parameters[parts[-1]]
                prm.data = v
                if kind == 0:
                    k = v.numel()
                    prm.grad = self._flat_grad[off:off + k].view(shape)
        for pre, cnt in list(self._bn_counters.items()):
            node = self
            for p in pre.split("."):
                node = node._modules[p]
            self._bn_counters[pre] = node._buffers["num_batches_tracked"]

    def _apply(self, fn, recurse=True):
        new_flat = fn(self._flat)
        if new_flat.dtype != torch.float32:
            raise Exception("the HIP path computes in fp32; dtype conversion of the model is not supported")
        self._flat = new_flat.contiguous()
        self._flat_grad = fn(self._flat_grad).contiguous()
        for pre in list(self._bn_counters):
            node = self
            for p in pre.split("."):
                node = node._modules[p]
            node._buffers["num_batches_tracked"] = fn(node._buffers["num_batches_tracked"])
        self._rebind()
        self._engines.clear()
        self._infer = None
        return self

    def zero_grad(self, set_to_none: bool = False):
        self._flat_grad.zero_()

    @property
    def module(self):            # reference callers unwrap nn.DataParallel via `.module` (training/trainer.py:434)
        return self

    @property
    def centroid_estimator(self):
        return self.centroid_estimator_view

    def state_dict(self, *a, **k):
        sd = super().state_dict(*a, **k)
        sd.pop("centroid_estimator_view", None)
        return sd

    # ---- engines ------------------------------------------------------------------------------------------------------
    MAX_ENGINES = 2      # every Engine owns a workspace sized for BPTT (tens of GiB at BAIR): keep the current training shape + one evaluation shape

    def enable_perceptual(self, vgg_state_dict):
        """VGG19 weights for the perceptual loss (training/losses.py:379-491); engines created from now on carry the VGG19 buffers."""
        self._vgg_state = vgg_state_dict
        self._engines.clear()

    def engine(self, B: int, T: int) -> Engine:
        key = (B, T)
        if key not in self._engines:
            d = self.dims
            if self._flat.device.type != getattr(self._lib, "_caddy_device_type", "cuda"):
                raise Exception("playablevideogeneration_amd runs on the MI355X only: call model.cuda() first (no CPU fallback)")
            # the sequence-length curriculum (trainer.py:152-165) and the evaluators keep asking for new (B, T): evict the least recently used
            while len(self._engines) >= self.MAX_ENGINES:
                old = next(iter(self._engines))
                self._engines.pop(old)
                self._bn_seen.pop(old, None)
            use_vgg = self._vgg_state is not None and d["height"] >= 64 and d["width"] >= 64
            eng = Engine(batch=B, seq_len=T, device=self._flat.device, lib=self._lib, params=self._flat, grads=self._flat_grad, perceptual=use_vgg, **d)
            if use_vgg:
                eng.load_vgg(self._vgg_state)
            eng.set_deterministic(self._deterministic)
            self._engines[key] = eng
            self._bn_seen[key] = {}
        else:
            self._engines[key] = self._engines.pop(key)      # most recently used last
        return self._engines[key]

    def _sync_bn_counters(self, eng: Engine, key):
        seen = self._bn_seen.setdefault(key, {})
        for name, calls in eng.bn_calls().items():
            delta = calls - seen.get(name, 0)
            if delta:
                self._bn_counters[name] += delta
                seen[name] = calls

    # ---- Model.forward (model/main_model/model.py:57-82) -----------------------------------------------------------------
    def forward(self, batch_tuple, ground_truth_observations_init=0, pretraining=False, gumbel_temperature=None,
                action_sampler=None, action_variation_sampler=None, fetch_outputs=True):
        if pretraining and self._pretraining_detach:
            raise Exception("Pretraining detach is not supported by the current model")
        if not pretraining and ground_truth_observations_init <= 0:
            raise Exception("To forward the full model specify a number of ground truth observations > 0")
        if self._forbid_gt_actions:
            raise Exception("The use of ground truth actions during training is not supported by the selected model")
        observations = batch_tuple[0]
        B, T = observations.shape[:2]
        eng = self.engine(B, T)
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1 \
                and not getattr(eng, "_dp_active", False):
            eng.enable_data_parallel()           # one process per GPU replaces nn.DataParallel (train.py:40): hooks for the global-batch reductions
        gt_actions = None
        if action_sampler is not None:                       # model.py:172-173: actions[:, :-1].reshape((-1,))
            gt_actions = batch_tuple[1][:, :-1].reshape((-1,)).to(self._flat.device)
        eng.set_samplers(action_sampler, action_variation_sampler, gt_actions)
        # model.py:152 / :358: `random.choice(self.action_network)` -- one draw from Python's global `random` state per forward pass, also for an ensemble of one.  Under
        # torch.distributed the reference's single process drew once for all replicas: rank 0's draw is broadcast (the members that were not drawn are not stepped by Adam)
        member = random.choice(range(d_ens := self.dims["ensemble"]))
        if d_ens > 1 and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            mt = torch.tensor([member], dtype=torch.int64, device=self._flat.device if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.broadcast(mt, 0)
            member = int(mt.item())
        if d_ens > 1:
            eng.set_action_member(member)
        self.last_member = member
        if gumbel_temperature is not None:
            self.current_temperature = gumbel_temperature
        d = self.dims
        Da, K, n = d["action_dim"], d["actions"], T - 1
        # RNG draws in the reference's order; all independent of data, so they can be drawn up front
        noise = {"eps_states": torch.randn((B * T, Da), dtype=torch.float32), "eps_dirs": torch.randn((B, n, Da), dtype=torch.float32).reshape(B * n, Da)}
        draw_gumbel = d["use_gumbel"] and action_sampler is None      # model.py:171-176: an explicit sampler skips the Gumbel draw
        noise["gumbel_uniform"] = torch.rand((B * n, K)) if draw_gumbel else torch.full((B * n, K), 0.5)
        for _ in range(n):
            torch.randn((B, self.random_noise_size))        # model.py:220/496: drawn, never consumed by R
        noise["eps_states_rec"] = torch.randn((B * T, Da), dtype=torch.float32)
        noise["eps_dirs_rec"] = torch.randn((B, n, Da), dtype=torch.float32).reshape(B * n, Da)
        if pretraining:
            out = eng.forward_pretraining(observations, float(self.current_temperature), noise, training=self.training, fetch_outputs=fetch_outputs)
        else:
            out = eng.forward_full(observations, int(ground_truth_observations_init), float(self.current_temperature), noise,
                                   training=self.training, fetch_outputs=fetch_outputs)
        if self.training:
            self._sync_bn_counters(eng, (B, T))
        else:
            # (training passes report through the loss call: trainer._check_saturation.)  The poll synchronises both streams and copies the flag words to the host: every 16th
            # evaluation pass of an engine, like every 64th roll-out frame -- the flags are sticky on the device, nothing is lost in between
            eng._eval_passes = getattr(eng, "_eval_passes", 0) + 1
            if eng._eval_passes % 16 == 1:
                self._check_numerics(eng)
        self.last_engine = eng
        return tuple(out) if out is not None else None

    @staticmethod
    def _check_numerics(eng):
        """f16 range guards of the split-f16 forward on passes without a loss call (evaluation, roll-out): a clamped activation is reported once and the layers that met it move to a
        forward without a range limit; a NaN -- which the clamp would have hidden -- raises, as the reference's fp32 arithmetic would have produced NaN outputs"""
        bits = eng.numerics_flags()
        if bits & 2:      # (the reference's fp32 arithmetic would have returned NaN outputs, not raised: report loudly, let the caller decide)
            warnings.warn("NaN activation in a forward pass since the last check (found by the f16 range guard of the split-f16 convolutions, which clamps it): the outputs of "
                          "that pass are finite where the reference's would be NaN")
        if bits & 1:
            warnings.warn(f"a forward activation exceeded the f16 range (|x| > 65504) and was clamped in the pass that just ran; {eng.fallback_layers()} convolution "
                          "layer(s) now run without a range limit -- repeat the pass for unclamped results")

    # ---- play.py path (model.py:561-607) ---------------------------------------------------------------------------------
    def start_inference(self):
        if self._infer is None:
            d = self.dims
            self._infer = Engine(batch=1, seq_len=2, device=self._flat.device, lib=self._lib, params=self._flat, grads=self._flat_grad, **d)
        self._infer.start_inference()

    def generate_next(self, observation: torch.Tensor, action: int, noise=False):
        if self._infer is None:
            raise Exception("start_inference() must be called before generate_next()")
        variation = torch.randn((1, self.dims["action_dim"]), dtype=torch.float32)[0] if noise else None
        torch.randn((1, self.random_noise_size))             # generate_noise(batch_size=1), unused by R (model.py:596)
        out = self._infer.generate_next(observation, action, variation)
        self._frames_since_poll = getattr(self, "_frames_since_poll", 0) + 1
        if self._frames_since_poll >= 64:                    # (a poll waits for the stream: not every frame)
            self._frames_since_poll = 0
            self._check_numerics(self._infer)
        return out

    def generate_next_interpolation(self, observation: torch.Tensor, first_action: int, second_action: int, interpolation_factor: float):
        """model.py:609-655: act with the centroid nearer to the interpolated point, the offset to it as the action variation."""
        if self._infer is None:
            raise Exception("start_inference() must be called before generate_next_interpolation()")
        cen = self.centroid_estimator.get_estimated_centroids()
        selected = second_action if interpolation_factor > 0.5 else first_action
        point = (cen[second_action] - cen[first_action]) * interpolation_factor + cen[first_action]
        torch.randn((1, self.random_noise_size))             # generate_noise(batch_size=1), unused by R
        return self._infer.generate_next(observation, selected, (point - cen[selected]).detach())


def model(config):
    return Model(config)
