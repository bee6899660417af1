"""Thin Python driver over the C-ABI (include/caddy_hip.h): owns the flat parameter / gradient / workspace buffers
(torch tensors = device memory only) and turns raw output buffers into the reference's 20-tuple.

`Engine` mirrors what reference callers do with the model object: forward in full-model mode,
loss + backward, optimiser step, start_inference / generate_next.  `lib` may be injected by the tests (host simulator
build of the same kernel sources); the default is the gfx950 library and there is no CPU fallback.
"""
import ctypes as C
import os
from typing import Dict, List, Optional

import torch

from . import _lib

LOSS_NAMES = ["total", "rec", "states", "entropy", "dir_kl", "mi", "state_kl", "hidden", "l1_r0", "l1_r1", "l1_r2", "perceptual", "perceptual_term"]
LOSS_SLOTS, LOSS_PERC_R0, DIAG_0 = 56, 16, 40          # include/caddy_hip.h: CADDY_LOSS_SLOTS, CADDY_LOSS_PERC_R0, CADDY_DIAG_0
DIAG_NAMES = ["samples_entropy", "action_distribution_entropy", "states_magnitude", "hidden_states_magnitude", "action_directions_mean_magnitude",
              "action_directions_variance_magnitude", "reconstructed_action_directions_mean_magnitude", "reconstructed_action_directions_variance_magnitude",
              "action_directions_reconstruction_error", "reconstructed_action_directions_kl_loss", "centroids_mean_magnitude", "average_centroids_distance",
              "average_action_variations_norm_l2", "action_variations_mean"]      # logging-only entries of the reference's loss_info (trainer.py:475-491)


class CaddyConfig(C.Structure):
    _fields_ = [("variant", C.c_int), ("batch", C.c_int), ("seq_len", C.c_int), ("height", C.c_int), ("width", C.c_int),
                ("stacking", C.c_int), ("actions", C.c_int), ("action_dim", C.c_int), ("hidden", C.c_int),
                ("use_gumbel", C.c_int), ("hard_gumbel", C.c_int), ("use_variations", C.c_int), ("centroid_alpha", C.c_float),
                ("perceptual", C.c_int), ("ensemble", C.c_int)]


class ParamInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("offset", C.c_long), ("ndim", C.c_int), ("shape", C.c_int * 4), ("kind", C.c_int)]


class Noise(C.Structure):
    _fields_ = [("eps_states", C.c_void_p), ("eps_dirs", C.c_void_p), ("gumbel_uniform", C.c_void_p),
                ("eps_states_rec", C.c_void_p), ("eps_dirs_rec", C.c_void_p)]


class LossCfg(C.Structure):
    _fields_ = [("rec", C.c_double), ("states", C.c_double), ("entropy", C.c_double), ("dir_kl", C.c_double), ("mi", C.c_double),
                ("state_kl", C.c_double), ("hidden", C.c_double), ("mi_entropy_lambda", C.c_double),
                ("mi_ema", C.c_void_p), ("mi_ema_alpha", C.c_float), ("update_mi_ema", C.c_int),
                ("perceptual", C.c_double), ("perceptual_log", C.c_int), ("diagnostics", C.c_int), ("no_sync", C.c_int)]


def _losses_dict(host, diagnostics: bool, with_perceptual: bool) -> Dict[str, float]:
    """losses_host slots of caddy_loss_backward -> the names the trainer mirror uses"""
    res = {n: float(host[i]) for i, n in enumerate(LOSS_NAMES)}
    res["f16_saturated"] = bool(host[13])      # CADDY_LOSS_F16_SATURATED: a split-f16 forward convolution clamped an input beyond the f16 range (see Engine.f16_saturated)
    if diagnostics:      # evaluated on the device inside the same call: no tensor fetch, no extra synchronisation
        res["diagnostics"] = {n: float(host[DIAG_0 + i]) for i, n in enumerate(DIAG_NAMES)}
    if with_perceptual:      # loss_info keys of trainer.py:459-462
        for r in range(3):
            res[f"perceptual_loss_r{r}"] = float(host[LOSS_PERC_R0 + 6 * r])
            for l in range(5):
                res[f"perceptual_loss_r{r}_l{l}"] = float(host[LOSS_PERC_R0 + 6 * r + 1 + l])
    return res


class PendingLosses:
    """loss values of a caddy_loss_backward call issued with caddy_loss_cfg.no_sync: on their way into a pinned host buffer"""

    def __init__(self, buf, event, diagnostics, with_perceptual):
        self._buf, self._event, self._diag, self._perc, self._res = buf, event, diagnostics, with_perceptual, None

    def result(self) -> Dict[str, float]:
        if self._res is None:
            self._event.synchronize()
            self._res = _losses_dict(self._buf.tolist(), self._diag, self._perc)
            self._buf = None
        return self._res


def _bind(lib):
    if getattr(lib, "_caddy_bound", False):
        return lib
    lib.caddy_last_error.restype = C.c_char_p
    lib.caddy_param_floats.restype = C.c_long
    lib.caddy_trainable_floats.restype = C.c_long
    lib.caddy_workspace_bytes.restype = C.c_size_t
    lib.caddy_ctx_create.restype = C.c_void_p
    lib.caddy_ctx_create.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
    lib.caddy_ctx_destroy.argtypes = [C.c_void_p]
    lib.caddy_set_stream.argtypes = [C.c_void_p, C.c_void_p]
    lib.caddy_set_grads_ready_hook.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.caddy_set_sampler_hook.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
    lib.caddy_set_allreduce_hook.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.caddy_forward_full.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.caddy_forward_pretraining.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.caddy_get_output.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.caddy_get_output_grad.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.caddy_loss_backward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.caddy_adam_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_float]
    lib.caddy_adam_step_member.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_float]
    lib.caddy_adam_step_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(C.c_int), C.c_int, C.c_float]
    lib.caddy_set_action_member.argtypes = [C.c_void_p, C.c_int]
    lib.caddy_vgg_param_floats.restype = C.c_long
    lib.caddy_vgg_param_info_get.argtypes = [C.c_int, C.c_void_p]
    lib.caddy_load_vgg.argtypes = [C.c_void_p, C.c_void_p]
    lib.caddy_set_vgg_precision.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.caddy_set_perceptual_prefetch.argtypes = [C.c_void_p, C.c_int]
    lib.caddy_dp_unique_id.argtypes = [C.c_char_p]
    lib.caddy_dp_init.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_int, C.c_int]
    lib.caddy_allreduce_grads.argtypes = [C.c_void_p]
    lib.caddy_dp_bucket_floats.argtypes = [C.c_void_p]
    lib.caddy_dp_bucket_floats.restype = C.c_long
    lib.caddy_dp_shutdown.argtypes = [C.c_void_p]
    lib.caddy_set_rollout_fold.argtypes = [C.c_void_p, C.c_int]
    lib.caddy_set_precision.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.caddy_start_inference.argtypes = [C.c_void_p]
    lib.caddy_f16_saturated.argtypes = [C.c_void_p]
    lib.caddy_fallback_layers.argtypes = [C.c_void_p]
    lib.caddy_perceptual_per_frame.argtypes = [C.c_void_p, C.c_void_p]
    lib.caddy_sequence_losses_per_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    lib.caddy_set_deterministic.argtypes = [C.c_void_p, C.c_int]
    lib.caddy_generate_next.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.caddy_bn_layer_count.argtypes = [C.c_void_p]
    lib.caddy_bn_calls.argtypes = [C.c_void_p, C.c_int, C.c_char_p]
    lib.caddy_bn_calls.restype = C.c_long
    lib._caddy_bound = True
    return lib


class CaddyError(Exception):
    pass


class Engine:
    def __init__(self, *, variant: str, batch: int, seq_len: int, height: int, width: int, stacking: int, actions: int,
                 action_dim: int, hidden: int, use_gumbel=True, hard_gumbel=False, use_variations=True, centroid_alpha=0.1,
                 device="cuda", lib=None, params=None, grads=None, perceptual=False, ensemble=1):
        self.lib = _bind(lib if lib is not None else _lib.load())
        self.device = torch.device(device)
        self.cfg = CaddyConfig(0 if variant == "main" else 1, batch, seq_len, height, width, stacking, actions, action_dim, hidden,
                               int(use_gumbel), int(hard_gumbel), int(use_variations), centroid_alpha, int(perceptual), int(ensemble))
        self.perceptual, self.vgg_loaded, self.ensemble = bool(perceptual), False, int(ensemble)
        self.B, self.T, self.H, self.W, self.S, self.K, self.Da, self.Ch = batch, seq_len, height, width, stacking, actions, action_dim, hidden
        n = self.lib.caddy_param_floats(C.byref(self.cfg))
        if n <= 0:
            raise CaddyError(self._err())
        self.n_floats, self.n_train = n, self.lib.caddy_trainable_floats(C.byref(self.cfg))
        # flat buffers may be supplied by the caller (Model shares one parameter buffer across engines of different B/T)
        self.params = params if params is not None else torch.zeros(n, dtype=torch.float32, device=self.device)
        self.grads = grads if grads is not None else torch.zeros(self.n_train, dtype=torch.float32, device=self.device)
        assert self.params.numel() == n and self.grads.numel() == self.n_train and self.params.is_contiguous() and self.grads.is_contiguous()
        assert self.params.device == self.grads.device and self.params.device.type == self.device.type
        self.table = []
        info = ParamInfo()
        for i in range(self.lib.caddy_param_count(C.byref(self.cfg))):
            self.lib.caddy_param_info_get(C.byref(self.cfg), i, C.byref(info))
            self.table.append((info.name.decode(), info.offset, tuple(info.shape[:info.ndim]), info.kind))
        self.ws_bytes = self.lib.caddy_workspace_bytes(C.byref(self.cfg))
        raw = torch.empty(self.ws_bytes + 256, dtype=torch.uint8, device=self.device)
        off = (-raw.data_ptr()) % 256
        self._ws_raw, self._ws_ptr = raw, raw.data_ptr() + off
        self.ctx = self.lib.caddy_ctx_create(C.byref(self.cfg), self.params.data_ptr(), self.grads.data_ptr(), self._ws_ptr, self.ws_bytes)
        if not self.ctx:
            raise CaddyError(self._err())
        self.adam_m = self.adam_v = None
        self.mi_ema = None
        self._pinned_losses, self._pinned_next = None, 0      # two pinned host buffers of loss_backward(deferred=True)
        self._keep = []

    def set_samplers(self, action_sampler=None, action_variation_sampler=None, gt_actions: Optional[torch.Tensor] = None):
        """Evaluation samplers of the reference (model.py:171-190): `action_sampler(log_probs (n,K), gt_actions (n,)) -> (n,K)` and
        `action_variation_sampler(sampled_dirs (n,Da), samples (n,K)) -> (n,Da)`, called mid-forward on views of the workspace.
        Pass None for both to clear."""
        if action_sampler is None and action_variation_sampler is None:
            self._sampler_keepalive = None
            self._check(self.lib.caddy_set_sampler_hook(self.ctx, None, None, 0, 0))
            return
        base = self._ws_raw.data_ptr()

        def view(ptr, n, cols):
            off = ptr - base
            return self._ws_raw[off:off + 4 * n * cols].view(torch.float32).view(n, cols)

        def _hook(logp, dirs, samples, variations, n, K, Da, stage, _user):
            if stage == 0:
                view(samples, n, K).copy_(action_sampler(view(logp, n, K), gt_actions).to(torch.float32))
            else:
                view(variations, n, Da).copy_(action_variation_sampler(view(dirs, n, Da), view(samples, n, K)).to(torch.float32))

        self._sampler_keepalive = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)(_hook)
        self._check(self.lib.caddy_set_sampler_hook(self.ctx, C.cast(self._sampler_keepalive, C.c_void_p), None,
                                                    int(action_sampler is not None), int(action_variation_sampler is not None)))

    def enable_data_parallel(self, process_group=None, force=False, overlap=True, native=None):
        """Data parallelism, one process per GPU.  native (default on the GPU when the library finds an RCCL): the three reductions are issued from C straight
        into ncclAllReduce on a communicator the context owns (caddy_dp_init; the two 128-byte unique ids travel through torch.distributed once) -- no Python on the
        per-step path.  Otherwise torch.distributed hooks (gloo in the CPU tests, or RCCL through torch):

        Data parallelism over torch.distributed (RCCL on the MI355X, gloo in the CPU tests), one process per GPU.
        Registers (a) the all-reduce hook of the small global-batch reductions (centroid sums, MI joint matrix) and (b) with
        `overlap`, the gradient-bucket hook: the dynamics / rendering ranges of the flat gradient buffer (~91 % of the bytes) are
        all-reduced asynchronously as soon as the time loop's backward is done, behind the side HIP stream, while the backward of A
        and of E on the ground-truth frames is still running.  After `loss_backward` call `allreduce_gradients()`."""
        import torch.distributed as dist
        world = dist.get_world_size(process_group)
        self._dp_group = process_group
        self._dp_active = True
        self._dp_native = False
        self._early = []                     # (offset, count, work) of the buckets already in flight
        if world == 1 and not force:
            return
        if native is None:      # (CADDY_DP_NATIVE=0: the torch.distributed hooks even where RCCL is loadable; =force: the native path on any device type -- the simulator
                                # tests drive csrc/dp_rccl.cpp through a stand-in library named by CADDY_RCCL_LIB)
            mode = os.environ.get("CADDY_DP_NATIVE", "1")
            native = ((self.device.type == "cuda" or mode == "force") and mode != "0"
                      and hasattr(self.lib, "caddy_dp_available") and self.lib.caddy_dp_available() == 1)
        if world > 1:      # every rank must take the same branch BEFORE anyone enters the unique-id broadcast / ncclCommInitRank: agree on the weakest rank's capability
            cap = torch.tensor([int(bool(native))], device=self.device if dist.get_backend(process_group) == "nccl" else "cpu", dtype=torch.int32)
            dist.all_reduce(cap, op=dist.ReduceOp.MIN, group=process_group)
            native = bool(cap.item())
        if native:
            rank = dist.get_rank(process_group)
            buf = C.create_string_buffer(256)      # two ncclUniqueIds: one communicator per stream (dp_rccl.cpp)
            ok = 1
            if rank == 0:
                ok = int(self.lib.caddy_dp_unique_id(buf) == 0)
            box = [(buf.raw, ok) if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0) if process_group is not None else 0, group=process_group)
            uid, ok = box[0]
            rc = 0
            if ok:
                self._stream()
                rc = int(self.lib.caddy_dp_init(self.ctx, uid, world, rank, int(bool(overlap))))
                ok = int(rc == 0)
            if rc == -3:      # the library's own code for "ncclCommInitRank did not return in time" (dp_rccl.cpp): a rank never arrived, the others' state is unknown, nothing to fall back to
                raise RuntimeError(self._err())
            if world > 1:      # every rank must take the same path: one failed communicator sends all of them to the hook path
                flag = torch.tensor([ok], device=self.device, dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
                ok = int(flag.item())
            if ok:
                self._dp_native = True
                return
            import sys
            print(f"[playablevideogeneration_amd] native RCCL data parallelism unavailable ({self._err()}): using the torch.distributed hooks", file=sys.stderr)
            if hasattr(self.lib, "caddy_dp_shutdown"):
                self.lib.caddy_dp_shutdown(self.ctx)
        base = self._ws_raw.data_ptr()

        def _hook(ptr, count, _user):
            off = ptr - base
            dist.all_reduce(self._ws_raw[off:off + 4 * count].view(torch.float32), group=process_group)

        self._hook_keepalive = C.CFUNCTYPE(None, C.c_void_p, C.c_int, C.c_void_p)(_hook)
        self._check(self.lib.caddy_set_allreduce_hook(self.ctx, C.cast(self._hook_keepalive, C.c_void_p), None, world))
        if not overlap:
            return
        on_gpu = self.device.type == "cuda"

        self.hook_host_seconds, self.hook_calls = 0.0, 0      # host time spent inside the bucket callback (measured: the GIL / ctypes cost of this path)

        def _ready(_grads, offset, count, stream, _user):
            import time as _time
            t0 = _time.perf_counter()
            try:
                return _ready_inner(offset, count, stream)
            finally:
                self.hook_host_seconds += _time.perf_counter() - t0
                self.hook_calls += 1

        def _ready_inner(offset, count, stream):
            sl = self.grads[offset:offset + count]
            if on_gpu and stream:               # enqueue behind the stream on which this range becomes valid (the driver's side stream)
                with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=self.device)):
                    work = dist.all_reduce(sl, group=process_group, async_op=True)
            else:
                work = dist.all_reduce(sl, group=process_group, async_op=True)
            self._early.append((offset, count, work))

        self._ready_keepalive = C.CFUNCTYPE(None, C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_void_p)(_ready)
        self._check(self.lib.caddy_set_grads_ready_hook(self.ctx, C.cast(self._ready_keepalive, C.c_void_p), None))

    def allreduce_gradients(self):
        """Sum the flat gradient buffer over the ranks: waits for the buckets started during `loss_backward` (the current stream
        waits, not the host) and all-reduces what is left (E, A, state_to_hidden_state)."""
        if getattr(self, "_dp_native", False):
            self._stream()
            self._check(self.lib.caddy_allreduce_grads(self.ctx))
            return
        import torch.distributed as dist
        group = getattr(self, "_dp_group", None)
        early = sorted(getattr(self, "_early", []), key=lambda e: e[0])
        pos = 0
        for off, cnt, _ in early:
            if off > pos:
                dist.all_reduce(self.grads[pos:off], group=group)
            pos = max(pos, off + cnt)
        if pos < self.grads.numel():
            dist.all_reduce(self.grads[pos:], group=group)
        for _, _, work in early:
            work.wait()
        self._early = []

    # ---- VGG19 weights of the perceptual loss (replaces torchvision.models.vgg19(pretrained=True).features, model/layers/vgg.py:16) ----
    def vgg_table(self):
        info, t = ParamInfo(), []
        for i in range(self.lib.caddy_vgg_param_count()):
            self.lib.caddy_vgg_param_info_get(i, C.byref(info))
            t.append((info.name.decode(), info.offset, tuple(info.shape[:info.ndim])))
        return t

    def load_vgg(self, sd: Dict[str, torch.Tensor]):
        """`sd`: torchvision naming (`features.{idx}.weight` / `.bias`; a `vgg19().state_dict()` or its `.features` sub-dict without the
        prefix); only the 13 convolutions up to conv5_1 are read -- the slices of model/layers/vgg.py:25-34 never run the last three."""
        if not self.perceptual:
            raise CaddyError("Engine was created with perceptual=False")
        flat = torch.zeros(self.lib.caddy_vgg_param_floats(), dtype=torch.float32, device=self.device)
        for name, off, shape in self.vgg_table():
            key = name if name in sd else name[len("features."):]
            if key not in sd:
                raise CaddyError(f"VGG19 state dict lacks {name}")
            t = sd[key].detach().to(self.device, torch.float32)
            assert tuple(t.shape) == shape, (name, tuple(t.shape), shape)
            flat[off:off + t.numel()] = t.reshape(-1)
        self._stream()
        self._check(self.lib.caddy_load_vgg(self.ctx, flat.data_ptr()))
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()      # `flat` may be freed: the library does not reference it after the call
        self.vgg_loaded = True

    def set_vgg_precision(self, forward: int, dgrad: int):
        """arithmetic of the VGG19 convolutions: forward 0 (exact fp32) | 16 (split f16, default) | 18 (plain f16); dgrad 0 | 17 (split bf16, default) | 19"""
        self._check(self.lib.caddy_set_vgg_precision(self.ctx, int(forward), int(dgrad)))

    def set_perceptual_prefetch(self, on: bool):
        """VGG19 features of the ground-truth frames on the side stream beside the forward pass (default) or inside loss_backward; switch it off for
        steps that will not use the perceptual term (weight 0, no logging): the branch would run for nothing"""
        self._check(self.lib.caddy_set_perceptual_prefetch(self.ctx, 1 if on else 0))

    def set_rollout_fold(self, on: bool):
        """roll-out with the eval-mode BatchNorms folded into the preceding convolutions (default) or as separate launches; takes effect at the next
        start_inference / generate_next"""
        self._check(self.lib.caddy_set_rollout_fold(self.ctx, 1 if on else 0))

    def set_deterministic(self, on: bool):
        """bit-reproducible backward pass (slabs + fixed-order reduces instead of fp32 atomics in arrival order; the forward pass always is): two backward passes over the
        same forward give bit-identical gradients, as the reference's CPU path does.  The library's default since round 5 (< 1 % of the step); False selects the atomics."""
        self._check(self.lib.caddy_set_deterministic(self.ctx, int(bool(on))))

    def set_precision(self, forward: int, backward: int):
        """arithmetic of the model's wide 3x3 convolutions: forward 0 (exact fp32) | 16 (split f16, default); backward 0 | 17 (split bf16, default)"""
        self._check(self.lib.caddy_set_precision(self.ctx, int(forward), int(backward)))

    def __del__(self):
        if getattr(self, "ctx", None):
            self.lib.caddy_ctx_destroy(self.ctx)
            self.ctx = None

    def _err(self):
        return (self.lib.caddy_last_error() or b"").decode()

    def _check(self, rc):
        if rc != 0:
            raise CaddyError(self._err() or f"caddy error {rc}")

    def _stream(self):
        s = torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0
        self.lib.caddy_set_stream(self.ctx, s)

    # ---- parameters (reference state_dict names / layouts) ----
    def view(self, name_or_entry) -> torch.Tensor:
        e = name_or_entry if isinstance(name_or_entry, tuple) else next(t for t in self.table if t[0] == name_or_entry)
        n = 1
        for s in e[2]:
            n *= s
        return self.params[e[1]:e[1] + n].view(e[2])

    def grad_view(self, name) -> torch.Tensor:
        e = next(t for t in self.table if t[0] == name)
        assert e[3] == 0
        n = 1
        for s in e[2]:
            n *= s
        return self.grads[e[1]:e[1] + n].view(e[2])

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        for e in self.table:
            self.view(e).copy_(sd[e[0]].detach().to(self.device, torch.float32))

    def state_dict(self) -> Dict[str, torch.Tensor]:
        return {e[0]: self.view(e).detach().clone() for e in self.table}

    # ---- forward (Model.forward: full-model mode and pretraining mode) ----
    def _prepare(self, obs, noise, samples_in, variations_in):
        B, T, S, H, W = self.B, self.T, self.S, self.H, self.W
        assert tuple(obs.shape) == (B, T, 3 * S, H, W), obs.shape
        dev = self.device
        obs = obs.to(dev, torch.float32).contiguous()
        nz = {k: v.to(dev, torch.float32).contiguous() for k, v in noise.items()}
        cn = Noise(nz["eps_states"].data_ptr(), nz["eps_dirs"].data_ptr(), nz["gumbel_uniform"].data_ptr(),
                   nz["eps_states_rec"].data_ptr(), nz["eps_dirs_rec"].data_ptr())
        si = samples_in.to(dev, torch.float32).contiguous() if samples_in is not None else None
        vi = variations_in.to(dev, torch.float32).contiguous() if variations_in is not None else None
        self._keep = [obs, nz, si, vi]
        self._stream()
        return obs, cn, si, vi

    def _fetch(self, pretraining: bool) -> List:
        """Outputs in the reference's tuple order: forward_full_model (model.py:280-286) or forward_pretraining (:463-468)."""
        B, T, S, H, W, K, Da, Ch = self.B, self.T, self.S, self.H, self.W, self.K, self.Da, self.Ch
        dev, hs, ws = self.device, H // 8, W // 8
        f32 = dict(dtype=torch.float32, device=dev)
        n = T - 1
        Tf = T if pretraining else n                      # reconstructed frames: all T in pretraining mode
        act = {"logits": (B, n, K), "samples": (B, n, K), "ddist": (B, n, 2, Da), "dirs": (B, n, Da), "sdist": (B, T, 2, Da), "ssamp": (B, T, Da), "var": (B, n, Da)}
        if pretraining:
            shapes = {0: (B, Tf, 3, H, W), 2: (B, T, 64, hs, ws), 3: (B, T, 64, hs, ws), 4: (B, T, Ch, hs, ws), 5: (B, n, Ch, hs, ws),
                      7: act["logits"], 8: act["samples"], 9: (B, T, 1, hs, ws)}
            sel_id = 6
        else:
            shapes = {0: (B, Tf, 3, H, W), 2: (B, T, 64, hs, ws), 3: (B, T, 64, hs, ws), 4: (B, n, Ch, hs, ws),
                      6: act["logits"], 7: act["samples"], 8: (B, T, 1, hs, ws), 9: (B, n, 1, hs, ws)}
            sel_id = 5
        shapes.update({10: act["ddist"], 11: act["dirs"], 12: act["sdist"], 13: act["ssamp"], 14: act["var"], 15: act["logits"],
                       16: act["ddist"], 17: act["dirs"], 18: act["sdist"], 19: act["ssamp"]})
        out = [None] * 20
        for i, shp in shapes.items():
            t = torch.empty(shp, **f32)
            self._check(self.lib.caddy_get_output(self.ctx, i, t.data_ptr()))
            out[i] = t
        sel = torch.empty((B, n), dtype=torch.int64, device=dev)
        self._check(self.lib.caddy_get_output(self.ctx, sel_id, sel.data_ptr()))
        out[sel_id] = sel
        multi = [out[0]]
        for r in (1, 2):
            t = torch.empty((B, Tf, 3, H >> r, W >> r), **f32)
            self._check(self.lib.caddy_get_output(self.ctx, 100 + r, t.data_ptr()))
            multi.append(t)
        out[1] = multi
        return out

    def output(self, idx: int, pretraining: bool = False) -> torch.Tensor:
        """One entry of the output tuple (same indices as the reference's tuple) without materialising the others."""
        B, T, K, Da, Ch, hs, ws = self.B, self.T, self.K, self.Da, self.Ch, self.H // 8, self.W // 8
        n = T - 1
        small = {10: (B, n, 2, Da), 11: (B, n, Da), 12: (B, T, 2, Da), 13: (B, T, Da), 14: (B, n, Da), 15: (B, n, K), 16: (B, n, 2, Da), 17: (B, n, Da),
                 18: (B, T, 2, Da), 19: (B, T, Da)}
        if pretraining:
            small.update({2: (B, T, 64, hs, ws), 3: (B, T, 64, hs, ws), 4: (B, T, Ch, hs, ws), 5: (B, n, Ch, hs, ws), 7: (B, n, K), 8: (B, n, K)})
        else:
            small.update({2: (B, T, 64, hs, ws), 3: (B, T, 64, hs, ws), 4: (B, n, Ch, hs, ws), 6: (B, n, K), 7: (B, n, K)})
        t = torch.empty(small[idx], dtype=torch.float32, device=self.device)
        self._stream()
        self._check(self.lib.caddy_get_output(self.ctx, idx, t.data_ptr()))
        return t

    def forward_full(self, obs: torch.Tensor, gt_init: int, tau: float, noise: Dict[str, torch.Tensor], training=True,
                     samples_in: Optional[torch.Tensor] = None, variations_in: Optional[torch.Tensor] = None, fetch_outputs=True) -> List:
        obs, cn, si, vi = self._prepare(obs, noise, samples_in, variations_in)
        self.last_pretraining = False
        self._check(self.lib.caddy_forward_full(self.ctx, obs.data_ptr(), gt_init, float(tau), C.byref(cn), int(training),
                                                si.data_ptr() if si is not None else None, vi.data_ptr() if vi is not None else None))
        return self._fetch(False) if fetch_outputs else None      # fused-loss training: outputs stay in the workspace

    def forward_pretraining(self, obs: torch.Tensor, tau: float, noise: Dict[str, torch.Tensor], training=True,
                            samples_in: Optional[torch.Tensor] = None, variations_in: Optional[torch.Tensor] = None, fetch_outputs=True) -> List:
        obs, cn, si, vi = self._prepare(obs, noise, samples_in, variations_in)
        self.last_pretraining = True
        self._check(self.lib.caddy_forward_pretraining(self.ctx, obs.data_ptr(), float(tau), C.byref(cn), int(training),
                                                       si.data_ptr() if si is not None else None, vi.data_ptr() if vi is not None else None))
        return self._fetch(True) if fetch_outputs else None

    def output_grad(self, idx: int, like: torch.Tensor) -> torch.Tensor:
        """d(loss)/d(output idx) after loss_backward (debug / autograd bridge)."""
        g = torch.empty_like(like)
        self._stream()
        self._check(self.lib.caddy_get_output_grad(self.ctx, idx, g.data_ptr()))
        return g

    # ---- losses + backward (Trainer.compute_losses terms + loss.backward()) ----
    def loss_backward(self, weights: Dict[str, float], smooth_mi=True, mi_alpha=0.2, update_mi_ema=True, perceptual_log=False, diagnostics=False, deferred=False):
        """fused losses + BPTT backward.  -> {name: value} after the GPU has finished the call; with `deferred=True` (GPU only) -> a PendingLosses whose
        `.result()` gives the same dict: the values travel by an asynchronous copy into pinned memory and the host does not wait for the backward pass"""
        if smooth_mi and self.mi_ema is None:
            self.mi_ema = torch.full((self.K, self.K), 1.0 / (self.K * self.K), dtype=torch.float32, device=self.device)
        deferred = bool(deferred) and self.device.type == "cuda"
        lc = LossCfg(weights.get("rec", 0.0), weights.get("states", 0.0), weights.get("entropy", 0.0), weights.get("dir_kl", 0.0),
                     weights.get("mi", 0.0), weights.get("state_kl", 0.0), weights.get("hidden", 0.0), weights.get("mi_entropy", 1.0),
                     self.mi_ema.data_ptr() if smooth_mi else None, mi_alpha, int(update_mi_ema),
                     weights.get("perceptual", 0.0), int(perceptual_log), int(diagnostics), int(deferred))
        with_perc = bool(self.perceptual and self.vgg_loaded and (lc.perceptual != 0.0 or perceptual_log))
        if deferred:
            if self._pinned_losses is None:
                self._pinned_losses = [torch.zeros(LOSS_SLOTS, dtype=torch.float64).pin_memory() for _ in range(2)]
            buf = self._pinned_losses[self._pinned_next]
            self._pinned_next ^= 1
            self._stream()
            self._check(self.lib.caddy_loss_backward(self.ctx, C.byref(lc), C.c_void_p(buf.data_ptr())))
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            return PendingLosses(buf, ev, bool(diagnostics), with_perc)
        host = (C.c_double * LOSS_SLOTS)()
        self._stream()
        self._check(self.lib.caddy_loss_backward(self.ctx, C.byref(lc), host))
        return _losses_dict(host, bool(diagnostics), with_perc)

    def set_action_member(self, member: int):
        """model.action_network.ensamble_size > 1: the action network both A calls of the NEXT forward pass use (model.py:152,274: random.choice(self.action_network))"""
        self._check(self.lib.caddy_set_action_member(self.ctx, int(member)))

    def adam_step(self, step: int, lr=4e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-6, grad_scale=1.0, member_step=None, member_steps=None, s2h_step=None):
        """member_step: step count of the ensemble member this pass used (torch.optim.Adam keeps `step` per parameter, and a member that was not drawn is not stepped);
        None = `step`.  member_steps (one count per member) / s2h_step: the explicit bookkeeping of caddy_adam_step_ex -- 0 skips a range, > 0 steps it (ranges without a
        gradient from the last pass with g = 0: the zero-filling optimizer.zero_grad() of torch < 2.0)"""
        if self.adam_m is None:
            self.adam_m, self.adam_v = torch.zeros_like(self.grads), torch.zeros_like(self.grads)
        self._stream()
        if member_steps is not None or s2h_step is not None:
            ms = (C.c_int * 8)(*([int(x) for x in member_steps] + [0] * 8)[:8]) if member_steps is not None else None
            self._check(self.lib.caddy_adam_step_ex(self.ctx, self.adam_m.data_ptr(), self.adam_v.data_ptr(), lr, betas[0], betas[1], eps, weight_decay, step, ms,
                                                    int(s2h_step or 0), grad_scale))
            return
        self._check(self.lib.caddy_adam_step_member(self.ctx, self.adam_m.data_ptr(), self.adam_v.data_ptr(), lr, betas[0], betas[1], eps,
                                                    weight_decay, step, step if member_step is None else int(member_step), grad_scale))

    # ---- roll-out (Model.start_inference / generate_next) ----
    def sequence_losses_per_frame(self, pretraining=False):
        """evaluator quantities of the last forward from the loss kernels: (l1 (B, Trec), state_mse (B, T)) float64 CPU tensors -- mean |frame - observation[:3]| per
        reconstructed frame and mean (reconstructed state - state)^2 per frame (evaluation/evaluator.py:192,194)"""
        Trec = self.T if pretraining else self.T - 1
        l1 = torch.zeros(self.B, Trec, dtype=torch.float64)
        mse = torch.zeros(self.B, self.T, dtype=torch.float64)
        self._stream()
        self._check(self.lib.caddy_sequence_losses_per_frame(self.ctx, l1.data_ptr(), mse.data_ptr()))
        return l1, mse

    def perceptual_per_frame(self, pretraining=False) -> torch.Tensor:
        """(5, B, Trec) float64 CPU tensor: full-resolution VGG19 feature distance mean |relu{l+1}_1(rec) - relu{l+1}_1(gt)| per reconstructed frame and level, through
        the HIP loss network (evaluation/evaluator.py:55,193: the per-position perceptual loss is sum_l mean_b of column t)"""
        Trec = self.T if pretraining else self.T - 1
        out = torch.zeros(5, self.B, Trec, dtype=torch.float64)
        self._stream()
        self._check(self.lib.caddy_perceptual_per_frame(self.ctx, out.data_ptr()))
        return out

    def numerics_flags(self) -> int:
        """Poll of the per-layer f16 range guards (waits for the stream; reads and clears them).  Bit 0: a split-f16 forward convolution (model or VGG19) met |x| > 65504 since the
        last poll and clamped it; bit 1: a NaN was among them.  The layers that reported move to a forward without a range limit (exact fp32 / split bf16 for VGG19) for good."""
        return int(self.lib.caddy_f16_saturated(self.ctx))

    def f16_saturated(self) -> bool:
        return bool(self.numerics_flags() & 1)

    def fallback_layers(self) -> int:
        """convolution layers this engine has moved off the split-f16 forward after a range-guard report"""
        return int(self.lib.caddy_fallback_layers(self.ctx))

    def start_inference(self):
        self._stream()
        self._check(self.lib.caddy_start_inference(self.ctx))

    def generate_next(self, observation: torch.Tensor, action: int, variation: Optional[torch.Tensor] = None):
        S, H, W = self.S, self.H, self.W
        obs = observation.to(self.device, torch.float32).contiguous()
        assert tuple(obs.shape) == (3 * S, H, W)
        frame = torch.empty((3, H, W), dtype=torch.float32, device=self.device)
        nxt = torch.empty((3 * S, H, W), dtype=torch.float32, device=self.device)
        v = variation.to(self.device, torch.float32).contiguous() if variation is not None else None
        self._stream()
        self._check(self.lib.caddy_generate_next(self.ctx, obs.data_ptr(), int(action), v.data_ptr() if v is not None else None,
                                                 frame.data_ptr(), nxt.data_ptr()))
        return frame, nxt

    CONV_FAMILIES = ["k_conv_fwd<2, 2, 2, 2, 0, 1>", "k_conv_fwd<2, 1, 2, 2, 0, 1>", "k_conv_fwd<1, 1, 2, 2, 0, 1>", "k_conv_fwd<1, 1, 4, 1, 0, 1>",
                     "k_conv_thin_out", "k_conv_thin_in", "k_conv_wgrad<2, 2, 2, 2>", "k_conv_wgrad<1, 2, 2, 2>", "k_conv_wgrad<1, 1, 1, 4>",
                     "k_conv_wgrad_small", "k_wgrad_thin", "k_conv_wgrad_tile", "k_conv_narrow",
                     "k_conv_hx<128>", "k_conv_hx<64>", "k_conv_hx<32>", "k_wgrad_hx", "k_conv_hx<128, 8 waves>"]      # csrc/common.h: CK_* (kernel families of the profiling API)

    def profile_begin(self):
        self._check(self.lib.caddy_profile_begin(C.c_void_p(self.ctx)))

    def profile_records(self, max_records=20000):
        buf = (C.c_double * (7 * max_records))()
        n = self.lib.caddy_profile_records(C.c_void_p(self.ctx), buf, max_records)
        return [tuple(buf[7 * i + j] for j in range(7)) for i in range(n)]

    def profile_phases(self, max_marks=256):
        """[(phase, ms since the previous mark on the main stream)] of the profiled steps; call before profile_end()"""
        names = C.create_string_buffer(48 * max_marks)
        ms = (C.c_float * max_marks)()
        n = self.lib.caddy_profile_phases(C.c_void_p(self.ctx), names, ms, max_marks)
        return [(names.raw[48 * i:48 * (i + 1)].split(b"\0")[0].decode(), ms[i]) for i in range(n)]

    def profile_end(self):
        """-> {kernel: (launches, algorithmic FLOPs, milliseconds, algorithmic bytes)}, HIP events on the launch stream."""
        out = (C.c_double * (4 * len(self.CONV_FAMILIES)))()
        self._check(self.lib.caddy_profile_end(C.c_void_p(self.ctx), out))
        return {n: (int(out[4 * i]), out[4 * i + 1], out[4 * i + 2], out[4 * i + 3]) for i, n in enumerate(self.CONV_FAMILIES)}

    def fusion_counts(self) -> Dict[str, int]:
        """BatchNorm fusion bookkeeping since creation (debug / tests)"""
        out = (C.c_long * 3)()
        self.lib.caddy_debug_fusion_counts(C.c_void_p(self.ctx), out)
        return {"bn_calls": out[0], "stats_from_conv_epilogue": out[1], "never_materialised": out[2]}

    def bn_calls(self) -> Dict[str, int]:
        buf = C.create_string_buffer(128)
        res = {}
        for i in range(self.lib.caddy_bn_layer_count(self.ctx)):
            n = self.lib.caddy_bn_calls(self.ctx, i, buf)
            res[buf.value.decode()] = n
        return res
