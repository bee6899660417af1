#!/usr/bin/env python
"""Headline benchmark: CADDY full-model training step (E -> A -> R -> D forward, fused losses, BPTT backward, gradient
all-reduce for N>1, Adam) on synthetic BAIR-shaped clips -- BASELINE.json configs[1]: 256x256, T=16, B=8 per GPU.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `value` = clips/s over all GPUs with inputs resident in HBM (observations never cross PCIe inside
the step).  The step is the reference's real one: forward_full_model + L1 / VGG19-perceptual / states / KL / MI losses + BPTT + Adam.
Two legs are timed in one run: the full step (`value`, `ms_per_step`; VGG19 perceptual term inside) and `erad_only` -- the same step with the
perceptual term switched off, i.e. exactly the E -> R -> A -> D forward + backward the north star's roofline target is quoted on.
`roofline` is for the dominant kernel of the E/R/A/D leg (`k_conv_hx`: 3x3 convolutions on the 16-bit matrix pipe with split fp32
operands), timed with HIP events on the launch stream during extra profiled steps, plus `erad_hbm_frac` = SURVEY 8(d)'s 55.2 GB / t_erad /
8 TB/s; the dominant kernel of the full step (a VGG19 layer) is kept as `roofline_full_step`.  `exact_fp32_ms_per_step` is the same step on the
exact-fp32 kernels.  `plugin` times the reference's plugin path (`model(config)` / `trainer(...)` factories, `train_epoch` over a synthetic
BatchElement dataset through the pinned double-buffered prefetcher).  `cpu_baseline` is the CPU oracle (port of the reference arithmetic) on a
bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from playablevideogeneration_amd import configs  # noqa: E402
from playablevideogeneration_amd.engine import Engine  # noqa: E402
from playablevideogeneration_amd.init import init_parameters, random_vgg19_state  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, = fp32 vector peak
MFMA16_DENSE_PEAK_TFLOPS = 2500.0 # same guide: bf16 / f16 dense MFMA (v_mfma_f32_32x32x16_{f16,bf16})
SPLIT_PRODUCTS = 3                # conv_hx: a*b = a_hi*b_hi + a_hi*b_lo + a_lo*b_hi on the 16-bit pipe -> 3 MFMA FLOPs per algorithmic FLOP
HBM_PEAK_GBS = 8000.0
PMC_FILE = "profiles/r06_pmc_traffic_{workload}{suffix}.json"      # HBM bytes per launch from THIS round's separate rocprofv3 --pmc passes (tools/gpu_pmc.sh <workload>)
HX_FAMILIES = ("k_conv_hx<128>", "k_conv_hx<64>", "k_conv_hx<32>", "k_wgrad_hx", "k_conv_hx<128, 8 waves>")
VGG_CONVS = [(3, 64, 0), (64, 64, 0), (64, 128, 1), (128, 128, 1), (128, 256, 2), (256, 256, 2), (256, 256, 2), (256, 256, 2), (256, 512, 3),
             (512, 512, 3), (512, 512, 3), (512, 512, 3), (512, 512, 4)]      # (Cin, Cout, number of 2x2 max-pools before): conv1_1 .. conv5_1


DEFER_LOSSES = os.environ.get("CADDY_BENCH_SYNC_LOSSES", "0") != "1"


def vgg_work(n_images, H, W):
    """Algorithmic work of the perceptual loss per step (SURVEY 8d definitions): two forward passes (ground truth + reconstruction) and
    one dgrad pass of the 13 VGG19 convolutions at the three resolutions.  -> (FLOPs, bytes)"""
    fl = by = 0.0
    for r in range(3):
        for cin, cout, pools in VGG_CONVS:
            h, w = (H >> r) >> pools, (W >> r) >> pools
            px = n_images * h * w
            fl += 3 * 2.0 * px * 9 * cin * cout
            by += 3 * 4.0 * (px * cin + px * cout + 9 * cin * cout)
    return fl, by
# SURVEY.md 8(d) / BASELINE.md section 3, BAIR 256^2 T=16 gt=6: per clip forward 336.08 GFLOP, 2.2374 GB activations, 0.5159 GB weights/call
ALGO = {"bair256_t16_b8": dict(gflop_clip_fwd=336.08, act_gb_clip_fwd=2.2374, w_gb_fwd=0.5159),
        "breakout160_t9_b8": dict(gflop_clip_fwd=24.48, act_gb_clip_fwd=0.2943, w_gb_fwd=0.0772),
        "breakout64_t8_b4": dict(gflop_clip_fwd=3.40, act_gb_clip_fwd=0.0405, w_gb_fwd=0.0676)}


T_START = time.time()


def log(msg):
    if os.environ.get("RANK", "0") == "0":
        print(f"[bench +{time.time() - T_START:7.1f}s] {msg}", file=sys.stderr, flush=True)


def make_noise(B, T, K, Da, dev, gen):
    n = T - 1
    return {"eps_states": torch.randn(B * T, Da, device=dev, generator=gen), "eps_dirs": torch.randn(B * n, Da, device=dev, generator=gen),
            "gumbel_uniform": torch.rand(B * n, K, device=dev, generator=gen),
            "eps_states_rec": torch.randn(B * T, Da, device=dev, generator=gen), "eps_dirs_rec": torch.randn(B * n, Da, device=dev, generator=gen)}


def cpu_baseline(wl, perceptual=True):
    """Oracle (CPU port of the reference arithmetic, oracle/caddy_oracle.py) on a bounded sample: ONE clip of the same geometry, forward +
    all losses (VGG19 perceptual term included, random-init weights) + backward, 1 warm-up + 2 timed iterations on the host cores."""
    from oracle import caddy_oracle as O
    d = O.Dims(variant=wl["variant"], actions=wl["actions"], action_dim=wl["action_dim"], hidden=wl["hidden"], stacking=wl["stacking"],
               state_res=(wl["height"] // 8, wl["width"] // 8))
    P = {k: v.clone().requires_grad_(O.is_trainable(k)) for k, v in O.make_params(d, seed=0).items()}
    logical = os.cpu_count() or 1
    try:
        import psutil
        physical = psutil.cpu_count(logical=False) or logical
    except Exception:
        physical = logical
    cores = min(physical, 32)     # oneDNN degrades badly beyond ~32 threads on the many-core GPU host (measured in round 1: > 25 min at 256 threads)
    torch.set_num_threads(cores)
    obs = torch.rand(1, wl["seq_len"], 3 * wl["stacking"], wl["height"], wl["width"]) * 2 - 1
    V = O.make_vgg_params() if perceptual else None
    w = dict(O.DEFAULT_LOSS_WEIGHTS, perceptual=1.0 if perceptual else 0.0)
    times = []
    for it in range(3):
        t0 = time.time()
        orc = O.Oracle(d, P, training=True)
        out = orc.forward_full(obs, wl["gt_init"], tau=wl["tau"])
        total, _, _ = O.full_model_loss(out, obs, w, mi_ema=torch.full((d.K, d.K), 1.0 / d.K ** 2), vgg=V)
        total.backward()
        times.append(time.time() - t0)
        for p in P.values():
            p.grad = None
        P["centroid_estimator.estimated_centroids"] = P["centroid_estimator.estimated_centroids"].detach()
    dt = sum(times[1:]) / len(times[1:])
    res = {"value": 1.0 / dt, "unit": "clips/s", "cores": cores, "kind": "port", "host_physical_cores": physical, "host_logical_cpus": logical,
           "sample": f"1 clip (B=1, T={wl['seq_len']}, {wl['height']}x{wl['width']}): oracle forward + losses"
                     + (" incl. VGG19 perceptual" if perceptual else "") + f" + backward; 1 warm-up + 2 timed iterations ({times[1]:.1f} s, {times[2]:.1f} s)"}
    # ONE iteration of the workload's own batch (B clips through one forward / backward: BatchNorm over the batch, as the step the north star describes), no warm-up beyond the
    # single-clip iterations above; bounded: skipped when the single-clip iteration already took more than 8 s
    Bw = wl["batch"]
    if Bw > 1 and dt < 8.0 and os.environ.get("CADDY_BENCH_CPU_BATCH", "1") != "0":
        obs_b = torch.rand(Bw, wl["seq_len"], 3 * wl["stacking"], wl["height"], wl["width"]) * 2 - 1
        t0 = time.time()
        out = O.Oracle(d, P, training=True).forward_full(obs_b, wl["gt_init"], tau=wl["tau"])
        total, _, _ = O.full_model_loss(out, obs_b, w, mi_ema=torch.full((d.K, d.K), 1.0 / d.K ** 2), vgg=V)
        total.backward()
        tb = time.time() - t0
        res["batch_sample"] = {"value": Bw / tb, "unit": "clips/s", "batch": Bw, "seconds": tb,
                               "sample": f"one iteration at the workload's batch (B={Bw}): the same oracle step on {Bw} clips at once, {tb:.1f} s"}
    return res


def rollout_fps(dev, frames=32):
    """BASELINE.json configs[3]: Tennis hyper-parameters (main model, S=4, Da=5) at 256x256, play.py path:
    start_inference + `frames` x generate_next at batch 1, eval mode, actions i mod 7."""
    c = dict(configs.TENNIS)
    eng = Engine(variant=c["variant"], batch=1, seq_len=2, height=256, width=256, stacking=c["stacking"], actions=c["actions"],
                 action_dim=c["action_dim"], hidden=c["hidden"], device=dev)
    init_parameters(eng, seed=0)
    obs0 = torch.rand(3 * c["stacking"], 256, 256, device=dev) * 2 - 1
    runs = []
    for rep in range(4):                                     # first repetition = warm-up (graph capture, caches); three timed roll-outs
        obs = obs0
        eng.start_inference()
        for i in range(4):
            _, obs = eng.generate_next(obs, i % c["actions"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(frames):
            _, obs = eng.generate_next(obs, i % c["actions"])
        torch.cuda.synchronize()
        if rep:
            runs.append(frames / (time.perf_counter() - t0))
    runs.sort()
    med = runs[len(runs) // 2]
    return {"metric": "rollout frames/sec", "value": med, "unit": "frames/s", "ms_per_frame": 1e3 / med, "runs": runs, "spread": (runs[-1] - runs[0]) / med,
            "config": {"workload": "tennis256_s4_rollout32", "variant": "main", "batch": 1, "frames": frames,
                       "path": "start_inference + generate_next, one captured HIP graph launch per frame"}}


def plugin_config(wl, B, T):
    """The reference's YAML as a dict (configs/01_bair.yaml keys the hot path reads) with the two dotted factory paths pointing at this
    package -- what INTEGRATION.md tells a user of train.py to edit."""
    arch = "playablevideogeneration_amd.model" if wl["variant"] == "main" else "playablevideogeneration_amd.reduced_model"
    return {"data": {"actions_count": wl["actions"]},
            "model": {"architecture": arch, "representation_network": {"state_features": 64, "state_resolution": [wl["height"] // 8, wl["width"] // 8]},
                      "dynamics_network": {"hidden_state_size": wl["hidden"], "random_noise_size": 32},
                      "action_network": {"ensamble_size": 1, "use_gumbel": True, "hard_gumbel": False, "gumbel_temperature": 1.0, "action_space_dimension": wl["action_dim"]},
                      "centroid_estimator": {"alpha": 0.1}},
            "training": {"trainer": "playablevideogeneration_amd.smooth_mi_trainer",
                         "batching": {"batch_size": B, "observation_stacking": wl["stacking"], "observations_count": T, "observations_count_start": T, "observations_count_steps": 1, "num_workers": 0},
                         "use_ground_truth_actions": False, "pretraining_detach": False, "learning_rate": 4e-4, "weight_decay": 1e-6, "lr_schedule": [300000, 10 ** 10], "lr_gamma": 0.3333,
                         "ground_truth_observations_start": wl["gt_init"], "ground_truth_observations_end": wl["gt_init"], "ground_truth_observations_steps": 16000,
                         "gumbel_temperature_start": wl["tau"], "gumbel_temperature_end": wl["tau"], "gumbel_temperature_steps": 20000, "mutual_information_estimation_alpha": 0.2,
                         "pretraining_steps": 0, "max_steps_per_epoch": 10 ** 6,
                         "loss_weights": {"reconstruction_loss_lambda": configs.LOSS_WEIGHTS["rec"], "perceptual_loss_lambda": configs.LOSS_WEIGHTS["perceptual"],
                                          "states_rec_lambda": configs.LOSS_WEIGHTS["states"], "entropy_lambda": configs.LOSS_WEIGHTS["entropy"],
                                          "action_directions_kl_lambda": configs.LOSS_WEIGHTS["dir_kl"], "action_mutual_information_lambda": configs.LOSS_WEIGHTS["mi"],
                                          "action_state_distribution_kl_lambda": configs.LOSS_WEIGHTS["state_kl"]}},
            "logging": {"save_root_directory": "/tmp"}}


def plugin_leg(wl, dev, steps, warmup, perc):
    """The same training step taken through the reference's plugin seam instead of the Engine: `model(config)` / `trainer(config, model, dataset, logger)`
    resolved with importlib from the dotted paths (train.py:38-39,54), `.cuda()`, then `trainer.train_epoch(model, loader)` (trainer.py:552-609) over
    synthetic host batches that reach the GPU through DevicePrefetcher (pinned double-buffered H2D copies on their own stream).  Includes what the
    engine-level loop does not: the reference-order noise drawn on the CPU generator and uploaded, the loss_info diagnostics, the PCIe copy of the
    observations (hidden behind the previous step)."""
    import importlib
    from playablevideogeneration_amd.prefetch import DevicePrefetcher
    B, T, S, H, W = wl["batch"], wl["seq_len"], wl["stacking"], wl["height"], wl["width"]
    cfg = plugin_config(wl, B, T)
    if perc:
        cfg["training"]["vgg19_weights"] = random_vgg19_state(0)
    else:
        cfg["training"]["loss_weights"]["perceptual_loss_lambda"] = 0.0
    model = getattr(importlib.import_module(cfg["model"]["architecture"]), "model")(cfg).cuda()
    trainer = getattr(importlib.import_module(cfg["training"]["trainer"]), "trainer")(cfg, model, None, None)
    trainer.global_step = 20000
    model.train()
    host = torch.rand(B, T, 3 * S, H, W, generator=torch.Generator().manual_seed(4321)) * 2 - 1
    acts = torch.zeros(B, T, dtype=torch.int32)
    make = lambda n: DevicePrefetcher([(host, acts, None, None)] * n, dev)
    trainer.train_epoch(model, make(warmup))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    done = trainer.train_epoch(model, make(steps))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / max(1, done) * 1e3
    del trainer, model
    import gc
    gc.collect()                                             # Model <-> Engine reference cycles: release the BPTT workspace now
    torch.cuda.empty_cache()
    return {"ms_per_step": ms, "clips_per_s": B / ms * 1e3, "steps": done,
            "path": "importlib factories -> model.cuda() -> trainer.train_epoch(model, DevicePrefetcher(host batches)): CPU-generator noise in the reference's "
                    "order, pinned double-buffered H2D copy of the observations, fused losses + BPTT + Adam, loss_info diagnostics"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="bair256_t16_b8", choices=sorted(configs.WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-steps", type=int, default=1)
    ap.add_argument("--no-rollout", action="store_true")
    ap.add_argument("--no-perceptual", action="store_true", help="A/B aid: drop the VGG19 perceptual term (the reported step then says so)")
    ap.add_argument("--no-plugin", action="store_true", help="skip the leg that times the plugin path (model / trainer factories + train_epoch)")
    ap.add_argument("--no-extra-legs", action="store_true", help="only the contract's timed region (+ profiled steps): no erad_only / exact-fp32 / plugin legs")
    ap.add_argument("--quick", action="store_true", help="A/B aid: the timed region + the erad_only leg; no exact-fp32 / deterministic / plugin / roll-out legs")
    a = ap.parse_args()
    if a.quick:
        a.no_plugin = a.no_rollout = True

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    run(a, torch.device("cuda", local), lib=None, backend="nccl")


def load_pmc(workload, perc, erad):
    """HBM bytes per launch / per step: only from a PMC file of THIS round (tools/gpu_pmc.sh), never a stale number"""
    try:
        pmc_file = PMC_FILE.format(workload=workload, suffix="_erad" if erad else "")
        with open(os.path.join(ROOT, pmc_file)) as f:
            pm = json.load(f)
        if pm.get("workload") == workload and pm.get("perceptual") == bool(perc and not erad):
            return pm, pmc_file
    except (OSError, ValueError):
        pass
    return None, None


def roofline_of(fam, recs, nsteps, pm, pm_file, note):
    """-> roofline dict for the family with the most measured time in this profiled leg"""
    name, (n, fl, ms, by) = max(fam.items(), key=lambda kv: kv[1][2])
    ms = max(ms, 1e-9)                                   # (the simulator's events report 0)
    tot_ms = sum(v[2] for v in fam.values())
    split = name in HX_FAMILIES                          # the dominant family runs split operands on the 16-bit matrix pipe
    peak = MFMA16_DENSE_PEAK_TFLOPS / SPLIT_PRODUCTS if split else FP32_MATRIX_PEAK_TFLOPS
    traffic = pm["kernels"].get(name, {}).get("hbm_bytes_per_launch") if pm else None
    achieved = fl / ms / 1e9
    roof = {"kernel": name, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
            "traffic": traffic, "traffic_source": pm_file if traffic is not None else None,
            "peak_note": ("fp32-class products on the 16-bit matrix pipe: 3 MFMA products (hi*hi + hi*lo + lo*hi) per algorithmic product, so the "
                          "algorithmic peak is the dense f16/bf16 MFMA peak 2500 TFLOP/s / 3" if split else "exact-fp32 MFMA peak"),
            "mfma_flops_issued_frac_of_16bit_dense_peak": (SPLIT_PRODUCTS * achieved / MFMA16_DENSE_PEAK_TFLOPS) if split else None,
            "launches_per_step": n // nsteps, "avg_launch_us": ms / n * 1e3, "algorithmic_gflop_per_launch": fl / n / 1e9,
            "algorithmic_bytes_per_launch": by / n, "all_conv_kernels_ms_per_step": tot_ms / nsteps, "note": note,
            "kernels": {k: {"launches": v[0] // nsteps, "tflops": (v[1] / v[2] / 1e9 if v[2] > 0 else 0.0), "ms": v[2] / nsteps,
                            "avg_us": (v[2] / v[0] * 1e3 if v[0] else 0.0)} for k, v in fam.items() if v[0]}}
    kinds = {0: "model_forward", 1: "model_dgrad", 2: "model_wgrad", 3: "vgg19_forward", 4: "vgg19_dgrad"}
    grp = {}
    for kind, _px, _k, _co, _ks, flops, kms in recs:
        g = grp.setdefault(kinds.get(int(kind), "other"), [0, 0.0, 0.0])
        g[0] += 1; g[1] += flops; g[2] += kms
    roof["conv_groups"] = {k: {"launches": v[0] // nsteps, "algorithmic_tflop_per_step": v[1] / nsteps / 1e12, "ms_per_step": v[2] / nsteps,
                               "tflops": (v[1] / v[2] / 1e9 if v[2] > 0 else 0.0)} for k, v in grp.items()}
    return roof


def run(a, dev, lib=None, backend="nccl"):
    """The benchmark proper.  `lib` / `backend` exist so that tests/test_cabi_and_dp.py can drive the multi-rank control flow
    (hooks, collectives, profiled steps on every rank, max-over-ranks timing) on CPU with gloo and the simulator build."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    on_gpu = dev.type == "cuda"
    sync = torch.cuda.synchronize if on_gpu else (lambda: None)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if not dist.is_initialized():
            dist.init_process_group(backend, **({"device_id": dev} if on_gpu else {}))      # RCCL over xGMI
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    wl = configs.WORKLOADS[a.workload]
    B, T, H, W, S, K, Da = wl["batch"], wl["seq_len"], wl["height"], wl["width"], wl["stacking"], wl["actions"], wl["action_dim"]
    too_small = H < 64 or W < 64                           # the quarter-resolution image must survive VGG19's four max-pools
    perc = not getattr(a, "no_perceptual", False) and not too_small
    extra = not getattr(a, "no_extra_legs", False)
    eng = Engine(variant=wl["variant"], batch=B, seq_len=T, height=H, width=W, stacking=S, actions=K, action_dim=Da, hidden=wl["hidden"], device=dev, lib=lib,
                 perceptual=perc)
    log(f"engine created: workspace {eng.ws_bytes / 2**30:.1f} GiB, {eng.n_train} trainable floats")
    init_parameters(eng, seed=0)                            # identical replicas on every rank
    if perc:
        eng.load_vgg(random_vgg19_state(0))                 # VGG19 of the perceptual loss: random-init weights of the real architecture (no network)
    loss_w = dict(configs.LOSS_WEIGHTS, perceptual=configs.LOSS_WEIGHTS["perceptual"] if perc else 0.0)
    loss_w_erad = dict(configs.LOSS_WEIGHTS, perceptual=0.0)
    log("parameters initialised")
    ranks_seen, dp_native = 1, False
    if world > 1:
        # global-batch centroid sums + MI joint matrix (tiny all-reduces) and the bucketed gradient all-reduce (CADDY_DP_OVERLAP=0: one flat all-reduce)
        eng.enable_data_parallel(overlap=os.environ.get("CADDY_DP_OVERLAP", "1") != "0")
        one = torch.ones(1, device=dev)
        dist.all_reduce(one)                                # what the communicator really spans (RCCL over xGMI on the GPU box)
        ranks_seen = int(one.item())
        dp_native = bool(getattr(eng, "_dp_native", False))      # True: the three reductions are issued from C (csrc/dp_rccl.cpp), no Python on the per-step path
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    obs = torch.rand(B, T, 3 * S, H, W, device=dev, generator=gen) * 2 - 1     # each rank owns its shard of the global batch
    step_no = [0]

    def step(weights=None):
        step_no[0] += 1
        eng.forward_full(obs, wl["gt_init"], wl["tau"], make_noise(B, T, K, Da, dev, gen), training=True, fetch_outputs=False)
        # the loss values leave the GPU through an asynchronous copy (caddy_loss_cfg.no_sync), as in the trainer's train_epoch: the host does not wait for the
        # backward pass before it enqueues the optimiser step and the next forward (CADDY_BENCH_SYNC_LOSSES=1: the synchronous read-back)
        losses = eng.loss_backward(loss_w if weights is None else weights, smooth_mi=True, deferred=DEFER_LOSSES)
        if world > 1:
            eng.allreduce_gradients()                       # R / D buckets (91 % of 39.4 MB) were started behind the side stream during the backward; the rest here
        eng.adam_step(step_no[0], lr=4e-4, weight_decay=1e-6, grad_scale=1.0 / world)
        return losses

    def fence():
        if world > 1:
            dist.barrier()
        sync()

    def timed(nsteps, weights=None):
        """`nsteps` steps bracketed by barrier + synchronize on both sides, MAX over ranks -> ms per step"""
        fence()
        t0 = time.perf_counter()
        for _ in range(nsteps):
            out = step(weights)
        fence()
        dt = time.perf_counter() - t0
        if hasattr(out, "result"):
            out = out.result()
        if world > 1:
            t = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = t.item()
        return dt / nsteps * 1e3, out

    for i in range(a.warmup):
        losses = step()
        sync()
        if hasattr(losses, "result"):
            losses = losses.result()
        log(f"warm-up step {i} done, loss {losses['total']:.5f}")
    ms_step, losses = timed(a.steps)                        # ---- the contract's timed region: EXACTLY K steps of the full step ----
    log(f"timed region done: {ms_step:.1f} ms/step")
    clips_s = world * B * 1e3 / ms_step

    # ---- E -> R -> A -> D leg: the same step without the VGG19 term (weight 0 and no ground-truth VGG19 branch beside the forward) ----
    ms_erad = ms_step
    if perc and extra:
        eng.set_perceptual_prefetch(False)
        step(loss_w_erad)
        ms_erad, _ = timed(a.steps, loss_w_erad)
        log(f"erad_only leg: {ms_erad:.1f} ms/step")

    # ---- live per-kernel timing (HIP events on the launch stream) of extra, untimed-for-throughput steps: E/R/A/D leg first, then the full step ----
    roof = roof_full = None
    nprof = a.profile_steps
    legs = [("erad", loss_w_erad)] + ([("full", loss_w)] if perc else [])
    for leg, weights in legs if nprof > 0 else []:
        if leg == "full":
            eng.set_perceptual_prefetch(True)
            step(weights)                                    # (re-establish the prefetched ground-truth branch before profiling)
        if rank == 0:
            eng.profile_begin()
        for _ in range(nprof):                              # the profiled steps contain collectives: every rank takes part
            step(weights)
        if rank == 0:
            recs = eng.profile_records()                    # per launch: (kind, pixels, K, Cout, KS, algorithmic FLOPs, ms)
            fam = eng.profile_end()
            pm, pm_file = load_pmc(a.workload, perc, erad=(leg == "erad" and perc))
            note = "weight-gradient kernels run on a side stream concurrently with the main stream: per-launch durations measured inside the step include that sharing"
            if leg == "full":
                note += "; while profiling, the three VGG19 resolution levels run one after the other (un-profiled steps overlap the two small ones with the full-resolution one)"
            r = roofline_of(fam, recs, nprof, pm, pm_file, note)
            if pm:
                r["step_hbm_measured"] = {"gbytes_per_step": pm["total_hbm_bytes_per_step"] / 1e9, "source": pm_file}
            if leg == "erad":
                roof = r
            else:
                roof_full = r
    if perc:
        eng.set_perceptual_prefetch(True)
    alg = ALGO.get(a.workload)
    if rank == 0 and roof is not None and alg:
        # whole-leg figures on SURVEY 8(d)'s algorithmic work: model bytes = 3*(B*act + W), flops = 3*B*fwd (+ the VGG19 loss network for the full step)
        e_bytes = 3 * (B * alg["act_gb_clip_fwd"] + alg["w_gb_fwd"]) * 1e9
        e_flops = 3 * B * alg["gflop_clip_fwd"] * 1e9
        roof["erad_algorithmic"] = {"gbytes": e_bytes / 1e9, "tflop": e_flops / 1e12}
        roof["erad_ms_per_step"] = ms_erad
        roof["erad_hbm_frac"] = e_bytes / (ms_erad * 1e-3) / (HBM_PEAK_GBS * 1e9)         # the north star's "fraction of the HBM roofline on E->R->A->D fwd+bwd"
        roof["erad_algorithmic_tflops"] = e_flops / (ms_erad * 1e-3) / 1e12
        roof["erad_frac_of_split_mfma_peak"] = e_flops / (ms_erad * 1e-3) / (MFMA16_DENSE_PEAK_TFLOPS / SPLIT_PRODUCTS * 1e12)
        if roof.get("step_hbm_measured"):
            m = roof["step_hbm_measured"]
            m["gb_per_s"] = m["gbytes_per_step"] / (ms_erad * 1e-3); m["frac_of_hbm_peak"] = m["gb_per_s"] / HBM_PEAK_GBS
            m["over_algorithmic"] = m["gbytes_per_step"] / (e_bytes / 1e9)
        if roof_full is not None:
            vfl, vby = vgg_work(B * (T - 1), H, W)
            roof_full["vgg19_algorithmic"] = {"gbytes": vby / 1e9, "tflop": vfl / 1e12}
            roof_full["step_hbm_roofline_frac"] = (e_bytes + vby) / (ms_step * 1e-3) / (HBM_PEAK_GBS * 1e9)
            roof_full["step_algorithmic_tflops"] = (e_flops + vfl) / (ms_step * 1e-3) / 1e12
            roof_full["step_frac_of_split_mfma_peak"] = (e_flops + vfl) / (ms_step * 1e-3) / (MFMA16_DENSE_PEAK_TFLOPS / SPLIT_PRODUCTS * 1e12)
            if roof_full.get("step_hbm_measured"):
                m = roof_full["step_hbm_measured"]
                m["gb_per_s"] = m["gbytes_per_step"] / (ms_step * 1e-3); m["frac_of_hbm_peak"] = m["gb_per_s"] / HBM_PEAK_GBS

    # ---- the same full step on the exact-fp32 kernels (what `dtype: f32` would cost without the split-operand scheme) ----
    ms_exact = None
    quick = getattr(a, "quick", False)
    if extra and world == 1 and on_gpu and not quick:
        eng.set_precision(0, 0)
        if perc:
            eng.set_vgg_precision(0, 0)
        step()
        ms_exact, _ = timed(2)
        eng.set_precision(16, 17)
        if perc:
            eng.set_vgg_precision(16, 17)
        log(f"exact-fp32 kernels: {ms_exact:.1f} ms/step")
    ms_det = ms_det_erad = None      # the arrival-order backward (caddy_set_deterministic(0): fp32 atomics) beside the default bit-reproducible one: what reproducibility costs
    if extra and world == 1 and on_gpu and not quick:
        eng.set_deterministic(False)
        step()
        ms_det, _ = timed(max(2, a.steps // 2))
        if perc:
            eng.set_perceptual_prefetch(False)
            step(loss_w_erad)
            ms_det_erad, _ = timed(max(2, a.steps // 2), loss_w_erad)
            eng.set_perceptual_prefetch(True)
        eng.set_deterministic(True)
        log(f"arrival-order (atomic) backward: {ms_det:.1f} ms/step" + (f", E/R/A/D-only {ms_det_erad:.1f} ms/step" if ms_det_erad else ""))
    if world > 1:
        dist.barrier()
    res = None
    if rank == 0:
        res = {"metric": f"training clips/sec (B x{T}x{H}x{W})", "value": clips_s, "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "arithmetic": "fp32 results; wide 3x3 convolutions as split-f16 (forward) / split-bf16 (gradients) operands on the 16-bit MFMA with fp32 accumulation, everything else fp32",
               "data": "synthetic",
               "config": {"workload": a.workload, "variant": wl["variant"], "per_gpu_batch": B, "global_batch": B * world, "seq_len": T, "frame": [H, W],
                          "gt_init": wl["gt_init"], "parallelism": f"dp{world}",
                          "step": "forward_full_model + L1 / VGG19-perceptual / states / KL / MI losses + BPTT backward + grad all-reduce + Adam"
                                  + (" (VGG19: random-init weights)" if perc else
                                     (" (VGG19 perceptual term not applicable: frames below 64x64)" if too_small else " (VGG19 perceptual term DISABLED by --no-perceptual)"))},
               "loss": losses["total"], "rccl_ranks_seen": ranks_seen, "dp_native": dp_native,
               "erad_only": {"ms_per_step": ms_erad, "clips_per_s": world * B * 1e3 / ms_erad, "what": "the same step with perceptual weight 0 and no VGG19 branch: E -> R -> A -> D forward + L1 / states / KL / MI losses + BPTT + Adam"},
               "exact_fp32_ms_per_step": ms_exact,
               "backward": "bit-reproducible (library default since round 5: slabs + fixed-order folds; two backward passes over one forward give bit-identical gradients)",
               "atomic_backward": {"ms_per_step": ms_det, "erad_only_ms_per_step": ms_det_erad,
                                   "what": "caddy_set_deterministic(0): fp32 atomics in arrival order instead of slabs + fixed-order folds -- the non-reproducible form, for comparison"},
               "roofline": roof, "roofline_full_step": roof_full}
    del eng
    if on_gpu:
        torch.cuda.empty_cache()
    if rank == 0:
        if world == 1 and not a.no_rollout and on_gpu:
            res["rollout"] = rollout_fps(dev)
            log(f"roll-out: {res['rollout']['value']:.1f} frames/s")
        if world == 1 and extra and not getattr(a, "no_plugin", False) and on_gpu:
            # (after the roll-out: the ROCm runtime multiplexes HIP streams onto 4 hardware queues; with this leg's streams -- prefetcher, side, decoder -- still alive
            #  in the process, the roll-out's graph stream shares a queue with one of them and ran at a quarter of its rate.  play.py is its own process.)
            res["plugin"] = plugin_leg(wl, dev, max(a.steps, 12), a.warmup, perc)      # (>= 12 steps: an epoch's fixed costs -- first un-overlapped H2D copy, the last deferred loss read -- are not per-step costs)
            res["plugin"]["vs_engine_step"] = res["plugin"]["ms_per_step"] / ms_step
            log(f"plugin path: {res['plugin']['ms_per_step']:.1f} ms/step ({res['plugin']['vs_engine_step']:.3f} x the engine-level step)")
        if world == 1 and not a.no_cpu_baseline and on_gpu:
            log("cpu baseline (oracle, 1 clip) ...")
            res["cpu_baseline"] = cpu_baseline(wl, perceptual=perc)
            log("cpu baseline done")
        print(json.dumps(res))
    if world > 1 and on_gpu:
        dist.destroy_process_group()
    return res if rank == 0 else None


if __name__ == "__main__":
    main()
