#!/usr/bin/env python
"""Headline benchmark: CADDY full-model training step (E -> A -> R -> D forward, fused losses, BPTT backward, gradient
all-reduce for N>1, Adam) on synthetic BAIR-shaped clips -- BASELINE.json configs[1]: 256x256, T=16, B=8 per GPU.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Prints ONE JSON line (rank 0).  `value` = clips/s over all GPUs with inputs resident in HBM (observations never cross PCIe inside
the step).  The step is the reference's real one: forward_full_model + L1 / VGG19-perceptual / states / KL / MI losses + BPTT + Adam.
`roofline` is for the dominant kernel family (conv_hx: 3x3 convolutions on the 16-bit matrix pipe with split fp32 operands), timed
with HIP events on the launch stream during extra profiled steps; `cpu_baseline` is the CPU oracle (port of the reference arithmetic)
on a bounded sample of the same workload.
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
PMC_FILE = "profiles/r02_pmc_traffic_{workload}.json"      # HBM bytes per launch from THIS round's separate rocprofv3 --pmc passes (tools/gpu_pmc.sh <workload>)
HX_FAMILIES = ("k_conv_hx<128>", "k_conv_hx<64>", "k_conv_hx<32>", "k_wgrad_hx", "k_conv_hx<128, 8 waves>")
VGG_CONVS = [(3, 64, 0), (64, 64, 0), (64, 128, 1), (128, 128, 1), (128, 256, 2), (256, 256, 2), (256, 256, 2), (256, 256, 2), (256, 512, 3),
             (512, 512, 3), (512, 512, 3), (512, 512, 3), (512, 512, 4)]      # (Cin, Cout, number of 2x2 max-pools before): conv1_1 .. conv5_1


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
    return {"value": 1.0 / dt, "unit": "clips/s", "cores": cores, "kind": "port", "host_physical_cores": physical, "host_logical_cpus": logical,
            "sample": f"1 clip (B=1, T={wl['seq_len']}, {wl['height']}x{wl['width']}): oracle forward + losses"
                      + (" incl. VGG19 perceptual" if perceptual else "") + f" + backward; 1 warm-up + 2 timed iterations ({times[1]:.1f} s, {times[2]:.1f} s)"}


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
    a = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback on the product path)")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    run(a, torch.device("cuda", local), lib=None, backend="nccl")


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
    perc = not getattr(a, "no_perceptual", False) and H >= 64 and W >= 64
    eng = Engine(variant=wl["variant"], batch=B, seq_len=T, height=H, width=W, stacking=S, actions=K, action_dim=Da, hidden=wl["hidden"], device=dev, lib=lib,
                 perceptual=perc)
    log(f"engine created: workspace {eng.ws_bytes / 2**30:.1f} GiB, {eng.n_train} trainable floats")
    init_parameters(eng, seed=0)                            # identical replicas on every rank
    if perc:
        eng.load_vgg(random_vgg19_state(0))                 # VGG19 of the perceptual loss: random-init weights of the real architecture (no network)
    loss_w = dict(configs.LOSS_WEIGHTS, perceptual=configs.LOSS_WEIGHTS["perceptual"] if perc else 0.0)
    log("parameters initialised")
    if world > 1:
        # global-batch centroid sums + MI joint matrix (tiny all-reduces) and the bucketed gradient all-reduce (CADDY_DP_OVERLAP=0: one flat all-reduce)
        eng.enable_data_parallel(overlap=os.environ.get("CADDY_DP_OVERLAP", "1") != "0")
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    obs = torch.rand(B, T, 3 * S, H, W, device=dev, generator=gen) * 2 - 1     # each rank owns its shard of the global batch
    step_no = [0]

    def step():
        step_no[0] += 1
        eng.forward_full(obs, wl["gt_init"], wl["tau"], make_noise(B, T, K, Da, dev, gen), training=True, fetch_outputs=False)
        losses = eng.loss_backward(loss_w, smooth_mi=True)
        if world > 1:
            eng.allreduce_gradients()                       # R / D buckets (91 % of 39.4 MB) were started behind the side stream during the backward; the rest here
        eng.adam_step(step_no[0], lr=4e-4, weight_decay=1e-6, grad_scale=1.0 / world)
        return losses

    def fence():
        if world > 1:
            dist.barrier()
        sync()

    for i in range(a.warmup):
        losses = step()
        sync()
        log(f"warm-up step {i} done, loss {losses['total']:.5f}")
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()
    ms_step = dt / a.steps * 1e3
    log(f"timed region done: {ms_step:.1f} ms/step")
    clips_s = world * B * a.steps / dt

    # live per-kernel timing (HIP events on the launch stream) of extra, untimed-for-throughput steps
    roof = None
    if world > 1 and a.profile_steps > 0 and rank != 0:
        for _ in range(a.profile_steps):                    # the profiled steps contain collectives: every rank takes part
            step()
    if rank == 0 and a.profile_steps > 0:
        eng.profile_begin()
        for _ in range(a.profile_steps):
            step()
        recs = eng.profile_records()                         # per launch: (kind, pixels, K, Cout, KS, algorithmic FLOPs, ms)
        fam = eng.profile_end()
        name, (n, fl, ms, by) = max(fam.items(), key=lambda kv: kv[1][2])
        ms = max(ms, 1e-9)                                   # (the simulator's events report 0)
        tot_ms = sum(v[2] for v in fam.values())
        split = name in HX_FAMILIES                          # the dominant family runs split operands on the 16-bit matrix pipe
        peak = MFMA16_DENSE_PEAK_TFLOPS / SPLIT_PRODUCTS if split else FP32_MATRIX_PEAK_TFLOPS
        traffic, traffic_src, step_pmc = None, None, None    # HBM bytes per launch: only from a PMC file of THIS round; never a stale number
        try:
            pmc_file = PMC_FILE.format(workload=a.workload)
            with open(os.path.join(ROOT, pmc_file)) as f:
                pm = json.load(f)
            if pm.get("workload") == a.workload and pm.get("perceptual") == bool(perc):
                traffic = pm["kernels"].get(name, {}).get("hbm_bytes_per_launch")
                traffic_src = pmc_file
                step_pmc = pm.get("total_hbm_bytes_per_step")
        except (OSError, ValueError):
            pass
        achieved = fl / ms / 1e9
        roof = {"kernel": name, "bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
                "traffic": traffic, "traffic_source": traffic_src,
                "peak_note": ("fp32-class products on the 16-bit matrix pipe: 3 MFMA products (hi*hi + hi*lo + lo*hi) per algorithmic product, so the "
                              "algorithmic peak is the dense f16/bf16 MFMA peak 2500 TFLOP/s / 3" if split else "exact-fp32 MFMA peak"),
                "mfma_flops_issued_frac_of_16bit_dense_peak": (SPLIT_PRODUCTS * achieved / MFMA16_DENSE_PEAK_TFLOPS) if split else None,
                "launches_per_step": n // a.profile_steps, "avg_launch_us": ms / n * 1e3, "algorithmic_gflop_per_launch": fl / n / 1e9,
                "algorithmic_bytes_per_launch": by / n, "all_conv_kernels_ms_per_step": tot_ms / a.profile_steps,
                "note": "weight-gradient kernels run on a side stream concurrently with the main stream: per-launch durations measured inside the step include that sharing",
                "kernels": {k: {"launches": v[0] // a.profile_steps, "tflops": (v[1] / v[2] / 1e9 if v[2] > 0 else 0.0), "ms": v[2] / a.profile_steps,
                                "avg_us": (v[2] / v[0] * 1e3 if v[0] else 0.0)} for k, v in fam.items() if v[0]}}
        kinds = {0: "model_forward", 1: "model_dgrad", 2: "model_wgrad", 3: "vgg19_forward", 4: "vgg19_dgrad"}
        grp = {}
        for kind, _px, _k, _co, _ks, flops, kms in recs:
            g = grp.setdefault(kinds.get(int(kind), "other"), [0, 0.0, 0.0])
            g[0] += 1; g[1] += flops; g[2] += kms
        roof["conv_groups"] = {k: {"launches": v[0] // a.profile_steps, "algorithmic_tflop_per_step": v[1] / a.profile_steps / 1e12, "ms_per_step": v[2] / a.profile_steps,
                                   "tflops": (v[1] / v[2] / 1e9 if v[2] > 0 else 0.0)} for k, v in grp.items()}
        alg = ALGO.get(a.workload)
        if alg:   # whole-step figures on SURVEY 8(d)'s algorithmic work: model bytes = 3*(B*act + W), flops = 3*B*fwd; + the VGG19 loss network
            step_bytes = 3 * (B * alg["act_gb_clip_fwd"] + alg["w_gb_fwd"]) * 1e9
            step_flops = 3 * B * alg["gflop_clip_fwd"] * 1e9
            roof["erad_algorithmic"] = {"gbytes": step_bytes / 1e9, "tflop": step_flops / 1e12}
            if perc:
                vfl, vby = vgg_work(B * (T - 1), H, W)
                roof["vgg19_algorithmic"] = {"gbytes": vby / 1e9, "tflop": vfl / 1e12}
                step_bytes += vby; step_flops += vfl
            roof["step_hbm_roofline_frac"] = step_bytes / (ms_step * 1e-3) / (HBM_PEAK_GBS * 1e9)
            if step_pmc:      # rocprofv3 PMC passes of this round (all kernels of one step, FETCH_SIZE doubled + WRITE_SIZE) over THIS run's step time
                roof["step_hbm_measured"] = {"gbytes_per_step": step_pmc / 1e9, "gb_per_s": step_pmc / 1e9 / (ms_step * 1e-3),
                                             "frac_of_hbm_peak": step_pmc / (ms_step * 1e-3) / (HBM_PEAK_GBS * 1e9), "source": traffic_src}
            roof["step_algorithmic_tflops"] = step_flops / (ms_step * 1e-3) / 1e12
            roof["step_frac_of_split_mfma_peak"] = step_flops / (ms_step * 1e-3) / (MFMA16_DENSE_PEAK_TFLOPS / SPLIT_PRODUCTS * 1e12)
    if world > 1:
        dist.barrier()
    if rank == 0:
        res = {"metric": f"training clips/sec (B x{T}x{H}x{W})", "value": clips_s, "unit": "clips/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
               "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
               "arithmetic": "fp32 results; wide 3x3 convolutions as split-f16 (forward) / split-bf16 (gradients) operands on the 16-bit MFMA with fp32 accumulation, everything else fp32",
               "data": "synthetic",
               "config": {"workload": a.workload, "variant": wl["variant"], "per_gpu_batch": B, "global_batch": B * world, "seq_len": T, "frame": [H, W],
                          "gt_init": wl["gt_init"], "parallelism": f"dp{world}",
                          "step": "forward_full_model + L1 / VGG19-perceptual / states / KL / MI losses + BPTT backward + grad all-reduce + Adam"
                                  + (" (VGG19: random-init weights)" if perc else " (VGG19 perceptual term DISABLED by --no-perceptual)")},
               "loss": losses["total"], "roofline": roof}
        if world == 1 and not a.no_rollout and on_gpu:
            del eng
            torch.cuda.empty_cache()
            res["rollout"] = rollout_fps(dev)
            log(f"roll-out: {res['rollout']['value']:.1f} frames/s")
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
