"""CPU-side checks: the gfx950 library builds, loads and exports every symbol include/caddy_hip.h declares; host-only entry
points (parameter table, workspace sizing, argument validation) behave; data-parallel step over gloo with world_size 2."""
import ctypes as C
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def hiplib():
    from playablevideogeneration_amd.csrc import build as B
    return C.CDLL(B.build())


def test_every_declared_symbol_is_exported(hiplib):
    hdr = open(os.path.join(ROOT, "include", "caddy_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(caddy_\w+)\s*\(", hdr))
    assert len(names) > 25
    missing = [n for n in sorted(names) if not hasattr(hiplib, n)]
    assert not missing, missing


def test_host_only_entry_points(hiplib):
    from playablevideogeneration_amd.engine import CaddyConfig, ParamInfo, _bind
    from oracle import caddy_oracle as O
    lib = _bind(hiplib)
    cfg = CaddyConfig(0, 8, 16, 256, 256, 1, 7, 2, 128, 1, 0, 1, 0.1)
    assert lib.caddy_trainable_floats(C.byref(cfg)) >= 9856353          # trainable scalars of BAIR-main (+ 16-byte padding)
    d = O.Dims(variant="main", actions=7, action_dim=2, hidden=128, stacking=1, state_res=(32, 32))
    ref = {n: tuple(s) for n, s in O.param_table(d) if not n.endswith("num_batches_tracked")}
    info, got = ParamInfo(), {}
    for i in range(lib.caddy_param_count(C.byref(cfg))):
        assert lib.caddy_param_info_get(C.byref(cfg), i, C.byref(info)) == 0
        got[info.name.decode()] = tuple(info.shape[:info.ndim])
    assert got == ref                                                     # reference state_dict names and shapes
    assert 30 * 2 ** 30 < lib.caddy_workspace_bytes(C.byref(cfg)) < 64 * 2 ** 30
    bad = CaddyConfig(0, 8, 16, 250, 256, 1, 7, 2, 128, 1, 0, 1, 0.1)    # height not a multiple of 16
    assert lib.caddy_workspace_bytes(C.byref(bad)) == 0 and b"invalid" in lib.caddy_last_error()
    assert not lib.caddy_ctx_create(C.byref(cfg), None, None, None, 0)   # null buffers are rejected, nothing is launched


def test_product_loader_has_no_fallback(monkeypatch, tmp_path):
    from playablevideogeneration_amd import _lib
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "missing.so"))
    monkeypatch.setattr(_lib, "_lib", None)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.load()


def _dp_worker(rank, world, port, out, gpu=False):
    import torch.distributed as dist
    from playablevideogeneration_amd.engine import Engine
    from oracle import caddy_oracle as O
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    if gpu:                                           # one process per GPU over RCCL (tests/test_model_gpu.py, needs >= 2 devices)
        torch.cuda.set_device(rank)
        dev, lib = f"cuda:{rank}", None
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    else:
        from tests.emu.loader import load_emu
        dev, lib = "cpu", load_emu()
        dist.init_process_group("gloo", rank=rank, world_size=world)
    d = O.Dims(variant="reduced", actions=3, action_dim=1, hidden=64, stacking=1, state_res=(2, 2))
    P = O.make_params(d, seed=3)
    g = torch.Generator().manual_seed(100 + rank)
    obs = torch.rand(1, 3, 3, 16, 16, generator=g) * 2 - 1
    noise = {"eps_states": torch.randn(3, 1, generator=g), "eps_dirs": torch.randn(2, 1, generator=g), "gumbel_uniform": torch.rand(2, 3, generator=g),
             "eps_states_rec": torch.randn(3, 1, generator=g), "eps_dirs_rec": torch.randn(2, 1, generator=g)}

    def fresh(overlap):
        e = Engine(variant="reduced", batch=1, seq_len=3, height=16, width=16, stacking=1, actions=3, action_dim=1, hidden=64, device=dev, lib=lib)
        e.load_state_dict(P)
        e.enable_data_parallel(overlap=overlap)
        e.forward_full(obs, 1, 0.8, noise, training=True, fetch_outputs=False)
        e.loss_backward(dict(O.DEFAULT_LOSS_WEIGHTS))
        return e

    eng = fresh(overlap=False)                      # plain path: one flat all-reduce after the backward
    small = {"logits": eng.output(6), "rec_logits": eng.output(15), "dir_dist": eng.output(10)}      # this rank's shard of the small action tensors
    local = eng.grads.clone()
    dist.all_reduce(eng.grads)
    eng2 = fresh(overlap=True)                      # bench.py's path: R / D buckets start during loss_backward, the rest afterwards
    if getattr(eng2, "_dp_native", False):          # MI355X: the library's own RCCL communicator (dp_rccl.cpp) -- no Python on the per-step path
        eng2.lib.caddy_dp_bucket_floats.restype = __import__("ctypes").c_long
        assert eng2.lib.caddy_dp_bucket_floats(eng2.ctx) > 0.5 * eng2.grads.numel()
        eng2.hook_host_seconds, eng2.hook_calls = 0.0, 0
    else:
        assert len(eng2._early) == 2 and sum(c for _, c, _ in eng2._early) > 0.5 * eng2.grads.numel()
    eng2.allreduce_gradients()
    eng.adam_step(1, grad_scale=1.0 / world)
    if gpu:
        torch.cuda.synchronize()
    cpu = lambda t: t.detach().cpu().clone()
    torch.save({"local": cpu(local), "reduced": cpu(eng.grads), "reduced_overlap": cpu(eng2.grads), "params": cpu(eng.params[:eng.n_train]),
                "centroids": cpu(eng.view("centroid_estimator.estimated_centroids")), "mi_ema": cpu(eng.mi_ema), "small": {k: cpu(v) for k, v in small.items()},
                "centroids_before": P["centroid_estimator.estimated_centroids"].clone(),
                "hook_host_us": 1e6 * eng2.hook_host_seconds / max(1, eng2.hook_calls)}, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_data_parallel_step_gloo_world2(tmp_path):
    dp_world2_case(tmp_path, gpu=False)


def dp_world2_case(tmp_path, gpu):
    import torch.multiprocessing as mp
    port = 29500 + os.getpid() % 2000
    mp.spawn(_dp_worker, args=(2, port, str(tmp_path), gpu), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.allclose(r0["reduced"], r0["local"] + r1["local"], atol=1e-6)       # sum over ranks
    assert torch.equal(r0["reduced"], r1["reduced"]) and torch.equal(r0["params"], r1["params"])   # trainable replicas stay identical (BN running stats are rank-local, as under nn.DataParallel)
    assert not torch.equal(r0["local"], r1["local"])                                 # shards really differed
    # bucketed, overlapped all-reduce == flat all-reduce (two separate runs: the simulator's float atomics are not order-deterministic)
    assert torch.equal(r0["reduced_overlap"], r1["reduced_overlap"])
    assert torch.allclose(r0["reduced_overlap"], r0["reduced"], atol=1e-4 * r0["reduced"].abs().max().item())
    # global-batch semantics of the small reductions: identical centroids and MI estimator state on every rank
    assert torch.equal(r0["centroids"], r1["centroids"]) and torch.equal(r0["mi_ema"], r1["mi_ema"])
    from oracle import caddy_oracle as O
    # ... and they EQUAL what one process computes on the concatenated batch, as the reference does on GPU0 over the gathered outputs
    # (training/losses.py:262-265: joint matrix summed over all B*(T-1) samples before symmetrise / normalise; centroid_estimator.py:61-63)
    cat = {k: torch.cat([r0["small"][k], r1["small"][k]], 0) for k in r0["small"]}
    K = cat["logits"].shape[-1]
    _, ema = O.mutual_information_loss(torch.softmax(cat["logits"], -1), torch.softmax(cat["rec_logits"], -1), lamb=1.0, ema=torch.full((K, K), 1.0 / (K * K)), alpha=0.2)
    assert torch.allclose(r0["mi_ema"], ema, atol=1e-6), (r0["mi_ema"], ema)
    d = O.Dims(variant="reduced", actions=3, action_dim=1, hidden=64, stacking=1, state_res=(2, 2))
    orc = O.Oracle(d, {"centroid_estimator.estimated_centroids": r0["centroids_before"].clone()}, training=True)
    orc.update_centroids(cat["dir_dist"], torch.softmax(cat["logits"], -1))
    assert torch.allclose(r0["centroids"], orc.P["centroid_estimator.estimated_centroids"], atol=1e-6)
    return r0["hook_host_us"]


def _bench_worker(rank, world, port, out):
    """bench.run() exactly as torchrun would drive it on N GPUs, but gloo + the simulator build + a tiny workload."""
    import argparse
    import json
    import torch.distributed as dist
    from tests.emu.loader import load_emu
    from playablevideogeneration_amd import configs
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    configs.WORKLOADS["tiny"] = dict(configs.BREAKOUT, batch=1, seq_len=3, height=32, width=32, gt_init=2, tau=0.8)
    a = argparse.Namespace(gpus=world, steps=2, warmup=1, workload="tiny", no_cpu_baseline=True, profile_steps=1, no_rollout=True)
    res = bench.run(a, torch.device("cpu"), lib=load_emu(), backend="gloo")
    if rank == 0:
        with open(os.path.join(out, "bench.json"), "w") as f:
            json.dump(res, f)
    else:
        assert res is None
    dist.barrier()


def test_bench_multi_rank_control_flow_gloo_world2(tmp_path):
    """The N>1 path of bench.py (all-reduce hook, gradient all-reduce, profiled step on every rank, barriers, max-over-ranks
    timing, one JSON line from rank 0) cannot be launched on a multi-GPU node from here: run its control flow on CPU."""
    import json
    import torch.multiprocessing as mp
    port = 31500 + os.getpid() % 2000
    mp.spawn(_bench_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    res = json.load(open(tmp_path / "bench.json"))
    assert res["n_gpus"] == 2 and res["config"]["global_batch"] == 2 and res["config"]["parallelism"] == "dp2"
    assert res["scaling"] == "weak" and res["value"] > 0 and res["steps"] == 2 and res["roofline"] is not None
    assert res["rccl_ranks_seen"] == 2 and res["erad_only"]["ms_per_step"] > 0 and "erad_hbm_frac" not in res["roofline"]      # (no SURVEY 8d figures for the tiny workload)
    for k in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data"):
        assert k in res


def test_bench_world8_through_the_launch_script_on_the_simulator(tmp_path):
    """`bash tools/launch_dp.sh 8 2 1` -- the exact command line the driver's scaling run uses (torch.distributed.run, one process per GPU, 127.0.0.1) -- with the
    script swapped for its simulator twin (tests/dp_sim_bench.py: gloo, tiny workload): eight ranks rendezvous, every rank takes part in the MI / centroid reductions and the bucketed
    gradient all-reduce, the line is printed once with n_gpus 8, the ranks the communicator really spans (rccl_ranks_seen) 8, weak scaling, global batch 8.  No hardware scaling
    curve exists (one GPU per lease): this is what can be checked without one."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "bench8.json"
    env = dict(os.environ, CADDY_DP_SCRIPT="tests/dp_sim_bench.py", CADDY_DP_SIM_OUT=str(out), MASTER_PORT=str(29600 + os.getpid() % 300), OMP_NUM_THREADS="1",
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(["bash", os.path.join(root, "tools", "launch_dp.sh"), "8", "2", "1"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    res = json.load(open(out))
    assert res["n_gpus"] == 8 and res["rccl_ranks_seen"] == 8 and res["config"]["global_batch"] == 8 and res["config"]["parallelism"] == "dp8"
    assert res["scaling"] == "weak" and res["value"] > 0 and res["steps"] == 2 and res["warmup"] == 1 and res["roofline"] is not None
    assert res["value"] == pytest.approx(8 * 1 * 1e3 / res["ms_per_step"])      # whole-job clips/s over all ranks


def _trainer_worker(rank, world, port, out, skew=0.0, ens=1):
    import time
    import torch.distributed as dist
    from tests.test_host_api_emu import _config, _make_model
    from playablevideogeneration_amd import smooth_mi_trainer
    from oracle import caddy_oracle as O
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = _config()
    cfg["logging"] = {"save_root_directory": out}
    if ens > 1:
        import random
        cfg["model"]["action_network"]["ensamble_size"] = ens
        random.seed(1000 + 17 * rank)      # every rank's own draws would differ: rank 0's must win
    m = _make_model(cfg)
    d = O.Dims.from_config(dict(cfg, model=dict(cfg["model"], architecture="model.reduced_model.model")))
    m.load_state_dict(O.make_params(d, seed=7))
    m.train()
    tr = smooth_mi_trainer.trainer(cfg, m, dataset=None, logger=None)
    tr.global_step = 20000
    obs = torch.rand(1, 4, 3, 32, 32, generator=torch.Generator().manual_seed(10 + rank)) * 2 - 1      # each rank: its own shard
    members = []
    for i in range(2 if ens == 1 else 4):
        torch.manual_seed(50 + 7 * rank + i)
        time.sleep(skew * ((rank * 3 + i) % world))      # ranks arrive at the collectives in a different order every step
        tr.compute_losses(m, (obs, None, None, None), 4)
        members.append(m.last_member)
        time.sleep(skew * ((rank + 2 * i + 1) % world))
        tr.optimizer_step(m)
    torch.save({"params": m._flat[:m.n_train].clone(), "centroids": m.centroid_estimator.get_estimated_centroids().clone(), "members": members,
                "native": bool(getattr(m.engine(1, 4), "_dp_native", False))}, os.path.join(out, f"t{rank}.pt"))
    dist.barrier()


def test_trainer_mirror_is_data_parallel_aware_gloo_world2(tmp_path):
    """Model / Trainer mirrors under torch.distributed: hooks registered on first use, gradients summed in optimizer_step -> identical replicas"""
    import torch.multiprocessing as mp
    port = 33500 + os.getpid() % 2000
    mp.spawn(_trainer_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    t0, t1 = torch.load(tmp_path / "t0.pt"), torch.load(tmp_path / "t1.pt")
    assert torch.equal(t0["params"], t1["params"]) and torch.equal(t0["centroids"], t1["centroids"])


def test_trainer_mirror_ensemble_under_data_parallel_gloo_world2(tmp_path):
    """ensamble_size = 2 under torch.distributed (round 5): the reference's single process draws ONE action network per forward for all its replicas (model.py:152) -- here rank 0's
    draw is broadcast.  The ranks seed Python's `random` differently, so without the broadcast they would train different members; with it every step uses the same member on both
    ranks (both members get drawn over the four steps) and the replicas end bit-identical, per-member Adam state included"""
    import random
    import torch.multiprocessing as mp
    port = 37500 + os.getpid() % 2000
    mp.spawn(_trainer_worker, args=(2, port, str(tmp_path), 0.0, 2), nprocs=2, join=True)
    t0, t1 = torch.load(tmp_path / "t0.pt"), torch.load(tmp_path / "t1.pt")
    assert t0["members"] == t1["members"]
    random.seed(1000)
    assert t0["members"] == [random.choice(range(2)) for _ in range(4)]       # rank 0's own sequence
    random.seed(1017)
    assert t0["members"] != [random.choice(range(2)) for _ in range(4)]       # (rank 1's would have differed: the case tests something)
    assert torch.equal(t0["params"], t1["params"]) and torch.equal(t0["centroids"], t1["centroids"])


def test_trainer_mirror_data_parallel_gloo_world4_skewed_ranks(tmp_path):
    """four ranks with unequal step timing (every rank sleeps a different time before the forward and before the optimiser step, differently each step): the order in which
    ranks reach the three reductions varies, the replicas must still end identical -- ordering bugs of the collective sequence show up as a hang or as diverging replicas"""
    import torch.multiprocessing as mp
    port = 35500 + os.getpid() % 2000
    mp.spawn(_trainer_worker, args=(4, port, str(tmp_path), 0.15), nprocs=4, join=True)
    ts = [torch.load(tmp_path / f"t{r}.pt") for r in range(4)]
    for t in ts[1:]:
        assert torch.equal(ts[0]["params"], t["params"]) and torch.equal(ts[0]["centroids"], t["centroids"])


# ---- the library's NATIVE data-parallel path (csrc/dp_rccl.cpp) on more than one rank without a GPU: a shared-memory stand-in for librccl.so (tests/emu/fake_rccl.cpp) ----
def _fake_rccl_env():
    from tests.emu.build_emu import build_fake_rccl
    return dict(CADDY_RCCL_LIB=build_fake_rccl(), CADDY_DP_NATIVE="force", FAKE_RCCL_TIMEOUT_S="120")


def _native_worker(rank, world, port, out, kind, skew):
    os.environ.update(_fake_rccl_env())
    if kind == "step":
        _dp_worker(rank, world, port, out)
    else:
        _trainer_worker(rank, world, port, out, skew)


def test_native_dp_step_on_fake_rccl_world2(tmp_path):
    """dp_rccl.cpp -- unique ids, ncclCommInitRank x 2 (one communicator per stream) behind the watchdog thread, the MI / centroid reductions on `comm`, the
    R / D gradient buckets on `comm2` from inside the backward, caddy_allreduce_grads' gap arithmetic for the rest -- with TWO ranks, on the simulator, through a librccl stand-in that
    fails on any collective whose order or size differs between the ranks.  Same assertions as the gloo twin: sum over ranks, identical replicas, bucketed == flat, MI matrix /
    centroids equal to one process on the concatenated batch."""
    import torch.multiprocessing as mp
    port = 38500 + os.getpid() % 1000
    mp.spawn(_native_worker, args=(2, port, str(tmp_path), "step", 0.0), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.allclose(r0["reduced"], r0["local"] + r1["local"], atol=1e-6) and not torch.equal(r0["local"], r1["local"])
    assert torch.equal(r0["reduced"], r1["reduced"]) and torch.equal(r0["params"], r1["params"])
    assert torch.equal(r0["reduced_overlap"], r1["reduced_overlap"])
    assert torch.allclose(r0["reduced_overlap"], r0["reduced"], atol=1e-4 * r0["reduced"].abs().max().item())
    assert torch.equal(r0["centroids"], r1["centroids"]) and torch.equal(r0["mi_ema"], r1["mi_ema"])
    assert r0["hook_host_us"] == 0.0          # no Python callback on the per-step path: the worker took the native branch (it asserts caddy_dp_bucket_floats > half the gradient)


def test_native_dp_trainer_on_fake_rccl_world4_skewed_ranks(tmp_path):
    """the trainer mirror on the native path with four ranks that reach the collectives at different times (sleeps that differ per rank and per step): both communicators keep one
    total order on every rank (the stand-in returns ncclInvalidUsage otherwise, a missing rank is a time-out) and the replicas end bit-identical"""
    import torch.multiprocessing as mp
    port = 39500 + os.getpid() % 1000
    mp.spawn(_native_worker, args=(4, port, str(tmp_path), "trainer", 0.1), nprocs=4, join=True)
    ts = [torch.load(tmp_path / f"t{r}.pt") for r in range(4)]
    assert all(t["native"] for t in ts)                     # every rank's engine took the C path
    for t in ts[1:]:
        assert torch.equal(ts[0]["params"], t["params"]) and torch.equal(ts[0]["centroids"], t["centroids"])


def test_bench_world8_native_path_on_fake_rccl(tmp_path):
    """`tools/launch_dp.sh 8` with bench.run() on the NATIVE path: eight processes, 16 communicators' worth of rendezvous, every collective of the two timed steps, the profiled step
    and the erad leg through dp_rccl.cpp (`dp_native` in the JSON line; nobody fell back to the hooks)"""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = tmp_path / "bench8.json"
    env = dict(os.environ, CADDY_DP_SCRIPT="tests/dp_sim_bench.py", CADDY_DP_SIM_OUT=str(out), MASTER_PORT=str(29900 + os.getpid() % 90), OMP_NUM_THREADS="1",
               PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), **_fake_rccl_env())
    r = subprocess.run(["bash", os.path.join(root, "tools", "launch_dp.sh"), "8", "2", "1"], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "using the torch.distributed hooks" not in r.stderr      # nobody fell back
    res = json.load(open(out))
    assert res["n_gpus"] == 8 and res["rccl_ranks_seen"] == 8 and res["config"]["parallelism"] == "dp8" and res["dp_native"] is True


def _misuse_worker(rank, world, name, out):
    import ctypes
    from tests.emu.build_emu import build_fake_rccl
    os.environ["FAKE_RCCL_TIMEOUT_S"] = "20"
    lib = ctypes.CDLL(build_fake_rccl())

    class Uid(ctypes.Structure):
        _fields_ = [("b", ctypes.c_char * 128)]
    uid = Uid(); uid.b = name.encode()
    comm = ctypes.c_void_p()
    lib.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Uid, ctypes.c_int]
    assert lib.ncclCommInitRank(ctypes.byref(comm), world, uid, rank) == 0
    lib.ncclAllReduce.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    buf = (ctypes.c_float * 8)(*[float(rank + 1)] * 8)
    rc1 = lib.ncclAllReduce(buf, buf, 8, 7, 0, comm, None)                        # same call on both ranks: 1 + 2
    first = list(buf)
    rc2 = lib.ncclAllReduce(buf, buf, 8 if rank == 0 else 4, 7, 0, comm, None)    # the ranks disagree about the collective
    torch.save({"rc1": rc1, "first": first, "rc2": rc2}, os.path.join(out, f"m{rank}.pt"))


def test_fake_rccl_detects_mismatched_collectives(tmp_path):
    """the stand-in itself: a matching all-reduce sums in rank order, a collective whose size differs between the ranks is ncclInvalidUsage (5) on every rank, not a silent reduction"""
    import torch.multiprocessing as mp
    name = f"/caddy_fake_rccl_test_{os.getpid()}"
    mp.spawn(_misuse_worker, args=(2, name, str(tmp_path)), nprocs=2, join=True)
    m0, m1 = torch.load(tmp_path / "m0.pt"), torch.load(tmp_path / "m1.pt")
    assert m0["rc1"] == m1["rc1"] == 0 and m0["first"] == m1["first"] == [3.0] * 8
    assert m0["rc2"] == m1["rc2"] == 5


def test_dp_init_watchdog_turns_a_missing_rank_into_an_error(tmp_path):
    """caddy_dp_init with world 2 and only ONE rank calling it: ncclCommInitRank (the stand-in blocks like the real rendezvous) never returns; the watchdog thread of dp_rccl.cpp
    gives up after CADDY_DP_INIT_TIMEOUT_S and the call returns -3 with a message naming the rank -- in a subprocess, because the blocked helper thread is detached by design"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import ctypes as C, os, time, torch\n"
        "from tests.emu.loader import load_emu\n"
        "from playablevideogeneration_amd.engine import Engine\n"
        "lib = load_emu()\n"
        "e = Engine(variant='reduced', batch=1, seq_len=3, height=16, width=16, stacking=1, actions=3, action_dim=1, hidden=64, device='cpu', lib=lib)\n"
        "buf = C.create_string_buffer(256)\n"
        "assert e.lib.caddy_dp_unique_id(buf) == 0\n"
        "t0 = time.time()\n"
        "rc = e.lib.caddy_dp_init(e.ctx, buf.raw, 2, 0, 1)\n"
        "print('RC', rc, round(time.time() - t0, 1), e._err())\n"
    )
    env = dict(os.environ, CADDY_DP_INIT_TIMEOUT_S="2", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), **_fake_rccl_env())
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300, cwd=root)
    line = [l for l in r.stdout.splitlines() if l.startswith("RC")]
    assert line, (r.stdout[-500:], r.stderr[-2000:])
    parts = line[0].split()
    assert parts[1] == "-3" and float(parts[2]) < 30 and "did not return within 2 s on rank 0 of 2" in line[0]
