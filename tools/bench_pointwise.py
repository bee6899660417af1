"""Micro-benchmark of the HBM-bound point-wise / reduction kernels at the shapes of one BAIR time step (run on the GPU box):
    python tools/bench_pointwise.py [libA.so libB.so ...]       (default: the in-tree library)
Every kernel is timed as a dependent chain of launches on one stream (what the BPTT chain is), us per launch, and its algorithmic bytes / time."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from playablevideogeneration_amd import _lib  # noqa: E402
from playablevideogeneration_amd._lib import TV  # noqa: E402

SHAPES = [(8, 128, 128, 16), (8, 64, 64, 32), (8, 32, 32, 64), (8, 32, 32, 128), (8, 16, 16, 256), (8, 64, 64, 128), (8, 128, 128, 64), (8, 256, 256, 32)]
REPS = 30


def tv(t, Cc):
    N, H, W, ld = t.shape
    v = TV(t.data_ptr(), N, H, W, Cc, H * W * ld, ld)
    v._keep = t
    return v


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(REPS):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / REPS * 1e3


def run(path):
    lib = _lib.load(path)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = []
    for N, H, W, Cc in SHAPES:
        el = N * H * W * Cc
        x, dout, outm, dx = (torch.randn(N, H, W, Cc, device="cuda") for _ in range(4))
        hi = torch.randn(N, 2 * H, 2 * W, Cc, device="cuda") if el * 4 <= 1 << 26 else None
        mean, invstd, gamma, scale, shift, rm, rv = (torch.rand(Cc, device="cuda") + 0.5 for _ in range(7))
        sums = torch.zeros(2 * Cc, dtype=torch.float64, device="cuda")
        scratch = torch.zeros(512 * 2 * 1024, dtype=torch.float64, device="cuda")
        dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
        P = lambda t: C.c_void_p(t.data_ptr())
        B = C.byref
        tx, tdo, tom, tdx = tv(x, Cc), tv(dout, Cc), tv(outm, Cc), tv(dx, Cc)
        cases = [
            ("stats+finalize", 1, lambda: lib.caddy_k_bn_stats_finalize(B(tx), P(sums), P(scratch), P(gamma), P(gamma), P(rm), P(rv), P(mean), P(invstd), P(scale), P(shift), st)),
            ("bn_apply", 2, lambda: lib.caddy_k_bn_apply(B(tx), P(scale), P(shift), None, None, None, 1, B(tdx), st)),
            ("bn_bwd lazy (reduce+apply)", 5, lambda: lib.caddy_k_bn_bwd_lazy(B(tdo), B(tx), P(mean), P(invstd), P(gamma), P(scale), P(shift), 1, P(sums), P(scratch), B(tdx), P(dg), P(db), st)),
            ("bn_bwd_apply outm", 4, lambda: lib.caddy_k_bn_bwd_apply_assign(B(tdo), B(tom), B(tx), P(mean), P(invstd), P(gamma), P(sums), B(tdx), st)),
        ]
        if hi is not None:
            th = tv(hi, Cc)
            cases += [("pool2 (from 2x)", 1.25, lambda: lib.caddy_k_pool2(B(th), B(tdx), st)), ("pool2_bwd assign (to 2x)", 1.25, lambda: lib.caddy_k_pool2_bwd_assign(B(tdo), B(th), st)),
                      ("up2 (to 2x)", 5, lambda: lib.caddy_k_up2(B(tx), B(th), st)), ("up2_bwd (from 2x)", 5, lambda: lib.caddy_k_up2_bwd(B(th), B(tdx), st))]
        for name, passes, fn in cases:
            us = timed(fn)
            rows.append((f"{N}x{H}x{W}x{Cc}", name, us, passes * el * 4 / us / 1e6))
    return rows


if __name__ == "__main__":
    libs = sys.argv[1:] or [None]
    res = [run(p) for p in libs]
    print("shape            kernel                          " + "".join(f"{os.path.basename(p or 'in-tree'):>28s}" for p in libs))
    for i, (shape, name, _, _) in enumerate(res[0]):
        print(f"{shape:16s} {name:30s} " + "".join(f"{r[i][2]:12.1f} us {r[i][3]:8.2f} TB/s" for r in res))
