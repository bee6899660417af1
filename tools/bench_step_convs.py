"""The wide 3x3 convolutions of one BAIR time step (batch 8) in isolation: forward (split f16), dgrad (split bf16, accumulating) and weight gradient (split bf16) at the
shapes the in-situ profile lists (profiles/*_phases_and_layers.txt).  Run on the GPU box:  python tools/bench_step_convs.py [lib.so ...]   (BENCH_ONLY=<substring> filters)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from playablevideogeneration_amd import _lib  # noqa: E402
from playablevideogeneration_amd._lib import ConvArgs, ConvSrc, PackDesc, WgradArgs, round_up  # noqa: E402

ONLY = os.environ.get("BENCH_ONLY")
REPS = int(os.environ.get("BENCH_REPS", "20"))
# kind, name, N, H, W, Cin (reduction side), Cout
SHAPES = [
    ("fwd", "D res 128->128 @64", 8, 64, 64, 128, 128), ("fwd", "D res 64->64 @128", 8, 128, 128, 64, 64), ("fwd", "D up 64->32 @256", 8, 256, 256, 64, 32),
    ("fwd", "D up 128->64 @128", 8, 128, 128, 128, 64), ("fwd", "R lstm1 gates 528->1024 @16", 8, 16, 16, 528, 1024), ("fwd", "R lstm0 gates 272->512 @32", 8, 32, 32, 272, 512),
    ("fwd", "R lstm2 gates 208->512 @32", 8, 32, 32, 208, 512), ("fwd", "R same 272->128 @16", 8, 16, 16, 272, 128), ("fwd", "E res 64->64 @32", 8, 32, 32, 64, 64),
    ("dgrad", "R gates 512->128 @32", 8, 32, 32, 512, 128), ("dgrad", "D res 128->128 @64", 8, 64, 64, 128, 128), ("dgrad", "D res 64->64 @128", 8, 128, 128, 64, 64),
    ("dgrad", "D up 32->64 @256", 8, 256, 256, 32, 64), ("dgrad", "R gates 1024->256 @16", 8, 16, 16, 1024, 256), ("dgrad", "D up 64->128 @128", 8, 128, 128, 64, 128),
    ("dgrad", "R 256->128 @32", 8, 32, 32, 256, 128), ("dgrad", "R 512->64 @32", 8, 32, 32, 512, 64), ("dgrad", "R 128->128 @32", 8, 32, 32, 128, 128), ("dgrad", "E 64->64 @32", 8, 32, 32, 64, 64),
    ("wgrad", "D up 64->32 @256 x5", 40, 256, 256, 64, 32), ("wgrad", "D res 128->128 @64 x5", 40, 64, 64, 128, 128), ("wgrad", "D res 64->64 @128 x5", 40, 128, 128, 64, 64),
    ("wgrad", "D up 128->64 @128 x5", 40, 128, 128, 128, 64), ("wgrad", "R lstm1 gates 528->1024 @16 x5", 40, 16, 16, 528, 1024), ("wgrad", "R lstm0 gates 272->512 @32 x5", 40, 32, 32, 272, 512),
    ("wgrad", "R lstm2 gates 208->512 @32 x5", 40, 32, 32, 208, 512),
    ("head7", "D final 32->3 k7 @256", 8, 256, 256, 32, 3), ("head3", "D final 64->3 k3 @128", 8, 128, 128, 64, 3), ("head3", "D final 128->3 k3 @64", 8, 64, 64, 128, 3),
]


def run(path):
    lib = _lib.load(path)
    lib.caddy_k_hx_weight_bytes.restype = C.c_long
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rows = []
    for kind, name, N, H, W, Cin, Cout in SHAPES:
        if ONLY and ONLY not in name and ONLY != kind:
            continue
        ldx = round_up(Cin, 4)
        x = torch.randn(N, H, W, ldx, device="cuda")
        KS = 7 if kind == "head7" else 3
        flops = 2.0 * N * H * W * KS * KS * Cin * Cout
        d = PackDesc()
        d.nw, d.Co_each, d.Cin, d.KS, d.nseg = 1, Cout, Cin, 3, 1
        d.seg_off[0], d.seg_C[0], d.seg_Cpad[0] = 0, Cin, round_up(Cin, 16)
        d.Cout, d.Cout_pad, d.Ktot = Cout, round_up(Cout, lib.caddy_k_conv_pick_bn(Cout)), round_up(Cin, 16)
        if kind.startswith("head"):      # FinalBlock heads: conv_head.hip (precision 16, weights split inside the kernel) + tanh
            wp = torch.randn(KS * KS * d.Cout_pad * d.Ktot, device="cuda") * 0.05
            out = torch.zeros(N, H, W, 4, device="cuda")
            bias = torch.zeros(4, device="cuda")
            a = ConvArgs()
            a.src[0] = ConvSrc(x.data_ptr(), H * W * ldx, ldx, Cin, round_up(Cin, 16), 0)
            a.nsrc, a.N, a.H, a.W, a.KS, a.wp, a.Ktot, a.Cout, a.Cout_pad = 1, N, H, W, KS, wp.data_ptr(), d.Ktot, Cout, d.Cout_pad
            a.out, a.out_sn, a.out_ld, a.bias, a.act = out.data_ptr(), H * W * 4, 4, bias.data_ptr(), 1
            a.precision = int(os.environ.get("BENCH_HEAD_PREC", "16"))
            aux = torch.zeros(128 * 1024 // 4, device="cuda")
            a.aux = aux.data_ptr()
            fn = lambda: lib.caddy_k_conv_fwd(C.byref(a), st)
        elif kind == "wgrad":
            dy = torch.randn(N, H, W, round_up(Cout, 4), device="cuda")
            dwp = torch.zeros(9 * d.Cout_pad * d.Ktot, device="cuda")
            wa = WgradArgs()
            wa.src[0] = ConvSrc(x.data_ptr(), H * W * ldx, ldx, Cin, round_up(Cin, 16), 0)
            wa.nsrc, wa.N, wa.H, wa.W, wa.KS = 1, N, H, W, 3
            wa.dy, wa.dy_sn, wa.dy_ld, wa.Cout, wa.Cout_pad, wa.Ktot, wa.dwp, wa.slabs = dy.data_ptr(), H * W * dy.shape[3], dy.shape[3], Cout, d.Cout_pad, d.Ktot, dwp.data_ptr(), 0
            wa.precision = 17
            fn = lambda: lib.caddy_k_conv_wgrad(C.byref(wa), st)
        else:
            w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5).contiguous()
            d.w[0] = w.data_ptr()
            wp = torch.zeros(9 * d.Cout_pad * d.Ktot, device="cuda")
            out = torch.zeros(N, H, W, round_up(Cout, 4), device="cuda")
            split = torch.zeros(9 * 4096 * 256, device="cuda")
            prec = 16 if kind == "fwd" else 17
            rowsp = round_up(Cout, lib.caddy_k_hx_pick_bn(Cout))
            wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), -1, rowsp, 2), dtype=torch.uint8, device="cuda")
            lib.caddy_k_pack_hx(C.byref(d), C.c_void_p(wq.data_ptr()), rowsp, -1, prec, st)
            a = ConvArgs()
            a.src[0] = ConvSrc(x.data_ptr(), H * W * ldx, ldx, Cin, round_up(Cin, 16), 0)
            a.nsrc, a.N, a.H, a.W, a.KS, a.wp, a.Ktot, a.Cout, a.Cout_pad = 1, N, H, W, 3, wp.data_ptr(), d.Ktot, Cout, d.Cout_pad
            a.out, a.out_sn, a.out_ld = out.data_ptr(), H * W * out.shape[3], out.shape[3]
            a.wq, a.precision = wq.data_ptr(), prec
            if kind == "dgrad":
                a.accumulate = 1
            else:
                a.split_scratch, a.split_cap = split.data_ptr(), split.numel()
            fn = lambda: lib.caddy_k_conv_fwd(C.byref(a), st)
        for _ in range(3):
            assert fn() == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / REPS
        rows.append((kind, name, ms * 1e3, flops / ms / 1e9))
    return rows


if __name__ == "__main__":
    libs = sys.argv[1:] or [None]
    res = [run(p) for p in libs]
    print("kind  layer                             " + "".join(f"{os.path.basename(os.path.dirname(p)) + '/' + os.path.basename(p) if p else 'in-tree':>30s}" for p in libs))
    for i, (kind, name, _, _) in enumerate(res[0]):
        print(f"{kind:5s} {name:34s} " + "".join(f"{r[i][2]:14.1f} us {r[i][3]:7.1f} TF/s" for r in res), flush=True)
