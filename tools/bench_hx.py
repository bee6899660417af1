"""Micro-benchmark of conv_hx.hip (split-operand 16-bit MFMA convolution) vs the exact-fp32 k_conv_fwd at the BAIR / VGG19 shapes.
Run on the GPU box:  python tools/bench_hx.py        (BENCH_ONLY=<substring> filters)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from playablevideogeneration_amd import _lib  # noqa: E402
from playablevideogeneration_amd._lib import ConvArgs, ConvSrc, PackDesc, round_up  # noqa: E402

lib = _lib.load()
lib.caddy_k_hx_weight_bytes.restype = C.c_long
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ONLY = os.environ.get("BENCH_ONLY")
SHAPES = [  # name, N, H, W, Cin, Cout
    ("R lstm0 201->512 @32", 8, 32, 32, 224, 512), ("R lstm1 521->1024 @16", 8, 16, 16, 544, 1024), ("R same0 137->256 @32", 8, 32, 32, 160, 256),
    ("R up 265->128 @16", 8, 16, 16, 288, 128), ("D 128->128 @64", 8, 64, 64, 128, 128), ("D 128->64 @128", 8, 128, 128, 128, 64),
    ("D 64->64 @128", 8, 128, 128, 64, 64), ("D 64->32 @256", 8, 256, 256, 64, 32), ("A 64->128 @32 x128f", 128, 32, 32, 64, 128),
    ("E 32->64 @64 x128f", 128, 64, 64, 32, 64), ("E 64->64 @32 x128f", 128, 32, 32, 64, 64),
    ("VGG 64->64 @256 x60", 60, 256, 256, 64, 64), ("VGG 128->128 @128 x60", 60, 128, 128, 128, 128), ("VGG 256->256 @64 x60", 60, 64, 64, 256, 256),
    ("VGG 512->512 @32 x60", 60, 32, 32, 512, 512), ("VGG 512->512 @16 x60", 60, 16, 16, 512, 512),
]
for name, N, H, W, Cin, Cout in SHAPES:
    if ONLY and ONLY not in name:
        continue
    ldx = round_up(Cin, 4)
    x = torch.randn(N, H, W, ldx, device="cuda")
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5).contiguous()
    out = torch.empty(N, H, W, round_up(Cout, 4), device="cuda")
    d = PackDesc()
    d.nw, d.Co_each, d.Cin, d.KS, d.nseg = 1, Cout, Cin, 3, 1
    d.seg_off[0], d.seg_C[0], d.seg_Cpad[0] = 0, Cin, round_up(Cin, 16)
    d.Cout, d.Cout_pad, d.Ktot = Cout, round_up(Cout, lib.caddy_k_conv_pick_bn(Cout)), round_up(Cin, 16)
    d.w[0] = w.data_ptr()
    wp = torch.zeros(9 * d.Cout_pad * d.Ktot, device="cuda")
    lib.caddy_k_pack_fwd(C.byref(d), C.c_void_p(wp.data_ptr()), st)
    flops = 2.0 * N * H * W * 9 * Cin * Cout
    split = torch.zeros(8 * N * H * W * round_up(Cout, 4) if N * H * W <= 16384 else 4, device="cuda")
    res = []
    for prec, planes in ((0, 0), (16, 2), (17, 2), (18, 1)):
        a = ConvArgs()
        a.src[0] = ConvSrc(x.data_ptr(), H * W * ldx, ldx, Cin, round_up(Cin, 16), 0)
        a.nsrc, a.N, a.H, a.W, a.KS, a.wp, a.Ktot, a.Cout, a.Cout_pad = 1, N, H, W, 3, wp.data_ptr(), d.Ktot, Cout, d.Cout_pad
        a.out, a.out_sn, a.out_ld = out.data_ptr(), H * W * out.shape[3], out.shape[3]
        if N * H * W <= 16384:
            a.split_scratch, a.split_cap = split.data_ptr(), split.numel()
        if prec:
            rows = round_up(Cout, lib.caddy_k_hx_pick_bn(Cout))
            wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), -1, rows, planes), dtype=torch.uint8, device="cuda")
            lib.caddy_k_pack_hx(C.byref(d), C.c_void_p(wq.data_ptr()), rows, -1, prec, st)
            a.wq, a.precision = wq.data_ptr(), prec
        for _ in range(3):
            assert lib.caddy_k_conv_fwd(C.byref(a), st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            lib.caddy_k_conv_fwd(C.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        res.append(f"{ {0: 'fp32', 16: 'f16x3', 17: 'bf16x3', 18: 'f16x1'}[prec]} {ms * 1e3:8.1f} us {flops / ms / 1e9:6.1f} TF")
    print(f"{name:26s} " + " | ".join(res), flush=True)
