"""Micro-benchmark of the MFMA conv kernels at the BAIR shapes (run on the GPU box):  python tools/bench_conv.py"""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from playablevideogeneration_amd import _lib
from playablevideogeneration_amd._lib import ConvArgs, ConvSrc, WgradArgs, round_up

lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
import os
pass
ONLY = os.environ.get("BENCH_ONLY")
KIND = os.environ.get("BENCH_KIND")
SHAPES = [  # name, N, H, W, Cin, Cout, KS   (names starting with "dgrad" run the forward kernel in accumulate mode)
    ("dgrad lstm1-h 1024->256 @16", 8, 16, 16, 1024, 256, 3), ("dgrad lstm0-h 512->128 @32", 8, 32, 32, 512, 128, 3),
    ("same0 144->256 @32", 8, 32, 32, 144, 256, 3), ("up 272->128 @16", 8, 16, 16, 272, 128, 3),
    ("lstm0 201->512 @32", 8, 32, 32, 208, 512, 3), ("lstm1 521->1024 @16", 8, 16, 16, 528, 1024, 3),
    ("dec 128->128 @64", 8, 64, 64, 128, 128, 3), ("dec 128->64 @128", 8, 128, 128, 128, 64, 3),
    ("dec 64->64 @128", 8, 128, 128, 64, 64, 3), ("dec 64->32 @256", 8, 256, 256, 64, 32, 3),
    ("final 32->3 k7 @256", 8, 256, 256, 32, 3, 7), ("enc 16->16 @128", 128, 128, 128, 16, 16, 3),
]
SHAPES += [("dgrad lstm0-x 512->64 @32", 8, 32, 32, 512, 64, 3), ("dgrad lstm1-x 1024->256 @16", 8, 16, 16, 1024, 256, 3), ("same2 144->128 @32", 8, 32, 32, 144, 128, 3),
           ("stem 3->16 @256x128f", 128, 256, 256, 3, 16, 3), ("stem 3->16 @256x8f", 8, 256, 256, 3, 16, 3), ("final 64->3 k3 @128", 8, 128, 128, 64, 3, 3),
           ("fdgrad 3->32 k7 @256", 8, 256, 256, 3, 32, 7), ("fdgrad 3->64 k3 @128", 8, 128, 128, 3, 64, 3), ("fdgrad 3->128 k3 @64", 8, 64, 64, 3, 128, 3),
           ("final 128->3 k3 @64", 8, 64, 64, 128, 3, 3), ("sdgrad 16->3 k7 @256", 8, 256, 256, 16, 3, 7),
           ("narrow 32->32 @64x128f", 128, 64, 64, 32, 32, 3), ("narrow 16->32 @128x128f", 128, 128, 128, 16, 32, 3), ("narrow 16->16 @128x8f", 8, 128, 128, 16, 16, 3),
           ("narrow 32->32 @256x8f", 8, 256, 256, 32, 32, 3), ("narrow 32->16 @128x128f", 128, 128, 128, 32, 16, 3),
           ("A res0 64->128 @32x128f", 128, 32, 32, 64, 128, 3), ("A res1 128->128 @16x128f", 128, 16, 16, 128, 128, 3), ("E 32->64 @64x128f", 128, 64, 64, 32, 64, 3)]
for name, N, H, W, Cin, Cout, KS in SHAPES:
    if ONLY and ONLY not in name:
        continue
    ldx = round_up(Cin, 4)
    x = torch.randn(N, H, W, ldx, device="cuda")
    bn = lib.caddy_k_conv_pick_bn(Cout)
    cp = round_up(Cout, bn)
    Kt = round_up(Cin, 16)
    wp = torch.randn(KS * KS * cp * Kt, device="cuda") * 0.01
    out = torch.empty(N, H, W, round_up(Cout, 4), device="cuda")
    a = ConvArgs()
    a.src[0] = ConvSrc(x.data_ptr(), H * W * ldx, ldx, Cin, Kt, 0)
    a.nsrc, a.N, a.H, a.W, a.KS, a.wp, a.Ktot, a.Cout, a.Cout_pad = 1, N, H, W, KS, wp.data_ptr(), Kt, Cout, cp
    a.out, a.out_sn, a.out_ld = out.data_ptr(), H * W * out.shape[3], out.shape[3]
    a.accumulate = 1 if name.startswith('dgrad') else 0
    aux = torch.zeros(128 * 1024 // 4, device='cuda')
    a.aux = aux.data_ptr()
    split = torch.zeros(9 * 4096 * 256, device='cuda')
    a.split_scratch, a.split_cap = split.data_ptr(), split.numel()
    dy = torch.randn(N, H, W, round_up(Cout, 4), device="cuda")
    dwp = torch.zeros_like(wp)
    wa = WgradArgs()
    wa.src[0] = a.src[0]
    wa.nsrc, wa.N, wa.H, wa.W, wa.KS = 1, N, H, W, KS
    wa.dy, wa.dy_sn, wa.dy_ld, wa.Cout, wa.Cout_pad, wa.Ktot, wa.dwp, wa.slabs = dy.data_ptr(), H * W * dy.shape[3], dy.shape[3], Cout, cp, Kt, dwp.data_ptr(), 0
    wa.precision = int(os.environ.get("BENCH_WGRAD_PREC", "0"))
    flops = 2.0 * N * H * W * KS * KS * Cin * Cout
    for label, fn in (("fwd", lambda: lib.caddy_k_conv_fwd(C.byref(a), st)), ("wgrad", lambda: lib.caddy_k_conv_wgrad(C.byref(wa), st))):
        if KIND and KIND != label:
            continue
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:24s} {label:5s} {ms*1e3:9.1f} us  {flops/ms/1e9:7.1f} TFLOP/s (useful)")
