"""Latency of the under-filled (batch-1 roll-out) convolutions: back-to-back launches of one shape through caddy_k_conv_fwd.
   python tools/bench_tiny.py        (A/B: a second library through CADDY_HIP_LIB)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from playablevideogeneration_amd import _lib
from playablevideogeneration_amd._lib import ConvArgs, ConvSrc, PackDesc, round_up
lib = _lib.load(); lib.caddy_k_hx_weight_bytes.restype = C.c_long
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [("E 64->64 @32", 1, 32, 32, 64, 64), ("R lstm0 208->512 @32", 1, 32, 32, 201, 512), ("R lstm1 528->1024 @16", 1, 16, 16, 521, 1024), ("R 272->128 @16", 1, 16, 16, 265, 128),
          ("D 128->128 @64", 1, 64, 64, 128, 128), ("D 128->64 @128", 1, 128, 128, 128, 64), ("D 64->64 @128", 1, 128, 128, 64, 64), ("D 64->32 @256", 1, 256, 256, 64, 32)]
for name, N, H, W, Cin, Cout in SHAPES:
    ldx = round_up(Cin, 4)
    x = torch.randn(N, H, W, ldx, device="cuda"); w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5).contiguous()
    out = torch.empty(N, H, W, round_up(Cout, 4), device="cuda")
    d = PackDesc(); d.nw, d.Co_each, d.Cin, d.KS, d.nseg = 1, Cout, Cin, 3, 1
    d.seg_off[0], d.seg_C[0], d.seg_Cpad[0] = 0, Cin, round_up(Cin, 16)
    d.Cout, d.Cout_pad, d.Ktot = Cout, round_up(Cout, lib.caddy_k_conv_pick_bn(Cout)), round_up(Cin, 16)
    d.w[0] = w.data_ptr()
    wp = torch.zeros(9 * d.Cout_pad * d.Ktot, device="cuda"); lib.caddy_k_pack_fwd(C.byref(d), C.c_void_p(wp.data_ptr()), st)
    split = torch.zeros(16 * N * H * W * round_up(Cout, 4), device="cuda")
    res = []
    for label, prec, use_split in (("fp32+split", 0, True), ("fp32", 0, False), ("f16x3+split", 16, True), ("f16x3", 16, False)):
        a = ConvArgs(); a.src[0] = ConvSrc(x.data_ptr(), H * W * ldx, ldx, Cin, round_up(Cin, 16), 0)
        a.nsrc, a.N, a.H, a.W, a.KS, a.wp, a.Ktot, a.Cout, a.Cout_pad = 1, N, H, W, 3, wp.data_ptr(), d.Ktot, Cout, d.Cout_pad
        a.out, a.out_sn, a.out_ld = out.data_ptr(), H * W * out.shape[3], out.shape[3]
        if use_split: a.split_scratch, a.split_cap = split.data_ptr(), split.numel()
        if prec:
            rows = round_up(Cout, lib.caddy_k_hx_pick_bn(Cout))
            wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), -1, rows, 2), dtype=torch.uint8, device="cuda")
            lib.caddy_k_pack_hx(C.byref(d), C.c_void_p(wq.data_ptr()), rows, -1, prec, st); a.wq, a.precision = wq.data_ptr(), prec
        for _ in range(5): lib.caddy_k_conv_fwd(C.byref(a), st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): lib.caddy_k_conv_fwd(C.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        res.append(f"{label} {e0.elapsed_time(e1) / 50 * 1e3:6.1f} us")
    print(f"{name:24s} " + " | ".join(res), flush=True)
