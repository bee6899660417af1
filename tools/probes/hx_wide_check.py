"""Result vs the exact-fp32 kernel, and time, of k_conv_hx at VGG19 shapes with >= 256 output channels.  Written for the 256-channel-block variant of the 8-wave tile
(CADDY_HX_WIDE=1 in the experiment build; the variant was not kept -- profiles/r03_experiments.md -- and the switch no longer exists: the script now times the product kernel)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from playablevideogeneration_amd import _lib  # noqa: E402
from playablevideogeneration_amd._lib import ConvArgs, ConvSrc, PackDesc, round_up  # noqa: E402

lib = _lib.load()
lib.caddy_k_hx_weight_bytes.restype = C.c_long
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, N, H, W, Cin, Cout in [("VGG 256->256 @64 x60", 60, 64, 64, 256, 256), ("VGG 512->512 @32 x60", 60, 32, 32, 512, 512), ("VGG 256->512 @32 x60", 60, 32, 32, 256, 512),
                                 ("VGG 128->256 @64 x120", 120, 64, 64, 128, 256), ("VGG 512->512 @32 x120", 120, 32, 32, 512, 512)]:
    torch.manual_seed(0)
    x = torch.randn(N, H, W, Cin, device="cuda")
    w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5).contiguous()
    bias = torch.randn(Cout, device="cuda")
    d = PackDesc()
    d.nw, d.Co_each, d.Cin, d.KS, d.nseg = 1, Cout, Cin, 3, 1
    d.seg_off[0], d.seg_C[0], d.seg_Cpad[0] = 0, Cin, round_up(Cin, 16)
    d.Cout, d.Cout_pad, d.Ktot = Cout, round_up(Cout, lib.caddy_k_conv_pick_bn(Cout)), round_up(Cin, 16)
    d.w[0] = w.data_ptr()
    wp = torch.zeros(9 * d.Cout_pad * d.Ktot, device="cuda")
    lib.caddy_k_pack_fwd(C.byref(d), C.c_void_p(wp.data_ptr()), st)
    outs = {}
    for prec in (0, 16):
        out = torch.zeros(N, H, W, Cout, device="cuda")
        a = ConvArgs()
        a.src[0] = ConvSrc(x.data_ptr(), H * W * Cin, Cin, Cin, round_up(Cin, 16), 0)
        a.nsrc, a.N, a.H, a.W, a.KS, a.wp, a.Ktot, a.Cout, a.Cout_pad = 1, N, H, W, 3, wp.data_ptr(), d.Ktot, Cout, d.Cout_pad
        a.out, a.out_sn, a.out_ld = out.data_ptr(), H * W * Cout, Cout
        a.bias, a.act = bias.data_ptr(), 2
        if prec:
            rows = round_up(Cout, lib.caddy_k_hx_pick_bn(Cout))
            wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), -1, rows, 2), dtype=torch.uint8, device="cuda")
            lib.caddy_k_pack_hx(C.byref(d), C.c_void_p(wq.data_ptr()), rows, -1, prec, st)
            a.wq, a.precision = wq.data_ptr(), prec
        for _ in range(3):
            assert lib.caddy_k_conv_fwd(C.byref(a), st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.caddy_k_conv_fwd(C.byref(a), st)
        e1.record(); torch.cuda.synchronize()
        outs[prec] = (out, e0.elapsed_time(e1) / 20)
    err = ((outs[16][0] - outs[0][0]).abs().max() / outs[0][0].abs().max()).item()
    fl = 2.0 * N * H * W * 9 * Cin * Cout
    print(f"WIDE={os.environ.get('CADDY_HX_WIDE', '0')} {name:24s} f16x3 {outs[16][1] * 1e3:8.1f} us {fl / outs[16][1] / 1e9:6.1f} TF  max rel err vs exact fp32 {err:.2e}", flush=True)
