"""loops one conv_hx launch configuration for ~8 s (tools/probes/clock_probe.sh samples rocm-smi meanwhile)"""
import ctypes as C, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from playablevideogeneration_amd import _lib
from playablevideogeneration_amd._lib import ConvArgs, ConvSrc, PackDesc, round_up
lib = _lib.load(); lib.caddy_k_hx_weight_bytes.restype = C.c_long
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
prec = int(os.environ.get("BENCH_LOOP_PREC", "16")); planes = 1 if prec in (18, 19) else 2
N, H, W, Cin, Cout = 60, 32, 32, 512, 512
x = torch.randn(N, H, W, Cin, device="cuda"); w = (torch.randn(Cout, Cin, 3, 3, device="cuda") / (Cin * 9) ** 0.5).contiguous()
out = torch.empty(N, H, W, Cout, device="cuda")
d = PackDesc(); d.nw, d.Co_each, d.Cin, d.KS, d.nseg = 1, Cout, Cin, 3, 1
d.seg_off[0], d.seg_C[0], d.seg_Cpad[0] = 0, Cin, Cin
d.Cout, d.Cout_pad, d.Ktot = Cout, round_up(Cout, lib.caddy_k_conv_pick_bn(Cout)), Cin
d.w[0] = w.data_ptr()
wp = torch.zeros(9 * d.Cout_pad * d.Ktot, device="cuda"); lib.caddy_k_pack_fwd(C.byref(d), C.c_void_p(wp.data_ptr()), st)
a = ConvArgs(); a.src[0] = ConvSrc(x.data_ptr(), H * W * Cin, Cin, Cin, Cin, 0)
a.nsrc, a.N, a.H, a.W, a.KS, a.wp, a.Ktot, a.Cout, a.Cout_pad = 1, N, H, W, 3, wp.data_ptr(), d.Ktot, Cout, d.Cout_pad
a.out, a.out_sn, a.out_ld = out.data_ptr(), H * W * Cout, Cout
if prec:
    wq = torch.zeros(lib.caddy_k_hx_weight_bytes(C.byref(d), -1, 512, planes), dtype=torch.uint8, device="cuda")
    lib.caddy_k_pack_hx(C.byref(d), C.c_void_p(wq.data_ptr()), 512, -1, prec, st); a.wq, a.precision = wq.data_ptr(), prec
samples = []
def poll():
    for _ in range(6):
        time.sleep(1.0)
        r = subprocess.run("rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|Package Power'", shell=True, capture_output=True, text=True).stdout
        samples.append(" ".join(r.split()))
th = threading.Thread(target=poll); th.start()
torch.cuda.synchronize(); t0 = time.time(); n = 0
while time.time() - t0 < 8.0:
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): lib.caddy_k_conv_fwd(C.byref(a), st)
    e1.record(); torch.cuda.synchronize(); n += 1
    last = e0.elapsed_time(e1) / 50
th.join()
print(f"prec {prec}: {last * 1e3:.1f} us/launch steady state", flush=True)
for x in samples: print("   ", x)
