"""PROBE ONLY (never shipped, never loaded by the package): builds tools/_ab/libcaddy_hip_probe{,_noconv}.so from PATCHED COPIES of the kernel sources so that whole
kernel families can be switched off (garbage numerics, valid timing) -- the upper bound of what fusing / removing them could buy inside the real step:

  PROBE_SKIP_BN_REDUCE=1   lazily applied BatchNorm chains: no k_reduce<1> + k_sum_partials in the backward
  PROBE_SKIP_BN_APPLY=1    ... and no FBnBwdApply either
  PROBE_SKIP_NARROW_WGRAD=1 weight gradients that do not run on k_wgrad_hx are not launched
  PROBE_SKIP_HX_WGRAD=1    k_wgrad_hx launches are not launched
  libcaddy_hip_probe_noconv.so: k_conv_hx / k_wgrad_hx stage their operands WITHOUT the fp32 -> split 16-bit conversion (a 16-byte copy per float4: what
                                pre-split operands written by the producer would leave in the loaders)

    python tools/probes/build_sensitivity_probe.py && CADDY_HIP_LIB=tools/_ab/libcaddy_hip_probe.so PROBE_SKIP_BN_REDUCE=1 python bench.py ...
"""
import os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from playablevideogeneration_amd.csrc import build as B

def patch(text, old, new, count=1):
    assert text.count(old) >= 1, old[:80]
    return text.replace(old, new, count)

def make(tag, noconv):
    src = os.path.join("/tmp", "caddy_probe_" + tag)
    shutil.rmtree(src, ignore_errors=True)
    shutil.copytree(B.HERE, src, ignore=shutil.ignore_patterns("*.so", "build*", "__pycache__"))
    p = os.path.join(src, "net.cpp"); t = open(p).read()
    t = patch(t, "static thread_local std::string g_err;", "static thread_local std::string g_err;\nstatic bool probe(const char* n) { const char* e = getenv(n); return e && atoi(e) != 0; }")
    t = patch(t, "RUN(pw_bn_bwd_reduce(gv(out), nullptr, dv(x), s1.mean, s1.invstd, s1.sums, red_scratch, bn_dgamma(b1), bn_dbeta(b1), stream, ls, s1.shift));",
              "if (!probe(\"PROBE_SKIP_BN_REDUCE\")) RUN(pw_bn_bwd_reduce(gv(out), nullptr, dv(x), s1.mean, s1.invstd, s1.sums, red_scratch, bn_dgamma(b1), bn_dbeta(b1), stream, ls, s1.shift));")
    t = patch(t, "RUN(pw_bn_bwd_apply(gv(out), nullptr, dv(x), s1.mean, s1.invstd, b1->gamma, s1.sums, gv(x), nullptr, nullptr, x.nz ? 1 : 0, stream, ls, s1.shift));",
              "if (!probe(\"PROBE_SKIP_BN_APPLY\")) RUN(pw_bn_bwd_apply(gv(out), nullptr, dv(x), s1.mean, s1.invstd, b1->gamma, s1.sums, gv(x), nullptr, nullptr, x.nz ? 1 : 0, stream, ls, s1.shift));")
    open(p, "w").write(t)
    p = os.path.join(src, "conv_mfma.hip"); t = open(p).read()
    t = patch(t, "    if (conv_hx_wgrad_try(a, st, dry) == 1) return 0;       // wide 3x3",
              "    { static const bool sh = getenv(\"PROBE_SKIP_HX_WGRAD\") && atoi(getenv(\"PROBE_SKIP_HX_WGRAD\")), sn = getenv(\"PROBE_SKIP_NARROW_WGRAD\") && atoi(getenv(\"PROBE_SKIP_NARROW_WGRAD\"));\n"
              "      const bool is_hx = conv_hx_wgrad_try(a, st, true) == 1; if (!dry && ((is_hx && sh) || (!is_hx && sn))) return 0; }\n"
              "    if (conv_hx_wgrad_try(a, st, dry) == 1) return 0;       // wide 3x3")
    if "#include <cstdlib>" not in t: t = "#include <cstdlib>\n" + t
    open(p, "w").write(t)
    if noconv:
        p = os.path.join(src, "conv_hx.hip"); t = open(p).read()
        # k_conv_hx: raw 16-byte copy of the float4 into the LDS row (zero padding kept)
        a = t.index("                float4 v_ = ra[i];                                                                                                 \\")
        b = t.index("                if (NPL == 2) *reinterpret_cast<v4*>(&As[aoff[i] + KC]) = lo_;")
        b = t.index("\n", b) + 1
        t = t[:a] + ("                float4 v_ = ra[i]; v_.x = ok_ ? v_.x : 0.f; v_.y = ok_ ? v_.y : 0.f; v_.z = ok_ ? v_.z : 0.f; v_.w = ok_ ? v_.w : 0.f; (void)m1_; (void)m2_; (void)m3_; (void)bn_; (void)sl_; \\\n"
                     "                *reinterpret_cast<float4*>(&As[aoff[i] + 4 * q]) = v_;                                       \\\n") + t[b:]
        # k_wgrad_hx: raw copies as well
        a = t.index("#define WG_SPLIT_STORE(dst_, v_)")
        b = t.index("#define WG_X_ELEM(i_, ok_, dst_)")
        t = t[:a] + ("#define WG_SPLIT_STORE(dst_, v_) do { *reinterpret_cast<float4*>((dst_) + 4 * (q & 7) + (q >> 3) * 32) = (v_); } while (0)\n") + t[b:]
        open(p, "w").write(t)
    srcs = [os.path.join(src, s) for s in B.SOURCES]
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value", "-Wno-unused-result", "-x", "hip", "-I", src, "-I", os.path.join(ROOT, "include")]
    hdr = [os.path.join(src, f) for f in os.listdir(src) if f.endswith(".h")]
    objs = B._compile_objects(["/opt/rocm/bin/hipcc"] + flags, os.path.join(src, "build"), srcs, hdr)
    out = os.path.join(ROOT, "tools", "_ab", "libcaddy_hip_probe" + ("_noconv" if noconv else "") + ".so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    print(out)

if __name__ == "__main__":
    make("skip", False)
    make("noconv", True)
