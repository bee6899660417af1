"""Build the HIP library in-tree:  python -m playablevideogeneration_amd.csrc.build

  hipcc --offload-arch=gfx950 -> playablevideogeneration_amd/csrc/libcaddy_hip.so   (the product; there is no other backend)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["conv_mfma.hip", "conv_hx.hip", "conv_hx_wgrad.hip", "conv_head.hip", "conv_direct.hip", "conv_thin.hip", "conv_narrow.hip", "conv_stream.hip", "pointwise.hip", "pack.hip", "head.hip", "perceptual.hip", "net.cpp", "capi_kernels.cpp", "dp_rccl.cpp"]
LIB = os.path.join(HERE, "libcaddy_hip.so")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps():
    return [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith((".hip", ".cpp", ".h"))] + \
           [os.path.join(ROOT, "include", "caddy_hip.h")]


def _compile_objects(cmd_prefix, objdir, srcs, headers):
    os.makedirs(objdir, exist_ok=True)
    objs, procs = [], []
    for src in srcs:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if _stale(obj, [src] + headers):
            procs.append((src, subprocess.Popen(cmd_prefix + ["-c", src, "-o", obj])))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError("compile failed: " + src)
    return objs


def build(force=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    headers = [d for d in _deps() if d.endswith(".h")]
    if not force and not _stale(LIB, _deps()):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-value", "-Wno-unused-result", "-x", "hip",
             "-I", HERE, "-I", os.path.join(ROOT, "include")]
    objs = _compile_objects([hipcc] + flags, os.path.join(HERE, "build"), srcs, headers)
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"])
    return LIB


if __name__ == "__main__":
    print(build(True))
