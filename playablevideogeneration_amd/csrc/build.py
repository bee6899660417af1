"""Build the HIP library in-tree:  python -m playablevideogeneration_amd.csrc.build  [--emu]

  default : hipcc --offload-arch=gfx950 -> playablevideogeneration_amd/csrc/libcaddy_hip.so   (the product)
  --emu   : host clang++ against tests/emu/hip/hip_runtime.h -> tests/emu/_build/libcaddy_emu.so  (TEST ONLY: the
            functional simulator used by `pytest -m "not gpu"`; never loaded by the package)
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["conv_mfma.hip", "conv_thin.hip", "conv_narrow.hip", "pointwise.hip", "pack.hip", "head.hip", "net.cpp", "capi_kernels.cpp", "capi.cpp"]
LIB = os.path.join(HERE, "libcaddy_hip.so")
EMU_LIB = os.path.join(ROOT, "tests", "emu", "_build", "libcaddy_emu.so")


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
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


def build_emu(force=False):
    srcs = [os.path.join(HERE, s) for s in SOURCES if os.path.exists(os.path.join(HERE, s))]
    emu_dir = os.path.join(ROOT, "tests", "emu")
    srcs.append(os.path.join(emu_dir, "emu_rt.cpp"))
    deps = _deps() + [os.path.join(emu_dir, "emu_rt.cpp"), os.path.join(emu_dir, "hip", "hip_runtime.h")]
    headers = [d for d in deps if d.endswith(".h")]
    if not force and not _stale(EMU_LIB, deps):
        return EMU_LIB
    cxx = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    flags = ["-O2", "-g", "-std=c++17", "-fPIC", "-Wno-psabi", "-Wno-unused-value", "-x", "c++",
             "-I", emu_dir, "-I", HERE, "-I", os.path.join(ROOT, "include")]
    objs = _compile_objects([cxx] + flags, os.path.dirname(EMU_LIB), srcs, headers)
    subprocess.check_call([cxx, "-shared", "-fPIC", "-o", EMU_LIB] + objs + ["-lpthread"])
    return EMU_LIB


if __name__ == "__main__":
    print(build_emu(True) if "--emu" in sys.argv else build(True))
