"""TEST ONLY: compile the UNMODIFIED kernel sources of playablevideogeneration_amd/csrc for the host against the functional simulator
(tests/emu/hip/hip_runtime.h + emu_rt.cpp) -> tests/emu/_build/libcaddy_emu.so.   python -m tests.emu.build_emu"""
import os
import subprocess

from playablevideogeneration_amd.csrc import build as B

EMU_DIR = os.path.dirname(os.path.abspath(__file__))
EMU_LIB = os.path.join(EMU_DIR, "_build", "libcaddy_emu.so")


def build_emu(force=False):
    srcs = [os.path.join(B.HERE, s) for s in B.SOURCES if os.path.exists(os.path.join(B.HERE, s))]
    srcs.append(os.path.join(EMU_DIR, "emu_rt.cpp"))
    deps = B._deps() + [os.path.join(EMU_DIR, "emu_rt.cpp"), os.path.join(EMU_DIR, "hip", "hip_runtime.h")]
    headers = [d for d in deps if d.endswith(".h")]
    if not force and not B._stale(EMU_LIB, deps):
        return EMU_LIB
    cxx = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
    flags = ["-O3", "-march=native", "-fno-math-errno", "-std=c++17", "-fPIC", "-Wno-psabi", "-Wno-unused-value", "-x", "c++",
             "-I", EMU_DIR, "-I", B.HERE, "-I", os.path.join(B.ROOT, "include")]
    objs = B._compile_objects([cxx] + flags, os.path.dirname(EMU_LIB), srcs, headers)
    subprocess.check_call([cxx, "-shared", "-fPIC", "-o", EMU_LIB] + objs + ["-lpthread"])
    return EMU_LIB


FAKE_RCCL = os.path.join(EMU_DIR, "_build", "libfake_rccl.so")


def build_fake_rccl():
    """the shared-memory stand-in for librccl.so (tests/emu/fake_rccl.cpp) that the simulator build's dp_rccl.cpp loads through CADDY_RCCL_LIB"""
    src = os.path.join(EMU_DIR, "fake_rccl.cpp")
    if B._stale(FAKE_RCCL, [src]):
        os.makedirs(os.path.dirname(FAKE_RCCL), exist_ok=True)
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", FAKE_RCCL, src, "-lrt", "-lpthread"])
    return FAKE_RCCL


if __name__ == "__main__":
    print(build_emu(True))
