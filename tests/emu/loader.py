"""TEST ONLY: build + load the host functional-simulator build of the kernel sources (libcaddy_emu.so)."""
import ctypes as C

from playablevideogeneration_amd.csrc import build as B

_emu = None


def load_emu():
    """Simulator build; flagged so Model() accepts CPU tensors in tests."""
    global _emu
    if _emu is None:
        _emu = C.CDLL(B.build_emu())
        _emu._caddy_emulated = True
    return _emu
