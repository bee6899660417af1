"""TEST ONLY: build + load the host functional-simulator build of the kernel sources (libcaddy_emu.so)."""
import ctypes as C
import os

from tests.emu.build_emu import build_emu

_emu = None


def load_emu():
    """Simulator build.  `_caddy_device_type` tells Model() which torch device type the library's pointers live on (default: cuda)."""
    global _emu
    if _emu is None:
        # simulator runs use the exact-fp32 convolution path unless a test asks for the split-operand kernels explicitly (4x slower to simulate)
        os.environ.setdefault("CADDY_PRECISION", "exact")
        _emu = C.CDLL(build_emu())
        _emu._caddy_device_type = "cpu"
    return _emu
