"""TEST ONLY: build + load the host functional-simulator build of the kernel sources (libcaddy_emu.so)."""
import ctypes as C

from tests.emu.build_emu import build_emu

_emu = None


def load_emu():
    """Simulator build.  `_caddy_device_type` tells Model() which torch device type the library's pointers live on (default: cuda)."""
    global _emu
    if _emu is None:
        _emu = C.CDLL(build_emu())
        _emu._caddy_device_type = "cpu"
    return _emu
