"""ctypes binding of the C-ABI of libcaddy_hip.so (include/caddy_hip.h).

The product path has NO CPU fallback: `load()` raises if the HIP library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `python -m playablevideogeneration_amd.csrc.build`).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "csrc", "libcaddy_hip.so")
CONV_MAX_SRC = 3
CONV_BK = 16


class TV(C.Structure):
    _fields_ = [("p", C.c_void_p), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("C", C.c_int), ("sn", C.c_long), ("ld", C.c_int), ("s16", C.c_int)]      # s16: csrc/common.h (pre-split gradient)


class ConvSrc(C.Structure):
    _fields_ = [("p", C.c_void_p), ("sn", C.c_long), ("ld", C.c_int), ("C", C.c_int), ("Cpad", C.c_int), ("bcast", C.c_int),
                ("bn_scale", C.c_void_p), ("bn_shift", C.c_void_p), ("bn_act", C.c_int), ("bn_gn", C.c_int), ("bn_gs", C.c_long)]      # lazily applied BatchNorm of the producer (common.h)


class ConvArgs(C.Structure):
    _fields_ = [("src", ConvSrc * CONV_MAX_SRC), ("nsrc", C.c_int), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("KS", C.c_int),
                ("wp", C.c_void_p), ("Ktot", C.c_int), ("Cout", C.c_int), ("Cout_pad", C.c_int), ("bias", C.c_void_p), ("act", C.c_int),
                ("out", C.c_void_p), ("out_sn", C.c_long), ("out_ld", C.c_int), ("accumulate", C.c_int), ("precision", C.c_int), ("splitk", C.c_int), ("aux", C.c_void_p), ("split_scratch", C.c_void_p), ("split_cap", C.c_long), ("split_stride", C.c_long),
                ("mask", C.c_void_p), ("seed_ref", C.c_void_p), ("seed_w", C.c_float), ("wq", C.c_void_p), ("Kq", C.c_int), ("out_scale", C.c_float),
                ("res", C.c_void_p), ("res_sn", C.c_long), ("res_ld", C.c_int), ("xcd_map", C.c_int),
                ("pool_out", C.c_void_p), ("pool_sn", C.c_long), ("pool_ld", C.c_int), ("skip_out", C.c_int),
                ("stats", C.c_void_p), ("stats_ld", C.c_int), ("sat_flag", C.c_void_p), ("deterministic", C.c_int), ("direct_ok", C.c_int), ("lstm", C.c_void_p),
                ("in_s16", C.c_int), ("out_s16", C.c_int), ("pool_s16", C.c_int), ("mask_s16", C.c_int), ("seed_s16", C.c_int),
                ("avgpool", C.c_int), ("sat_out_next", C.c_int)]      # S16 tensors (csrc/common.h): pre-split 16-bit operand pairs


class LstmFuse(C.Structure):      # csrc/common.h: cell update applied by the slab reduce of a roll-out gate convolution (ConvArgs.lstm)
    _fields_ = [("cprev", C.c_void_p), ("cprev_sn", C.c_long), ("cprev_ld", C.c_int), ("h", C.c_void_p), ("h_sn", C.c_long), ("h_ld", C.c_int),
                ("c", C.c_void_p), ("c_sn", C.c_long), ("c_ld", C.c_int), ("hb", C.c_void_p), ("hb_sn", C.c_long), ("hb_ld", C.c_int),
                ("scale", C.c_void_p), ("shift", C.c_void_p), ("C", C.c_int)]


class WgradArgs(C.Structure):
    _fields_ = [("src", ConvSrc * CONV_MAX_SRC), ("nsrc", C.c_int), ("N", C.c_int), ("H", C.c_int), ("W", C.c_int), ("KS", C.c_int),
                ("dy", C.c_void_p), ("dy_sn", C.c_long), ("dy_ld", C.c_int), ("Cout", C.c_int), ("Cout_pad", C.c_int), ("Ktot", C.c_int),
                ("dwp", C.c_void_p), ("slabs", C.c_int), ("group_n", C.c_int), ("src_gs", C.c_long * CONV_MAX_SRC), ("dy_gs", C.c_long), ("precision", C.c_int),
                ("src_bn_gs", C.c_long * CONV_MAX_SRC), ("det_slab", C.c_void_p), ("det_cap", C.c_long), ("det_stride", C.c_long), ("dy_s16", C.c_int)]


class PackDesc(C.Structure):
    _fields_ = [("w", C.c_void_p * 4), ("gw", C.c_void_p * 4), ("nw", C.c_int), ("Co_each", C.c_int), ("Cin", C.c_int), ("KS", C.c_int),
                ("nseg", C.c_int), ("seg_off", C.c_int * CONV_MAX_SRC), ("seg_C", C.c_int * CONV_MAX_SRC), ("seg_Cpad", C.c_int * CONV_MAX_SRC),
                ("Cout", C.c_int), ("Cout_pad", C.c_int), ("Ktot", C.c_int), ("oscale", C.c_void_p)]


def round_up(a, b):
    return (a + b - 1) // b * b


_lib = None


def load(path=None):
    """Load the HIP shared library (cached).  Raises RuntimeError when it is missing -- there is no fallback."""
    global _lib
    if path is None and _lib is not None:
        return _lib
    p = path or os.environ.get("CADDY_HIP_LIB") or LIB_PATH      # (CADDY_HIP_LIB: timing experiments with an alternative build of the same library)
    if not os.path.exists(p):
        raise RuntimeError(f"{p} not found: the HIP extension is not built (run __graft_entry__.build()); "
                           "playablevideogeneration_amd has no CPU fallback")
    lib = C.CDLL(p)
    if path is None:
        _lib = lib
    return lib
