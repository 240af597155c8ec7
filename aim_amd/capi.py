"""ctypes binding of the C-ABI in include/aim_hip.h (libaim_hip.so).

This is the Python twin of the cgo/ctypes stub shown in INTEGRATION.md: plain
pointers and sizes only.  The library is built in-tree by aim_amd.build (hipcc,
gfx950); importing this module never compiles anything and never falls back to
a CPU implementation -- a missing library is an ImportError-like RuntimeError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AIM_LIB") or os.path.join(_HERE, "libaim_hip.so")   # AIM_LIB: A/B builds of the same ABI

AIM_OK, AIM_EINVAL, AIM_ENODEV, AIM_ENOMEM, AIM_ESTATE, AIM_EALIGN = 0, -1, -2, -3, -4, -5
ALGO_NW, ALGO_SWG, ALGO_WFA, ALGO_GENASM = 0, 1, 2, 3
ALGO_BY_NAME = {"nw": ALGO_NW, "swg": ALGO_SWG, "wfa": ALGO_WFA, "genasm": ALGO_GENASM}
FLAG_BACKTRACE, FLAG_REDUCE, FLAG_SWG_W16, FLAG_REQ8, FLAG_RES8 = 1, 2, 4, 8, 16
PAIR_OK, PAIR_WFA_NO_LINK, PAIR_SWG_NO_OP, PAIR_NOMEM = 0, 1, 2, 3


class Params(C.Structure):
    """aim_params_t"""
    _fields_ = [("algo", C.c_int32), ("match", C.c_int32), ("mismatch", C.c_int32), ("gap_o", C.c_int32),
                ("gap_e", C.c_int32), ("gap_i", C.c_int32), ("gap_d", C.c_int32), ("max_score", C.c_int32),
                ("read_size", C.c_int32), ("flags", C.c_uint32)]


REQUEST_DTYPE = np.dtype([("pattern_len", "<i4"), ("text_len", "<i4"), ("padding", "<i4"), ("idx", "<u4")])
RESULT_DTYPE = np.dtype([("max_operations", "<i4"), ("begin_offset", "<i4"), ("end_offset", "<i4"),
                         ("score", "<i4"), ("status", "<i4"), ("idx", "<u4")])
REQUEST8_DTYPE = np.dtype([("pattern_len", "<i2"), ("text_len", "<i2"), ("idx", "<u4")])     # AIM_FLAG_REQ8
RESULT8_DTYPE = np.dtype([("idx", "<u4"), ("score", "<i4")])                                 # AIM_FLAG_RES8
assert REQUEST_DTYPE.itemsize == 16 and RESULT_DTYPE.itemsize == 24
assert REQUEST8_DTYPE.itemsize == 8 and RESULT8_DTYPE.itemsize == 8

CIGAR_DTYPE = np.dtype([("idx", "<u4"), ("score", "<i4"), ("run_offset", "<u4"), ("n_runs", "<u2"), ("status", "<u2")])   # aim_cigar_t
assert CIGAR_DTYPE.itemsize == 16
CIGAR_OVERFLOW = 0x100


class BatchIO(C.Structure):
    """aim_batch_io_t"""
    _fields_ = [("n_pairs", C.c_uint32), ("requests", C.c_void_p), ("patterns", C.c_void_p), ("texts", C.c_void_p),
                ("packed_patterns", C.c_void_p), ("packed_texts", C.c_void_p), ("n_raw", C.c_uint32), ("raw_pairs", C.c_void_p),
                ("raw_patterns", C.c_void_p), ("raw_texts", C.c_void_p), ("results", C.c_void_p), ("ops", C.c_void_p),
                ("cigars", C.c_void_p), ("runs", C.c_void_p), ("runs_cap", C.c_uint32)]


# every symbol include/aim_hip.h declares: name -> (restype, argtypes)
_VP, _U32, _I32 = C.c_void_p, C.c_uint32, C.c_int32
SYMBOLS = {
    "aim_abi_version": (C.c_int, []),
    "aim_last_error": (C.c_char_p, []),
    "aim_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "aim_set_alloc": (C.c_int, [_U32, C.POINTER(C.c_int), C.POINTER(_VP)]),
    "aim_set_nr_devices": (C.c_int, [_VP, C.POINTER(_U32)]),
    "aim_set_configure": (C.c_int, [_VP, C.POINTER(Params), _U32]),
    "aim_set_push": (C.c_int, [_VP, _U32, _U32, _VP, _VP, _VP]),
    "aim_set_launch": (C.c_int, [_VP]),
    "aim_set_pull": (C.c_int, [_VP, _U32, _VP, _VP]),
    "aim_set_configure_slots": (C.c_int, [_VP, C.POINTER(Params), _U32, _U32, _U32, _U32]),
    "aim_set_submit": (C.c_int, [_VP, _U32, _U32, C.POINTER(BatchIO)]),
    "aim_set_wait": (C.c_int, [_VP, _U32, _U32, C.POINTER(_U32)]),
    "aim_pack_sequence": (C.c_int, [_VP, _I32, _I32, _VP]),
    "aim_pack_batch": (C.c_int, [C.POINTER(Params), _U32, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _U32, C.POINTER(_U32), C.c_int]),
    "aim_cigar_format_runs": (C.c_int, [_VP, _U32, _VP, _I32]),
    "aim_set_timers": (C.c_int, [_VP, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "aim_set_fallback_pairs": (C.c_int, [_VP, _U32, C.POINTER(_U32)]),
    "aim_set_plan_describe": (C.c_int, [_VP, _U32, C.c_char_p, C.c_size_t]),
    "aim_set_free": (C.c_int, [_VP]),
    "aim_host_alloc": (C.c_int, [C.POINTER(_VP), C.c_size_t]),
    "aim_host_free": (C.c_int, [_VP]),
    "aim_scratch_bytes": (C.c_size_t, [C.POINTER(Params), _U32]),
    "aim_align_device": (C.c_int, [C.POINTER(Params), _U32, _VP, _VP, _VP, _VP, _VP, _VP, C.c_size_t, _VP]),
    "aim_plan_describe": (C.c_int, [C.POINTER(Params), _U32, C.c_char_p, C.c_size_t]),
    "aim_kernel_name": (C.c_char_p, [C.POINTER(Params)]),
    "aim_launcher_sizes": (C.c_int, [_I32, _I32, C.c_double, _I32, _I32, _I32, _I32, C.POINTER(_I32), C.POINTER(_I32)]),
    "aim_cigar_format": (C.c_int, [_VP, _I32, _I32, _VP, _I32]),
    "aim_gen_pairs": (C.c_int, [C.c_uint64, C.c_uint64, _U32, _I32, C.c_double, _I32, _VP, _VP, _VP]),
}

_lib = None


class AimError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("aim_hip error %d: %s" % (code, message))
        self.code = code


def load():
    """Load libaim_hip.so (built by aim_amd.build).  Fails loudly when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("%s is missing: run `python -m aim_amd.build` (hipcc, gfx950). "
                           "There is no CPU fallback for the alignment path." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)   # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc < 0:
        raise AimError(rc, load().aim_last_error().decode(errors="replace"))
    return rc


def ptr(arr):
    """void* of a C-contiguous numpy array (or None)."""
    if arr is None:
        return None
    assert arr.flags["C_CONTIGUOUS"]
    return arr.ctypes.data_as(C.c_void_p)
