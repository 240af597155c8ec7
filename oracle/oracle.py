"""ctypes binding of the CPU oracle (oracle/aim_oracle.c).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by aim_amd/."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libaim_oracle.so")
ALGO = {"nw": 0, "swg": 1, "wfa": 2, "genasm": 3}


class OrcParams(C.Structure):
    _fields_ = [(n, C.c_int) for n in ("algo", "match", "mismatch", "gap_o", "gap_e", "gap_i", "gap_d", "max_score",
                                       "read_size", "backtrace", "reduce", "swg_cell_bytes")]


ORC_RESULT_DTYPE = np.dtype([("max_operations", "<i4"), ("begin_offset", "<i4"), ("end_offset", "<i4"),
                             ("score", "<i4"), ("idx", "<u4"), ("status", "<i4")])
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", HERE, "libaim_oracle.so", "oracle_cli"])


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        lib = C.CDLL(LIB_PATH)
        lib.orc_align_batch.restype = C.c_int
        lib.orc_align_batch.argtypes = [C.POINTER(OrcParams), C.c_uint32] + [C.c_void_p] * 6 + [C.c_int]
        lib.orc_cigar_format.restype = C.c_int
        lib.orc_cigar_format.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.orc_launcher_sizes.restype = None
        lib.orc_launcher_sizes.argtypes = [C.c_int, C.c_int, C.c_double] + [C.c_int] * 4 + [C.POINTER(C.c_int)] * 2
        _lib = lib
    return _lib


def params(algo, max_score, read_size, match=0, mismatch=3, gap_o=4, gap_e=1, gap=4, backtrace=False, reduce=False,
           swg_cell_bytes=0, gap_i=None, gap_d=None):
    gap_i = gap if gap_i is None else gap_i
    gap_d = gap if gap_d is None else gap_d
    return OrcParams(ALGO[algo], match, mismatch, gap_o, gap_e, gap_i, gap_d, max_score, read_size, int(backtrace),
                     int(reduce), swg_cell_bytes)


def launcher_sizes(algo, l, e, mismatch=3, gap_o=4, gap_e=1, gap=4):
    ms, rs = C.c_int(), C.c_int()
    load().orc_launcher_sizes(ALGO[algo], l, float(e), mismatch, gap_o, gap_e, gap, C.byref(ms), C.byref(rs))
    return ms.value, rs.value


def align_batch(p, plen, tlen, patterns, texts, nthreads=1):
    """patterns/texts: uint8 [n][read_size].  Returns (results, ops or None, worst_status)."""
    lib = load()
    n = len(plen)
    plen = np.ascontiguousarray(plen, dtype=np.int32)
    tlen = np.ascontiguousarray(tlen, dtype=np.int32)
    patterns = np.ascontiguousarray(patterns, dtype=np.uint8)
    texts = np.ascontiguousarray(texts, dtype=np.uint8)
    res = np.zeros(n, dtype=ORC_RESULT_DTYPE)
    ops = np.zeros((n, 2 * p.read_size), dtype=np.uint8) if p.backtrace else None
    vp = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
    worst = lib.orc_align_batch(C.byref(p), n, vp(plen), vp(tlen), vp(patterns), vp(texts), vp(res), vp(ops), nthreads)
    return res, ops, worst


def cigar_of(ops_row, begin, end):
    buf = C.create_string_buffer(int(4 * max(16, int(end) - int(begin)) + 32))
    n = load().orc_cigar_format(ops_row.ctypes.data_as(C.c_void_p), int(begin), int(end), buf, len(buf))
    assert n >= 0
    return buf.raw[:n]


def format_output(res, ops, backtrace):
    out = []
    for i in range(len(res)):
        out.append(b"%d, %d, \n" % (int(res["idx"][i]), int(res["score"][i])))
        if backtrace:
            out.append(cigar_of(ops[i], res["begin_offset"][i], res["end_offset"][i]))
    return b"".join(out)
