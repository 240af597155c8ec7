"""Host-side mirror of the reference's host program over the C-ABI.

Names follow the reference (safaad/aim host.c): a *device set* is what
`struct dpu_set_t` was, `push` are the host->device scatters
(host.c:246-268), `launch` is `dpu_launch(DPU_SYNCHRONOUS)` (host.c:289) and
`pull` the gathers (host.c:316-326).  Pairs are split over the devices of a
set in contiguous blocks exactly like host.c:191-209 splits them over DPUs.
"""
import ctypes as C
import math

import numpy as np

from . import capi
from .capi import (ALGO_BY_NAME, ALGO_NW, ALGO_SWG, ALGO_WFA, FLAG_BACKTRACE, FLAG_REDUCE, FLAG_REQ8, FLAG_RES8, FLAG_SWG_W16,
                   REQUEST8_DTYPE, REQUEST_DTYPE, RESULT8_DTYPE, RESULT_DTYPE, Params)


def round_up_8(x):
    return ((x + 7) // 8) * 8


def launcher_sizes(algo, read_length, error, mismatch=3, gap_o=4, gap_e=1, gap=4):
    """(MAX_SCORE, READ_SIZE) as the reference launchers derive them."""
    lib = capi.load()
    ms, rs = C.c_int32(), C.c_int32()
    a = ALGO_BY_NAME[algo] if isinstance(algo, str) else algo
    capi.check(lib.aim_launcher_sizes(a, read_length, float(error), mismatch, gap_o, gap_e, gap, C.byref(ms), C.byref(rs)))
    return ms.value, rs.value


def make_params(algo, max_score, read_size, match=0, mismatch=3, gap_o=4, gap_e=1, gap=4, backtrace=False,
                reduce=False, swg_w16=False, req8=False, res8=False, gap_i=None, gap_d=None):
    """`gap` is the launchers' single NW gap cost (run-nw-pim-wram.py: -DGAP_I = -DGAP_D); `gap_i` / `gap_d` set the two macros of
    nw.c:67-153 apart (NW/DPU-WRAM/common/common.h GAP_I, GAP_D)."""
    a = ALGO_BY_NAME[algo] if isinstance(algo, str) else algo
    gap_i = gap if gap_i is None else gap_i
    gap_d = gap if gap_d is None else gap_d
    flags = (FLAG_BACKTRACE if backtrace else 0) | (FLAG_REDUCE if reduce else 0) | (FLAG_SWG_W16 if swg_w16 else 0)
    flags |= (FLAG_REQ8 if req8 else 0) | (FLAG_RES8 if res8 else 0)
    return Params(a, match, mismatch, gap_o, gap_e, gap_i, gap_d, max_score, read_size, flags)


def params_for(algo, read_length, error, **kw):
    """Params from launcher-style (-l, -e) arguments."""
    cost = {k: kw[k] for k in ("mismatch", "gap_o", "gap_e", "gap") if k in kw}
    ms, rs = launcher_sizes(algo, read_length, error, **cost)
    return make_params(algo, ms, rs, **kw)


def gen_pairs(seed, first_idx, n_pairs, length, error, read_size):
    """Seeded synthetic pairs in the wire layout (requests, patterns[n][rs], texts[n][rs])."""
    lib = capi.load()
    req = np.zeros(n_pairs, dtype=REQUEST_DTYPE)
    pat = np.zeros((n_pairs, read_size), dtype=np.uint8)
    txt = np.zeros((n_pairs, read_size), dtype=np.uint8)
    capi.check(lib.aim_gen_pairs(seed, first_idx, n_pairs, length, float(error), read_size, capi.ptr(req),
                                 capi.ptr(pat), capi.ptr(txt)))
    return req, pat, txt


def to_request8(req):
    """aim_request_t[] -> aim_request8_t[] (the reference's own 8-byte WFA request_t; AIM_FLAG_REQ8)."""
    out = np.zeros(len(req), dtype=REQUEST8_DTYPE)
    out["pattern_len"], out["text_len"], out["idx"] = req["pattern_len"], req["text_len"], req["idx"]
    return out


def packed_row_dwords(read_size):
    return (read_size + 15) // 16


_CODE = np.full(256, 255, dtype=np.uint8)
for _c, _v in ((ord("A"), 0), (ord("C"), 1), (ord("T"), 2), (ord("G"), 3)):
    _CODE[_c] = _v


def pack_rows(req, rows, key):
    """2 bits per base (aim_hip.h, packed input): returns (packed[n][ceil(rs/16)] uint32, ok[n] bool); ok is False where a
    byte outside A/C/G/T lies inside the sequence (such pairs travel raw). Vectorised twin of aim_pack_sequence."""
    n, rs = rows.shape
    dw = packed_row_dwords(rs)
    lens = np.asarray(req[key], dtype=np.int64)
    inside = np.arange(rs)[None, :] < lens[:, None]
    codes = _CODE[rows]
    ok = ~((codes == 255) & inside).any(axis=1)
    c = np.where(inside & (codes != 255), codes, 0).astype(np.uint32)
    full = np.zeros((n, dw * 16), dtype=np.uint32)
    full[:, :rs] = c
    shifts = (2 * np.arange(16, dtype=np.uint32))[None, None, :]
    packed = (full.reshape(n, dw, 16) << shifts).sum(axis=2, dtype=np.uint64).astype(np.uint32)
    return np.ascontiguousarray(packed), ok


def pack_batch(req, pat, txt):
    """(packedP, packedT, raw_pairs, rawP, rawT) for aim_set_submit: pairs that cannot be packed go to the raw side list."""
    pp, okp = pack_rows(req, pat, "pattern_len")
    pt, okt = pack_rows(req, txt, "text_len")
    raw = np.nonzero(~(okp & okt))[0].astype(np.uint32)
    return pp, pt, raw, np.ascontiguousarray(pat[raw]), np.ascontiguousarray(txt[raw])


def pack_batch_native(params, req, pat, txt, threads=8):
    """aim_pack_batch (host threads in libaim_hip.so): same result as pack_batch, for batches too large for numpy temporaries."""
    lib = capi.load()
    n, rs = len(req), params.read_size
    dw = packed_row_dwords(rs)
    if (params.flags & FLAG_REQ8) and req.dtype != REQUEST8_DTYPE:
        req = to_request8(req)
    req, pat, txt = np.ascontiguousarray(req), np.ascontiguousarray(pat), np.ascontiguousarray(txt)
    pp, pt = np.zeros((n, dw), dtype=np.uint32), np.zeros((n, dw), dtype=np.uint32)
    cap = max(1, n // 8)
    raw, rawp, rawt = np.zeros(cap, dtype=np.uint32), np.zeros((cap, rs), dtype=np.uint8), np.zeros((cap, rs), dtype=np.uint8)
    nr = C.c_uint32()
    capi.check(lib.aim_pack_batch(C.byref(params), n, capi.ptr(req), capi.ptr(pat), capi.ptr(txt), capi.ptr(pp), capi.ptr(pt),
                                  capi.ptr(raw), capi.ptr(rawp), capi.ptr(rawt), cap, C.byref(nr), threads))
    k = nr.value
    return pp, pt, raw[:k].copy(), rawp[:k].copy(), rawt[:k].copy()


def format_output_runs(cig, runs):
    """Output file of the reference host (host.c:339-349) from the compact CIGAR (aim_cigar_t headers + run buffer)."""
    lib = capi.load()
    out = []
    buf = C.create_string_buffer(1 << 16)
    for i in range(len(cig)):
        out.append(b"%d, %d, \n" % (int(cig["idx"][i]), int(cig["score"][i])))
        nr, off = int(cig["n_runs"][i]), int(cig["run_offset"][i])
        r = np.ascontiguousarray(runs[off:off + nr])
        if len(buf) < 12 * nr + 16:
            buf = C.create_string_buffer(12 * nr + 16)
        n = capi.check(lib.aim_cigar_format_runs(capi.ptr(r), nr, buf, len(buf)))
        out.append(buf.raw[:n])
    return b"".join(out)


def pairs_to_text(req, pat, txt):
    """Render pairs in the reference input format ('>'pattern / '<'text lines)."""
    out = []
    for i in range(len(req)):
        out.append(b">" + pat[i, : req["pattern_len"][i]].tobytes() + b"\n")
        out.append(b"<" + txt[i, : req["text_len"][i]].tobytes() + b"\n")
    return b"".join(out)


def parse_pairs(data, read_size, max_pairs=None, first_idx=0):
    """get_reads (host.c:91-134): two lines per pair, first character dropped,
    length = line length - 2 (a final line without newline loses its last base)."""
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
        terminated = True
    else:
        terminated = False
    n = len(lines) // 2
    if max_pairs is not None:
        n = min(n, max_pairs)
    req = np.zeros(n, dtype=REQUEST_DTYPE)
    pat = np.zeros((n, read_size), dtype=np.uint8)
    txt = np.zeros((n, read_size), dtype=np.uint8)
    for i in range(n):
        for arr, key, j in ((pat, "pattern_len", 2 * i), (txt, "text_len", 2 * i + 1)):
            ln = lines[j]
            full = len(ln) + (1 if (terminated or j < len(lines) - 1) else 0)   # getline length incl. '\n'
            length = full - 2
            if length > read_size:
                raise ValueError("READ LENGTH less than length of the input reads")
            seq = ln[1 : 1 + length]
            arr[i, : len(seq)] = np.frombuffer(seq, dtype=np.uint8)
            req[key][i] = length
        req["idx"][i] = first_idx + i
    return req, pat, txt


def cigar_of(ops_row, begin_offset, end_offset):
    lib = capi.load()
    buf = C.create_string_buffer(int(4 * max(16, int(end_offset) - int(begin_offset)) + 32))
    n = capi.check(lib.aim_cigar_format(capi.ptr(ops_row), int(begin_offset), int(end_offset), buf, len(buf)))
    return buf.raw[:n]


def format_output(results, ops, backtrace):
    """Output file of the reference host (host.c:339-349): 'idx, score, \\n' [+ RLE CIGAR line]."""
    out = []
    for i in range(len(results)):
        out.append(b"%d, %d, \n" % (int(results["idx"][i]), int(results["score"][i])))
        if backtrace:
            out.append(cigar_of(ops[i], results["begin_offset"][i], results["end_offset"][i]))
    return b"".join(out)


class DeviceSet:
    """struct dpu_set_t counterpart: nr_devices GPUs, one stream each."""

    def __init__(self, nr_devices=1, device_ids=None):
        self.lib = capi.load()
        self.handle = C.c_void_p()
        ids = None
        if device_ids is not None:
            ids = (C.c_int * len(device_ids))(*device_ids)
            nr_devices = len(device_ids)
        capi.check(self.lib.aim_set_alloc(nr_devices, ids, C.byref(self.handle)))
        self.nr_devices = nr_devices
        self.params = None
        self.max_pairs = 0

    def close(self):
        if self.handle:
            self.lib.aim_set_free(self.handle)
            self.handle = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def configure(self, params, max_pairs_per_device):
        capi.check(self.lib.aim_set_configure(self.handle, C.byref(params), max_pairs_per_device))
        self.params = params
        self.max_pairs = max_pairs_per_device

    def configure_slots(self, params, max_pairs_per_device, slots=2, max_raw=0, max_runs=0):
        capi.check(self.lib.aim_set_configure_slots(self.handle, C.byref(params), max_pairs_per_device, slots, max_raw, max_runs))
        self.params = params
        self.max_pairs = max_pairs_per_device
        self._inflight = {}

    def submit(self, device, slot, req, pat=None, txt=None, packed=None, want_ops=False, cigar_runs_cap=0):
        """aim_set_submit: ASCII rows (pat, txt) or a packed batch (pack_batch(...)); results / ops / compact CIGAR buffers
        are allocated here and returned by wait()."""
        if (self.params.flags & FLAG_REQ8) and req.dtype != REQUEST8_DTYPE:
            req = to_request8(req)
        req = np.ascontiguousarray(req)
        n, rs = len(req), self.params.read_size
        io = capi.BatchIO()
        io.n_pairs = n
        keep = [req]
        io.requests = req.ctypes.data
        if packed is not None:
            pp, pt, raw, rawp, rawt = [np.ascontiguousarray(x) for x in packed]
            keep += [pp, pt, raw, rawp, rawt]
            io.packed_patterns, io.packed_texts = pp.ctypes.data, pt.ctypes.data
            io.n_raw = len(raw)
            if len(raw):
                io.raw_pairs, io.raw_patterns, io.raw_texts = raw.ctypes.data, rawp.ctypes.data, rawt.ctypes.data
        else:
            pat, txt = np.ascontiguousarray(pat), np.ascontiguousarray(txt)
            keep += [pat, txt]
            io.patterns, io.texts = pat.ctypes.data, txt.ctypes.data
        out = {}
        if cigar_runs_cap:
            out["cig"] = np.zeros(n, dtype=capi.CIGAR_DTYPE)
            out["runs"] = np.zeros(cigar_runs_cap, dtype=np.uint32)
            io.cigars, io.runs, io.runs_cap = out["cig"].ctypes.data, out["runs"].ctypes.data, cigar_runs_cap
        if not cigar_runs_cap or want_ops:
            out["res"] = np.zeros(n, dtype=RESULT8_DTYPE if (self.params.flags & FLAG_RES8) else RESULT_DTYPE)
            io.results = out["res"].ctypes.data
        if want_ops:
            out["ops"] = np.zeros((n, 2 * rs), dtype=np.uint8)
            io.ops = out["ops"].ctypes.data
        capi.check(self.lib.aim_set_submit(self.handle, device, slot, C.byref(io)))
        self._inflight[(device, slot)] = (io, keep, out)

    def wait(self, device, slot, check=True):
        io, keep, out = self._inflight.pop((device, slot), (None, None, {}))   # nothing in flight: the library reports AIM_ESTATE
        nr = C.c_uint32()
        rc = self.lib.aim_set_wait(self.handle, device, slot, C.byref(nr))
        if rc != capi.AIM_EALIGN or check:
            capi.check(rc)
        if "runs" in out:
            out["runs"] = out["runs"][: nr.value]
        return out

    def push(self, device, req, pat, txt):
        if (self.params.flags & FLAG_REQ8) and req.dtype != REQUEST8_DTYPE:
            req = to_request8(req)
        req = np.ascontiguousarray(req)
        pat = np.ascontiguousarray(pat)
        txt = np.ascontiguousarray(txt)
        self._keep = getattr(self, "_keep", {})
        self._keep[device] = (req, pat, txt)   # host buffers must outlive the async copies
        capi.check(self.lib.aim_set_push(self.handle, device, len(req), capi.ptr(req), capi.ptr(pat), capi.ptr(txt)))
        self._n = getattr(self, "_n", {})
        self._n[device] = len(req)

    def launch(self):
        capi.check(self.lib.aim_set_launch(self.handle))

    def pull(self, device, check=True):
        n = self._n[device]
        rs = self.params.read_size
        res = np.zeros(n, dtype=RESULT8_DTYPE if (self.params.flags & FLAG_RES8) else RESULT_DTYPE)
        ops = np.zeros((n, 2 * rs), dtype=np.uint8) if (self.params.flags & FLAG_BACKTRACE) else None
        rc = self.lib.aim_set_pull(self.handle, device, capi.ptr(res), capi.ptr(ops))
        if rc != capi.AIM_EALIGN or check:
            capi.check(rc)
        return res, ops

    def fallback_pairs(self, device=0):
        n = C.c_uint32()
        capi.check(self.lib.aim_set_fallback_pairs(self.handle, device, C.byref(n)))
        return n.value

    def plan_describe(self, device=0):
        """The plan line of the last launch on `device` (aim_set_plan_describe)."""
        buf = C.create_string_buffer(512)
        capi.check(self.lib.aim_set_plan_describe(self.handle, device, buf, len(buf)))
        return buf.value.decode()

    def timers(self):
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        capi.check(self.lib.aim_set_timers(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def align(self, params, req, pat, txt, check=True):
        """Whole batch over all devices of the set: contiguous blocks (host.c:191-209), results in input order."""
        n = len(req)
        per = max(1, math.ceil(n / self.nr_devices))
        if self.params is None or bytes(self.params) != bytes(params) or per > self.max_pairs:
            self.configure(params, per)
        blocks = []
        for d in range(self.nr_devices):
            lo, hi = min(n, d * per), min(n, (d + 1) * per)
            blocks.append((lo, hi))
            self.push(d, req[lo:hi], pat[lo:hi], txt[lo:hi])
        self.launch()
        parts = [self.pull(d, check=check) for d in range(self.nr_devices)]
        res = np.concatenate([p[0] for p in parts])
        ops = np.concatenate([p[1] for p in parts]) if parts[0][1] is not None else None
        return res, ops


def align(params, req, pat, txt, nr_devices=1, check=True):
    with DeviceSet(nr_devices) as s:
        return s.align(params, req, pat, txt, check=check)
