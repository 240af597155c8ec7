// wfa_lane_packed.hpp -- the short-read WFA fast path (one pair per lane, wfa_lane.hpp) on the PACKED wire format, with the
// compact CIGAR emitted by the kernel itself: one kernel per batch on the drop-in path.
//
// The reference ships ASCII rows to the device and gathers result_t + 2*READ_SIZE op bytes per pair
// (WFA/DPU-WRAM/host/host.c:258-268, 316-326) which edit_cigar_print then run-length encodes on the host (host.c:69-89).
// The default ABI keeps exactly that. This kernel is what aim_set_submit runs when the batch arrives packed (2 bits per
// base, aim_hip.h) and, with BACKTRACE, the caller asked for the compact CIGAR:
//   * input: each lane loads ITS pair's request and its two packed rows (ceil(READ_SIZE/16) dwords each) straight from
//     HBM into registers, one group ahead of the compute -- 72 instead of 232 bytes per pair at READ_SIZE 112, no LDS
//     staging, no validate/pack phase (40 % of the ASCII kernel's time), and no constraint on READ_SIZE: the row stride
//     rule of the ASCII kernel (odd number of 16-B LDS slots) does not exist here;
//   * WFA: the same statically unrolled score loop (wfa_scores_static below = the loop of wfa_lane_kernel as a function) or
//     the dynamic-bounds loop (wfa_scores_dynamic) over bit-parallel mismatch vectors;
//   * output, score-only: {idx, score} or the 24-B result_t; with BACKTRACE: affine_wavefronts_backtrace
//     (wfa_backtracing.c:210-351) walks the register-resident history and hands its operations to a RUN COLLECTOR instead of
//     an ops row pre-filled with 'M' (wfa.c:465): what leaves the kernel is one aim_cigar_t (16 B) and the (length << 8) | op
//     runs of the pair -- exactly what cigar_rle_kernel (batch_io.hpp) would have produced from the ops row, including its
//     clamping of begin_offset / end_offset.
// Pairs with a byte outside A/C/G/T cannot be packed: they travel in the batch's raw side list and are aligned by the
// ASCII kernels in a second, small launch whose results overwrite these (aim_capi.hip, raw side pass).
#pragma once

#include "aim_device.hpp"
#include "wfa_lane.hpp"

#ifndef AIM_LANEPK_WGS_PER_CU
#define AIM_LANEPK_WGS_PER_CU 16      // persistent single-wave workgroups per CU of the static shapes (<= 128 VGPRs: 4 per SIMD)
#endif
#ifndef AIM_LANEPK_DYN_WGS_PER_CU
#define AIM_LANEPK_DYN_WGS_PER_CU 8   // the dynamic-bounds shape holds its 13-diagonal history in registers (2 waves per SIMD)
#endif

namespace aim {

// ---- the score loop of wfa_lane_kernel as a function (affine_wfa_compute, wfa.c:342-379, statically unrolled) --------
template <int X, int O, int E, int MAXS, int NP, int KW>
__device__ __forceinline__ void wfa_scores_static(const uint32_t (&dk)[KW][NP], int plen, int tlen, int ms_run, bool active,
                                                  int (&Mv)[MAXS + 1][KW], int (&Iv)[MAXS + 1][KW], int (&Dv)[MAXS + 1][KW],
                                                  int &score, bool &done)
{
    constexpr WfShape<X, O, E, MAXS> SH{};
    const int ak = tlen - plen;
    score = MAXS + 1;
    done = false;
#pragma unroll
    for (int s = 0; s <= MAXS; ++s) {
        if (!SH.present[s]) continue;
        if (s > 0) {
            const int ss = s - X, so = s - O - E, se = s - E;
            const bool sub_ok = ss >= 0 && SH.present[ss];
            const bool o_ok = so >= 0 && SH.present[so];
            const bool ie_ok = se >= 0 && SH.present[se] && SH.hasI[se];
            const bool de_ok = se >= 0 && SH.present[se] && SH.hasD[se];
#pragma unroll
            for (int k = SH.lo[s]; k <= SH.hi[s]; ++k) {   // affine_wfa_compute_offsets, wfa.c:231-266
                const int kk = k - SH.kmin;
                const int km1 = kk > 0 ? kk - 1 : 0, kp1 = kk + 1 < KW ? kk + 1 : KW - 1;   // clamped: only read when in range
                int ins = -10;
                if (SH.hasI[s]) {
                    const int ins_g = (o_ok && SH.lo[so] <= k - 1 && k - 1 <= SH.hi[so]) ? Mv[so][km1] : kLaneNull;
                    const int ins_i = (ie_ok && SH.lo[se] <= k - 1 && k - 1 <= SH.hi[se]) ? Iv[se][km1] : kLaneNull;
                    ins = (ins_g == kLaneNull && ins_i == kLaneNull) ? kLaneNull : max(ins_g, ins_i) + 1;
                    Iv[s][kk] = ins;
                }
                int del = -10;
                if (SH.hasD[s]) {
                    const int del_g = (o_ok && SH.lo[so] <= k + 1 && k + 1 <= SH.hi[so]) ? Mv[so][kp1] : kLaneNull;
                    const int del_d = (de_ok && SH.lo[se] <= k + 1 && k + 1 <= SH.hi[se]) ? Dv[se][kp1] : kLaneNull;
                    del = max(del_g, del_d);
                    Dv[s][kk] = del;
                }
                int sub = -10;
                if (sub_ok) sub = (SH.lo[ss] <= k && k <= SH.hi[ss]) ? Mv[ss][kk] + 1 : kLaneNull;
                Mv[s][kk] = max(del, max(sub, ins));
            }
        } else {
            Mv[0][-SH.kmin] = 0;
        }
        // affine_wfa_extend (wfa.c:186-208), bit-parallel; then affine_wfa_end_reached (wfa.c:210-230); the run-time MAX_SCORE
        // cap is a term of the end test (wavefronts past it are computed and ignored: wfa.c:368-376)
        int m_end = kLaneNull;
        bool end_in_range = false;
#pragma unroll
        for (int k = SH.lo[s]; k <= SH.hi[s]; ++k) {
            const int kk = k - SH.kmin;
            int off = Mv[s][kk];
            const int v = off - k;
            const int limit = min(plen, tlen - k);
            if (off >= 0 && v >= 0 && v < limit) {
                const int stop = min(first_stop<NP>(dk[kk], v), limit);
                off += stop - v;
            }
            Mv[s][kk] = off;
            if (k == ak) { m_end = off; end_in_range = true; }
        }
        if (!done && end_in_range && m_end >= tlen && s <= ms_run) { done = true; score = s; }
        if (__ballot(!done && active) == 0ull) break;   // every pair of this wave has finished
    }
    if (!done) score = ms_run + 1;                      // wfa.c:368-376
}

// ---- affine_wavefronts_backtrace (wfa_backtracing.c:210-351) over the register-resident history ----------------------
// The fetchers (wfa_backtracing.c:73-172) are static select chains; kNone marks "no such cell" (score < 0, wavefronts[s] ==
// NULL, d_null / iwavefront == NULL, k outside [klo, khi]). Operations go to `sink`: put(ch) is the reference's
// operations[begin_offset--] = ch, matches(n) its begin_offset -= n over the 'M' pre-fill. Returns the AIM_PAIR_* status.
template <int X, int O, int E, int MAXS, int KW, typename Sink>
__device__ __forceinline__ int wfa_backtrace_static(const int (&Mv)[MAXS + 1][KW], const int (&Iv)[MAXS + 1][KW], const int (&Dv)[MAXS + 1][KW],
                                                    int score, int plen, int tlen, Sink &sink)
{
    constexpr WfShape<X, O, E, MAXS> SH{};
    constexpr int kNone = (int)0x80000000;
    const int ak = tlen - plen;
    auto getM = [&](int s_, int k_) {
        int r = kNone;
#pragma unroll
        for (int s2 = 0; s2 <= MAXS; ++s2)
            if (SH.present[s2])
#pragma unroll
                for (int k2 = SH.lo[s2]; k2 <= SH.hi[s2]; ++k2) r = (s_ == s2 && k_ == k2) ? Mv[s2][k2 - SH.kmin] : r;
        return r;
    };
    auto getI = [&](int s_, int k_) {
        int r = kNone;
#pragma unroll
        for (int s2 = 0; s2 <= MAXS; ++s2)
            if (SH.present[s2] && SH.hasI[s2])
#pragma unroll
                for (int k2 = SH.lo[s2]; k2 <= SH.hi[s2]; ++k2) r = (s_ == s2 && k_ == k2) ? Iv[s2][k2 - SH.kmin] : r;
        return r;
    };
    auto getD = [&](int s_, int k_) {
        int r = kNone;
#pragma unroll
        for (int s2 = 0; s2 <= MAXS; ++s2)
            if (SH.present[s2] && SH.hasD[s2])
#pragma unroll
                for (int k2 = SH.lo[s2]; k2 <= SH.hi[s2]; ++k2) r = (s_ == s2 && k_ == k2) ? Dv[s2][k2 - SH.kmin] : r;
        return r;
    };
    auto valid_loc = [&](int kk_, int off_) {
        const int v_ = off_ - kk_, h_ = off_;
        return v_ > 0 && v_ <= plen && h_ > 0 && h_ <= tlen;
    };
    enum { BT_M = 0, BT_I = 1, BT_D = 2 };
    int status = AIM_PAIR_OK;
    int sc = score, k = ak;
    int offset = getM(sc, k);
    bool valid = valid_loc(k, offset);
    int bt = BT_M;
    int v = offset - k, h = offset;
    while (v > 0 && h > 0 && sc > 0) {
        if (!valid) {
            valid = valid_loc(k, offset);
            if (valid) {   // add_trailing_gap, wfa_backtracing.c:48-69
                if (k < ak) for (int i = k; i < ak; ++i) sink.put('I');
                else if (k > ak) for (int i = ak; i < k; ++i) sink.put('D');
            }
        }
        const int s_o = sc - (O + E), s_e = sc - E, s_x = sc - X;
        int del_ext = kLaneNull, del_open = kLaneNull, ins_ext = kLaneNull, ins_open = kLaneNull, misms = kLaneNull;
        if (bt != BT_I) {
            const int a1 = getD(s_e, k + 1), a2 = getM(s_o, k + 1);
            if (a1 != kNone) del_ext = a1;
            if (a2 != kNone) del_open = a2;
        }
        if (bt != BT_D) {
            const int a1 = getI(s_e, k - 1), a2 = getM(s_o, k - 1);
            if (a1 != kNone) ins_ext = a1 + 1;
            if (a2 != kNone) ins_open = a2 + 1;
        }
        if (bt == BT_M) {
            const int a1 = getM(s_x, k);
            if (a1 != kNone) misms = a1 + 1;
        }
        const int max_all = max(misms, max(max(ins_ext, ins_open), max(del_ext, del_open)));
        if (bt == BT_M) {
            const int num_matches = offset - max_all;
            if (num_matches > 0) sink.matches(num_matches);
            offset = max_all;
            v = offset - k;
            h = offset;
            if (v <= 0 || h <= 0) break;
        }
        char op;
        if (max_all == del_ext) { op = 'D'; sc = s_e; ++k; bt = BT_D; }
        else if (max_all == del_open) { op = 'D'; sc = s_o; ++k; bt = BT_M; }
        else if (max_all == ins_ext) { op = 'I'; sc = s_e; --k; --offset; bt = BT_I; }
        else if (max_all == ins_open) { op = 'I'; sc = s_o; --k; --offset; bt = BT_M; }
        else if (max_all == misms) { op = 'X'; sc = s_x; --offset; }
        else { status = AIM_PAIR_WFA_NO_LINK; break; }
        if (valid) sink.put(op);
        v = offset - k;
        h = offset;
    }
    if (status == AIM_PAIR_OK) {
        if (sc == 0) {
            if (offset > 0) sink.matches(offset);
        } else {
            for (; v > 0; --v) sink.put('D');
            for (; h > 0; --h) sink.put('I');
        }
    }
    return status;
}

// ---- the same walk for the dynamic-bounds shapes (MAX_SCORE 6..10): history in the lane's LDS column ------------------
// wfa_scores_dynamic<HIST> left every cell at a compile-time index (WfHist) and every score's final klo / khi / flags behind
// them; the walk reads them with the range and null tests of the reference's fetchers (wfa_backtracing.c:73-172; the same
// tests as wfa_group_tb_kernel: del_ext wants !d_null, ins_ext wants the I component to EXIST -- W7's asymmetry).
template <int X, int O, int E, int MAXS, typename Sink>
__device__ __forceinline__ int wfa_backtrace_dynamic(const int16_t *hist, int score, int plen, int tlen, Sink &sink)
{
    constexpr WfShape<X, O, E, MAXS> SH{};
    constexpr WfHist<X, O, E, MAXS> HX{};
    const int ak = tlen - plen;
    // static row tables, looked up by a per-lane score (select chains over <= MAXS + 1 entries)
    auto lut = [&](const int (&arr)[MAXS + 1], int s_) { int r = 0;
#pragma unroll
        for (int s2 = 0; s2 <= MAXS; ++s2) r = (s_ == s2) ? arr[s2] : r;
        return r; };
    struct Row { int klo, khi, f, lo; };
    auto row = [&](int s_) {
        Row r;
        const int sc_ = s_ < 0 ? 0 : s_;
        r.klo = hist[(HX.meta + 3 * sc_) * kWave]; r.khi = hist[(HX.meta + 3 * sc_ + 1) * kWave]; r.f = hist[(HX.meta + 3 * sc_ + 2) * kWave];
        r.lo = lut(SH.lo, sc_);
        if (s_ < 0) { r.klo = 1; r.khi = -1; r.f = 0; }
        return r;
    };
    auto in = [&](const Row &r, int k_) { return r.klo <= k_ && k_ <= r.khi; };
    auto valid_loc = [&](int kk_, int off_) {
        const int v_ = off_ - kk_, h_ = off_;
        return v_ > 0 && v_ <= plen && h_ > 0 && h_ <= tlen;
    };
    enum { BT_M = 0, BT_I = 1, BT_D = 2 };
    int status = AIM_PAIR_OK;
    int sc = score, k = ak;
    int offset = hist[(lut(HX.m, sc) + k - lut(SH.lo, sc)) * kWave];
    bool valid = valid_loc(k, offset);
    int bt = BT_M;
    int v = offset - k, h = offset;
    while (v > 0 && h > 0 && sc > 0) {
        if (!valid) {
            valid = valid_loc(k, offset);
            if (valid) {   // add_trailing_gap, wfa_backtracing.c:48-69
                if (k < ak) for (int i = k; i < ak; ++i) sink.put('I');
                else if (k > ak) for (int i = ak; i < k; ++i) sink.put('D');
            }
        }
        const int s_o = sc - (O + E), s_e = sc - E, s_x = sc - X;
        const Row ro = row(s_o), re = row(s_e), rx = row(s_x);
        int del_ext = kLaneNull, del_open = kLaneNull, ins_ext = kLaneNull, ins_open = kLaneNull, misms = kLaneNull;
        if (bt != BT_I) {
            if ((re.f & LF_PRESENT) && !(re.f & LF_DNULL) && in(re, k + 1)) del_ext = hist[(lut(HX.d, s_e) + k + 1 - re.lo) * kWave];
            if ((ro.f & LF_PRESENT) && in(ro, k + 1)) del_open = hist[(lut(HX.m, s_o) + k + 1 - ro.lo) * kWave];
        }
        if (bt != BT_D) {
            if ((re.f & LF_PRESENT) && (re.f & LF_HASI) && in(re, k - 1)) ins_ext = hist[(lut(HX.i, s_e) + k - 1 - re.lo) * kWave] + 1;
            if ((ro.f & LF_PRESENT) && in(ro, k - 1)) ins_open = hist[(lut(HX.m, s_o) + k - 1 - ro.lo) * kWave] + 1;
        }
        if (bt == BT_M) {
            if ((rx.f & LF_PRESENT) && in(rx, k)) misms = hist[(lut(HX.m, s_x) + k - rx.lo) * kWave] + 1;
        }
        const int max_all = max(misms, max(max(ins_ext, ins_open), max(del_ext, del_open)));
        if (bt == BT_M) {
            const int num_matches = offset - max_all;
            if (num_matches > 0) sink.matches(num_matches);
            offset = max_all;
            v = offset - k;
            h = offset;
            if (v <= 0 || h <= 0) break;
        }
        char op;
        if (max_all == del_ext) { op = 'D'; sc = s_e; ++k; bt = BT_D; }
        else if (max_all == del_open) { op = 'D'; sc = s_o; ++k; bt = BT_M; }
        else if (max_all == ins_ext) { op = 'I'; sc = s_e; --k; --offset; bt = BT_I; }
        else if (max_all == ins_open) { op = 'I'; sc = s_o; --k; --offset; bt = BT_M; }
        else if (max_all == misms) { op = 'X'; sc = s_x; --offset; }
        else { status = AIM_PAIR_WFA_NO_LINK; break; }
        if (valid) sink.put(op);
        v = offset - k;
        h = offset;
    }
    if (status == AIM_PAIR_OK) {
        if (sc == 0) {
            if (offset > 0) sink.matches(offset);
        } else {
            for (; v > 0; --v) sink.put('D');
            for (; h > 0; --h) sink.put('I');
        }
    }
    return status;
}

// ---- run collector ------------------------------------------------------------------------------------------------
// The backtrace emits operations from the END of the alignment towards its beginning; edit_cigar_print (host.c:69-89)
// prints ops[begin_offset, end_offset) forwards. `pos` is the reference's begin_offset (the next position written,
// counting down from plen + tlen - 1); an operation that would land below position 0 is dropped, exactly like the bounds
// test of the ops-row kernels and cigar_rle_kernel's clamp of begin_offset to 0. Adjacent operations of one kind merge into
// a run. The four most recent runs sit in registers as a shift register -- r0 is the newest, i.e. the FIRST run of the
// printed CIGAR -- older ones spill to a lane-interleaved LDS column (spill[i * 64 + lane]: conflict-free; wfa_group's traceback kernel
// spills to a private HBM array instead); a short-read alignment with one edit never touches the spill area.
template <int STRIDE>
struct RunCollector {
    uint32_t r0 = 0u, r1 = 0u, r2 = 0u, r3 = 0u;   // scalars, not an array: a select chain over array elements becomes a dynamic index (scratch)
    uint32_t cur_op = 0u, cur_len = 0u;
    int n = 0;          // completed runs
    int pos;            // begin_offset
    uint32_t *spill;    // this lane's spill area: spill[i * STRIDE] (STRIDE = 64: lane-interleaved LDS column; 1: a private HBM array)
    int spill_cap;      // entries of it
    __device__ __forceinline__ RunCollector(int begin_offset, uint32_t *lane_spill, int cap) : pos(begin_offset), spill(lane_spill), spill_cap(cap) {}
    __device__ __forceinline__ void flush()
    {
        if (cur_len == 0u) return;
        if (n >= 4 && n - 4 < spill_cap) spill[(n - 4) * STRIDE] = r3;
        r3 = r2; r2 = r1; r1 = r0;
        r0 = (cur_len << 8) | cur_op;
        ++n;
        cur_len = 0u;
    }
    __device__ __forceinline__ void emit(uint32_t op, int len)
    {
        const int keep = min(len, pos + 1);   // positions pos, pos - 1, ... that are >= 0
        pos -= len;
        if (keep <= 0) return;
        if (op != cur_op) { flush(); cur_op = op; }
        cur_len += (uint32_t)keep;
    }
    __device__ __forceinline__ void put(char ch) { emit((uint32_t)(unsigned char)ch, 1); }
    __device__ __forceinline__ void matches(int len) { emit((uint32_t)'M', len); }
    // runs 0 .. 3 of the printed CIGAR are r0 .. r3; run i >= 4 (after the last flush()):
    __device__ __forceinline__ uint32_t spilled_run(int i) const { return spill[(n - 1 - i) * STRIDE]; }
    __device__ __forceinline__ bool overflowed() const { return n - 4 > spill_cap; }
};

constexpr int kLaneRunSpill = 12;   // LDS-spilled runs per lane beyond the four in registers (3 KiB per wavefront)
constexpr uint32_t kRunSlot = 3;    // runs per pair written at the pair's own fixed place in the run buffer (below)

// Where a pair's runs go. The shared run buffer of the compact-CIGAR format is bump-allocated through one cursor
// (cigar_rle_kernel: one atomic per wavefront). A one-pair-per-lane kernel whose alignments have <= 3 runs does better: when
// the buffer holds 3 runs per pair, pair p owns runs[3p, 3p + 3) -- one coalesced 12-B store per lane, no atomic, no
// dependence on scheduling -- and the cursor starts at 3 n_pairs (aim_capi.hip), so that only the pairs with more runs
// allocate behind the slots (one atomic per wavefront that holds such a pair). a.run_slot is 3 in that mode, 0 otherwise
// (run buffer smaller than 3 n_pairs: everything is bump-allocated). Three, not four: a pair without an edit has one run, with one
// edit three (M X M), with two five -- a fourth slot is filled by almost no pair and is 4 bytes per pair on the PCIe link,
// which is what bounds the end-to-end rate with CIGAR (e2e 6.1-6.4e8 pairs/s with four).
template <typename Coll>
__device__ __forceinline__ void store_cigar(const KArgs &a, uint32_t pair, bool active, uint32_t idx, int score, int status,
                                            Coll &c, uint32_t run_slot, int lane)
{
    const bool ok = active && status == AIM_PAIR_OK;
    uint32_t n_runs = ok ? (uint32_t)c.n : 0u;
    const bool coll_ovf = ok && c.overflowed();
    const bool in_slot = n_runs <= run_slot;
    // bump allocation for the pairs that do not fit their slot: wave-level exclusive prefix sum, one atomic per wavefront
    const uint32_t need = (ok && !in_slot && !coll_ovf) ? n_runs : 0u;
    uint32_t off = pair * run_slot;
    bool fits = true;
    if (__ballot(need != 0u) != 0ull) {   // wave-uniform, rare in slotted mode
        uint32_t incl = need;
#define AIM_PK_SCAN(ctrl, rmask) incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, ctrl, rmask, 0xf, false)
        AIM_PK_SCAN(0x111, 0xf);   // row_shr:1
        AIM_PK_SCAN(0x112, 0xf);   // row_shr:2
        AIM_PK_SCAN(0x114, 0xf);   // row_shr:4
        AIM_PK_SCAN(0x118, 0xf);   // row_shr:8
        AIM_PK_SCAN(0x142, 0xa);   // row_bcast:15
        AIM_PK_SCAN(0x143, 0xc);   // row_bcast:31
#undef AIM_PK_SCAN
        const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(a.cursor, total);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (need) {
            off = base + incl - need;
            fits = off + need <= a.runs_cap;
        }
    }
    if (ok && !coll_ovf && fits) {
        if (in_slot && run_slot == kRunSlot) {
            typedef uint32_t aim_u32x3 __attribute__((ext_vector_type(3), aligned(4)));
            aim_u32x3 v3;
            v3.x = c.r0; v3.y = n_runs > 1 ? c.r1 : 0u; v3.z = n_runs > 2 ? c.r2 : 0u;
            *reinterpret_cast<aim_u32x3 *>(a.runs + off) = v3;
        } else {
            // (no select chain over r0..r3 here: the compiler turns one into a dynamically indexed stack object)
            uint32_t *dst = a.runs + off;
            dst[0] = c.r0;
            if (n_runs > 1) dst[1] = c.r1;
            if (n_runs > 2) dst[2] = c.r2;
            if (n_runs > 3) dst[3] = c.r3;
            for (uint32_t i = 4; i < n_runs; ++i) dst[i] = c.spilled_run((int)i);
        }
    }
    if (active) {
        const bool ovf = ok && (coll_ovf || !fits || n_runs > 0xffffu);
        typedef uint32_t aim_u32x4 __attribute__((ext_vector_type(4)));
        aim_u32x4 h;
        h.x = idx;
        h.y = (uint32_t)score;
        h.z = off;
        h.w = ((ok && !ovf) ? n_runs : 0u) | (((uint32_t)status | (ovf ? AIM_CIGAR_OVERFLOW : 0u)) << 16);
        *reinterpret_cast<aim_u32x4 *>(a.cig + pair) = h;   // aim_cigar_t {idx, score, run_offset, n_runs:16 | status:16}
    }
}

// 4-byte-aligned vector loads of a packed row (rows are ceil(READ_SIZE/16) dwords: 28 B at READ_SIZE 112)
typedef uint32_t pk_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t pk_u32x2 __attribute__((ext_vector_type(2), aligned(4)));
template <int NP>
__device__ __forceinline__ void load_packed_row(const uint32_t *row, uint32_t (&out)[NP])
{
    constexpr int Q = NP / 4, R = NP % 4;
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        const pk_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const pk_u32x4 *>(row) + j);
        out[4 * j] = v.x; out[4 * j + 1] = v.y; out[4 * j + 2] = v.z; out[4 * j + 3] = v.w;
    }
    if (R >= 2) {
        const pk_u32x2 v = __builtin_nontemporal_load(reinterpret_cast<const pk_u32x2 *>(row + 4 * Q));
        out[4 * Q] = v.x; out[4 * Q + 1] = v.y;
    }
    if (R & 1) out[NP - 1] = __builtin_nontemporal_load(row + NP - 1);
}

// the reference's ops row as the backtrace's sink (default ABI): operations[begin_offset--] = ch over the 'M' pre-fill
struct OpsSink {
    char *ops;
    int cap, pos;
    __device__ __forceinline__ void put(char ch) { if (pos >= 0 && pos < cap) ops[pos] = ch; --pos; }
    __device__ __forceinline__ void matches(int n) { pos -= n; }
};

enum { PK_OUT_SCORE = 0, PK_OUT_RUNS = 1, PK_OUT_OPS = 2 };   // what the kernel writes: {idx, score} / result_t; aim_cigar_t + runs; result_t + ops rows

template <int X, int O, int E, int MAXS, int NP, int OUT, bool DYN>
__global__ __launch_bounds__(64, DYN ? 2 : 1) void wfa_lane_packed_kernel(KArgs a, uint32_t run_slot)
{
    constexpr WfShape<X, O, E, MAXS> SH{};
    constexpr WfHist<X, O, E, MAXS> HX{};
    constexpr bool BT = OUT != PK_OUT_SCORE;
    static_assert(DYN || SH.maxw < 10, "WFA-adaptive reduction could fire: shape needs the dynamic-bounds score loop (DYN)");
    constexpr int KW = SH.kmax - SH.kmin + 1;
    constexpr int kSpillBytes = OUT == PK_OUT_RUNS ? kLaneRunSpill * kWave * 4 : 0;   // LDS: [run spill][history column (DYN + BT)]
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    const int lane = threadIdx.x;
    const uint32_t n_groups = (a.n_pairs + kWave - 1) / kWave;
    const int ms_run = a.p.max_score;           // runtime MAX_SCORE <= MAXS
    const bool req8 = a.p.flags & AIM_FLAG_REQ8, res8 = a.p.flags & AIM_FLAG_RES8;   // wave-uniform

    // one group ahead, in registers: request + both packed rows of this lane's pair
    uint32_t Pn[NP], Tn[NP];
    int plen_n = 0, tlen_n = 0;
    uint32_t idx_n = 0;
    auto fetch = [&](uint32_t g_) {
        uint32_t pr = g_ * kWave + (uint32_t)lane;
        if (pr >= a.n_pairs) pr = a.n_pairs - 1;   // lanes past the batch tail re-read the last pair (never stored)
        if (req8) {
            const uint2 q = *reinterpret_cast<const uint2 *>(reinterpret_cast<const aim_request8_t *>(a.req) + pr);
            plen_n = (int16_t)(q.x & 0xffffu); tlen_n = (int16_t)(q.x >> 16); idx_n = q.y;
        } else {
            const uint4 q = *reinterpret_cast<const uint4 *>(a.req + pr);
            plen_n = (int)q.x; tlen_n = (int)q.y; idx_n = q.w;
        }
        load_packed_row<NP>(a.packedP + (uint64_t)pr * NP, Pn);
        load_packed_row<NP>(a.packedT + (uint64_t)pr * NP, Tn);
    };
    uint32_t grp;
    bool have = xcd_unit(n_groups, 0, &grp);
    if (have) fetch(grp);
    for (uint32_t it = 0; have; ++it) {
        const uint32_t pair = grp * kWave + lane;
        const bool active = pair < a.n_pairs;
        uint32_t P[NP], T[NP];
#pragma unroll
        for (int j = 0; j < NP; ++j) { P[j] = Pn[j]; T[j] = Tn[j]; }
        const int plen = active ? plen_n : 0, tlen = active ? tlen_n : 0;
        const uint32_t idx = idx_n;
        uint32_t ngrp = 0;
        const bool nhave = xcd_unit(n_groups, it + 1, &ngrp);
        if (nhave) fetch(ngrp);                    // flies under the compute below
        __builtin_amdgcn_sched_barrier(0);

        // ---- mismatch bit-vectors per diagonal: bit pair v of dk[k] != 0  <=>  P[v] != T[v + k] ----
        uint32_t dk[KW][NP];
#pragma unroll
        for (int kk = 0; kk < KW; ++kk) {
            const int k = SH.kmin + kk;
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                uint32_t ts;
                if (k == 0) ts = T[j];
                else if (k > 0) ts = __builtin_amdgcn_alignbit(j + 1 < NP ? T[j + 1] : 0u, T[j], 2 * k);
                else ts = __builtin_amdgcn_alignbit(T[j], j > 0 ? T[j - 1] : 0u, 32 + 2 * k);
                dk[kk][j] = P[j] ^ ts;
            }
        }
        int Mv[MAXS + 1][KW], Iv[MAXS + 1][KW], Dv[MAXS + 1][KW];
        int score;
        bool done;
        int16_t *hist = reinterpret_cast<int16_t *>(smem + kSpillBytes) + lane;   // this lane's history column (DYN + BT)
        if constexpr (DYN) {
            score = wfa_scores_dynamic<X, O, E, MAXS, NP, KW, BT>(dk, plen, tlen, ms_run, (a.p.flags & AIM_FLAG_REDUCE) != 0, active, hist);
            done = score <= ms_run;
        } else {
            wfa_scores_static<X, O, E, MAXS, NP, KW>(dk, plen, tlen, ms_run, active, Mv, Iv, Dv, score, done);
        }
        if constexpr (OUT == PK_OUT_OPS) {
            // default ABI: result_t + ops row. memset(cigar->operations, 'M', 2*READ_SIZE) (wfa.c:465), then the walk patches the edits in
            const int rs = a.p.read_size;
            OpsSink sink;
            sink.ops = a.ops + (uint64_t)(active ? pair : 0u) * (2 * rs);
            sink.cap = 2 * rs;
            sink.pos = plen + tlen - 1;                 // edit_cigar_allocate, wfa.c:57-67
            int status = AIM_PAIR_OK;
            // only the pieces that can hold a printed operation (wfa_lane.hpp, same bound: begin_offset >= min(plen, tlen) - MAX_SCORE / e), wave-uniform
            const int w_lo = wave_min_i32(active ? max(0, min(plen, tlen) - MAXS / E) >> 4 : (1 << 20));
            const int w_hi = -wave_min_i32(active ? -((plen + tlen + 15) >> 4) : 0);
            if (active) {
                uint4 *orow = reinterpret_cast<uint4 *>(sink.ops);
                const uint4 mm = make_uint4(0x4D4D4D4Du, 0x4D4D4D4Du, 0x4D4D4D4Du, 0x4D4D4D4Du);
                for (int j = w_lo; j < w_hi && j < (2 * rs) / 16; ++j) orow[j] = mm;
                if (done) {
                    if constexpr (DYN) status = wfa_backtrace_dynamic<X, O, E, MAXS>(hist, score, plen, tlen, sink);
                    else status = wfa_backtrace_static<X, O, E, MAXS, KW>(Mv, Iv, Dv, score, plen, tlen, sink);
                    if (status == AIM_PAIR_OK) ++sink.pos;
                }
                aim_result_t r;
                r.max_operations = plen + tlen;
                r.begin_offset = sink.pos;
                r.end_offset = plen + tlen;
                r.score = score;
                r.status = status;
                r.idx = idx;
                store_result(a, pair, r);
            }
        } else if constexpr (OUT == PK_OUT_RUNS) {
            uint32_t *spill = reinterpret_cast<uint32_t *>(smem) + lane;
            RunCollector<kWave> coll(plen + tlen - 1, spill, kLaneRunSpill);   // edit_cigar_allocate, wfa.c:57-67
            int status = AIM_PAIR_OK;
            if (active && done) {
                if constexpr (DYN) status = wfa_backtrace_dynamic<X, O, E, MAXS>(hist, score, plen, tlen, coll);
                else status = wfa_backtrace_static<X, O, E, MAXS, KW>(Mv, Iv, Dv, score, plen, tlen, coll);
            }
            coll.flush();
            if (coll.n == 0) {   // nothing inside [0, end): edit_cigar_print still prints operations[begin_offset] = 'M'
                coll.cur_op = (uint32_t)'M'; coll.cur_len = 1u;
                coll.flush();
            }
            store_cigar(a, pair, active, idx, score, status, coll, run_slot, lane);
        } else if (active) {
            if (res8) {            // ONE global_store_dwordx2 per lane (512 contiguous bytes per wavefront)
                typedef uint32_t aim_u32x2 __attribute__((ext_vector_type(2)));
                aim_u32x2 v2; v2.x = idx; v2.y = (uint32_t)score;
                *reinterpret_cast<aim_u32x2 *>(reinterpret_cast<aim_result8_t *>(a.res) + pair) = v2;
            } else {               // dwordx4 + dwordx2 (24-B struct, 8-B aligned)
                uint32_t *dst = reinterpret_cast<uint32_t *>(a.res + pair);
                *reinterpret_cast<uint4 *>(dst) = make_uint4((uint32_t)(plen + tlen), (uint32_t)(plen + tlen - 1), (uint32_t)(plen + tlen), (uint32_t)score);
                *reinterpret_cast<uint2 *>(dst + 4) = make_uint2((uint32_t)AIM_PAIR_OK, idx);
            }
        }
        have = nhave;
        grp = ngrp;
    }
}

// ---------------------------------------------------------------------------------------------------
// host-side planning / dispatch: ONE list of instantiations (penalties 3,4,1; NP = packed dwords per row)
// ---------------------------------------------------------------------------------------------------
#define AIM_LANEPK_NP_LIST(F) F(5) F(7) F(9) F(10) F(11)

inline bool wfa_lane_packed_np_ok(int np)
{
#define AIM_LANEPK_NP_TEST(N) if (np == N) return true;
    AIM_LANEPK_NP_LIST(AIM_LANEPK_NP_TEST)
#undef AIM_LANEPK_NP_TEST
    return false;
}

// shapes the packed kernel takes; with_cigar = the caller wants the compact CIGAR (BACKTRACE set)
inline bool wfa_lane_packed_supported(const aim_params_t &p, bool allow_dynamic = true)
{
    if (p.algo != AIM_ALGO_WFA) return false;
    const int ms = wfa_lane_static_max_score(p);                          // the penalty sets of AIM_LANE_COST_SETS (wfa_lane.hpp)
    if (ms < 0) return false;
    if (!wfa_lane_packed_np_ok((p.read_size + 15) / 16)) return false;
    if (p.max_score <= ms) return true;
    if (p.mismatch != 3 || p.gap_o != 4 || p.gap_e != 1) return false;   // the dynamic-bounds shape: the reference's default penalties only
    return allow_dynamic && p.max_score <= kLaneDynMaxScore;
}

inline void wfa_lane_packed_plan(const aim_params_t &p, uint32_t n_pairs, const Knobs &kn, uint32_t *grid, uint32_t *block, size_t *lds)
{
    const uint32_t n_groups = (n_pairs + kWave - 1) / kWave;
    const bool bt = p.flags & AIM_FLAG_BACKTRACE, dyn = p.max_score > wfa_lane_static_max_score(p);
    // LDS: run spill (compact CIGAR output only; reserved whenever BACKTRACE is set) + the dynamic-bounds shape's history column
    constexpr WfHist<3, 4, 1, kLaneDynMaxScore> HX{};
    *lds = bt ? (size_t)kLaneRunSpill * kWave * 4 + (dyn ? (size_t)HX.total * kWave * 2 : 0) : 0;
    // resident single-wave workgroups per CU by the register count of the instantiation (static shapes <= 89 VGPRs; dynamic-bounds
    // score-only 102 / 126 / 150 / 162 / 173 at 5 / 7 / 9 / 10 / 11 dwords per row, + 23 with CIGAR)
    const int np = (p.read_size + 15) / 16;
    uint32_t per_cu = !dyn ? AIM_LANEPK_WGS_PER_CU : ((np <= 7 && !bt) ? 16u : (np <= 9 ? 12u : (np <= 10 && !bt ? 12u : AIM_LANEPK_DYN_WGS_PER_CU)));
    if (*lds) per_cu = (uint32_t)std::min<size_t>(per_cu, lds_workgroups_per_cu(*lds));
    uint32_t g = resident_grid(kn, per_cu);
    const uint32_t need = ((n_groups + 7u) / 8u) * 8u;
    if (g > need) g = need < 8u ? 8u : need;
    *grid = g;
    *block = kWave;
}

// Output follows the buffers given: compact CIGAR when ka.cig is set, else (BACKTRACE) result_t + ops rows, else scores.
// Kernels are instantiated in ONE translation unit (tu_*.hip defines AIM_TU_WFA_LANE_PACKED); every other includer sees the declaration only.
#ifdef AIM_TU_WFA_LANE_PACKED
void wfa_lane_packed_launch(const aim_params_t &p, uint32_t grid, size_t lds, const KArgs &ka, uint32_t run_slot, hipStream_t s)
{
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    const int np = (p.read_size + 15) / 16;
    const int out = !bt ? PK_OUT_SCORE : (ka.cig ? PK_OUT_RUNS : PK_OUT_OPS);
#define AIM_LANEPK_ONE(X, O, E, N, MS, OUTV, DYNV) hipLaunchKernelGGL((wfa_lane_packed_kernel<X, O, E, MS, N, OUTV, DYNV>), dim3(grid), dim3(kWave), lds, s, ka, run_slot)
    if (p.max_score > wfa_lane_static_max_score(p)) {   // dynamic-bounds shape (3 / 4 / 1)
#define AIM_LANEPK_DYN(N)                                                                                   \
    if (np == N) {                                                                                          \
        if (out == PK_OUT_SCORE) AIM_LANEPK_ONE(3, 4, 1, N, kLaneDynMaxScore, PK_OUT_SCORE, true);          \
        else if (out == PK_OUT_RUNS) AIM_LANEPK_ONE(3, 4, 1, N, kLaneDynMaxScore, PK_OUT_RUNS, true);       \
        else AIM_LANEPK_ONE(3, 4, 1, N, kLaneDynMaxScore, PK_OUT_OPS, true);                                \
        return;                                                                                             \
    }
        AIM_LANEPK_NP_LIST(AIM_LANEPK_DYN)
#undef AIM_LANEPK_DYN
        return;
    }
#define AIM_LANEPK_STATIC(X, O, E, MS, N)                                                                   \
    if (np == N) {                                                                                          \
        if (out == PK_OUT_SCORE) AIM_LANEPK_ONE(X, O, E, N, MS, PK_OUT_SCORE, false);                       \
        else if (out == PK_OUT_RUNS) AIM_LANEPK_ONE(X, O, E, N, MS, PK_OUT_RUNS, false);                    \
        else AIM_LANEPK_ONE(X, O, E, N, MS, PK_OUT_OPS, false);                                             \
        return;                                                                                             \
    }
#define AIM_LANEPK_COST(X, O, E, MS)                                                                        \
    if (p.mismatch == X && p.gap_o == O && p.gap_e == E) {                                                  \
        AIM_LANEPK_STATIC(X, O, E, MS, 5) AIM_LANEPK_STATIC(X, O, E, MS, 7) AIM_LANEPK_STATIC(X, O, E, MS, 9) \
        AIM_LANEPK_STATIC(X, O, E, MS, 10) AIM_LANEPK_STATIC(X, O, E, MS, 11)                                 \
        return;                                                                                             \
    }
    AIM_LANE_COST_SETS(AIM_LANEPK_COST)
#undef AIM_LANEPK_COST
#undef AIM_LANEPK_STATIC
#undef AIM_LANEPK_ONE
}
#else
void wfa_lane_packed_launch(const aim_params_t &p, uint32_t grid, size_t lds, const KArgs &ka, uint32_t run_slot, hipStream_t s);
#endif

}  // namespace aim
