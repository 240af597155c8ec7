// wfa_wave.hpp -- general WFA / WFA-adaptive kernel: ONE PAIR PER 64-LANE WAVEFRONT.
//
// Replaces the tasklet loop + affine_wfa_compute of the reference
// (WFA/DPU-WRAM/dpu/wfa.c:342-503) and affine_wavefronts_backtrace
// (WFA/DPU-WRAM/dpu/wfa_backtracing.c:210-351).
//
// Mapping: lanes run along the diagonals k of the current wavefront (64 per
// step, looping for wider wavefronts).  Sequences are staged once per pair in
// LDS (the DPU's WRAM copy, wfa.c:417-459); the descriptors of the last 64
// scores live in an LDS ring; the offset vectors (M/I/D) of the live window --
// the max(x, o+e)+1 most recent scores, all that affine_wfa_compute_next ever
// reads -- live in a second LDS ring (one slot per score, slot_w diagonals), so a
// score step costs LDS round trips, not L2 ones.  With BACKTRACE every
// wavefront is also streamed to a per-wave HBM pool (the DPU's WRAM arena /
// MRAM spill, allocate_new_score wfa.c:143-183) that the traceback walks; a
// wavefront wider than a slot lives in that pool only (slower, same results).
// Handles any MAX_SCORE / READ_SIZE / penalties; the short-read fast path is
// wfa_lane.hpp.
#pragma once

#include "aim_device.hpp"

namespace aim {

typedef int16_t awf_t;               // AFFINE_WAVEFRONT_W16, common.h:92-100
constexpr int kAwfNull = -16384;     // AFFINE_WAVEFRONT_OFFSET_NULL = INT16_MIN/2

// wfa_component (common.h:126-138) with pool offsets instead of pointers.
struct __attribute__((aligned(16))) WfMeta {
    int klo, khi;   // current (possibly reduced) bounds
    int lo, hi;     // allocation bounds (lo_base / hi_base)
    int off_m;      // pool index of M[lo]
    int off_i;      // pool index of I[lo] (iwavefront != NULL <=> WF_HASI)
    int off_d;      // pool index of D[lo] (dwavefront != NULL <=> WF_HASD)
    int flags;
};
enum { WF_PRESENT = 1, WF_MNULL = 2, WF_INULL = 4, WF_DNULL = 8, WF_INLDS = 16, WF_HASI = 32, WF_HASD = 64 };

constexpr int kMetaRing = 64;        // scores kept in the LDS descriptor ring

struct WfaWaveCtx {
    WfMeta *ring;        // LDS, kMetaRing entries
    WfMeta *gmeta;       // HBM, meta_cap entries
    awf_t *pool;         // HBM
    int cur_score;
};

__device__ __forceinline__ WfMeta wf_get_meta(const WfaWaveCtx &c, int s)
{
    if (c.cur_score - s < kMetaRing) return c.ring[s & (kMetaRing - 1)];
    return c.gmeta[s];
}

__device__ __forceinline__ void wf_put_meta(const WfaWaveCtx &c, int s, const WfMeta &m, int lane)
{
    if (lane == 0) {
        c.ring[s & (kMetaRing - 1)] = m;
        c.gmeta[s] = m;
    }
}

// affine_wfa_extend inner loop (wfa.c:198-206) on 4-byte words.
template <typename PtrT>
__device__ __forceinline__ int wf_extend_count(PtrT P, PtrT T, int v, int h, int plen, int tlen, int last_word)
{
    if (v < 0 || h < 0) return 0;
    int rem = min(plen - v, tlen - h);
    if (rem <= 0) return 0;
    int count = 0;
    for (;;) {
        const uint32_t x = load4_unaligned(P, v + count, last_word) ^ load4_unaligned(T, h + count, last_word);
        const int m = x ? (__builtin_ctz(x) >> 3) : 4;
        const int r = rem - count;
        if (m < 4) { count += min(m, r); break; }
        if (r <= 4) { count += r; break; }
        count += 4;
    }
    return count;
}

template <bool BT, bool REDUCE, bool SEQ_LDS>
__global__ __launch_bounds__(64) void wfa_wave_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    const int lane = threadIdx.x;
    const int rs = a.p.read_size;
    const int rsw = rs >> 2;                 // dwords per sequence row (read_size % 8 == 0)
    WfMeta *ring = reinterpret_cast<WfMeta *>(smem);
    awf_t *oring = reinterpret_cast<awf_t *>(smem + kMetaRing * sizeof(WfMeta));
    const int ring_slots = (int)a.ring_slots, slot_w = (int)a.slot_w;
    uint32_t *ldsP = reinterpret_cast<uint32_t *>(smem + kMetaRing * sizeof(WfMeta) + (((size_t)ring_slots * 3 * slot_w * sizeof(awf_t) + 15) & ~(size_t)15));
    uint32_t *ldsT = ldsP + rsw + 2;

    char *wscr = a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave;
    WfaWaveCtx ctx;
    ctx.ring = ring;
    ctx.gmeta = reinterpret_cast<WfMeta *>(wscr);
    ctx.pool = reinterpret_cast<awf_t *>(wscr + (uint64_t)a.meta_cap * sizeof(WfMeta));
    awf_t *pool = ctx.pool;
    const int pool_cap = (int)a.pool_cap;

    const int X = a.p.mismatch, OE = a.p.gap_o + a.p.gap_e, E = a.p.gap_e;
    const int MS = a.p.max_score;
    // component comp (0 M, 1 I, 2 D) of the wavefront of score sc: LDS slot or HBM pool
    auto slot = [&](int sc, int comp) -> awf_t * { return oring + ((sc % ring_slots) * 3 + comp) * slot_w; };
    // row pointer (biased so that row[k] is diagonal k) -- computed once per score step, never per element
    auto rowp = [&](const WfMeta &m, int sc, int comp) -> const awf_t * {
        const awf_t *b = (m.flags & WF_INLDS) ? slot(sc, comp) : pool + (comp == 0 ? m.off_m : (comp == 1 ? m.off_i : m.off_d));
        return b - m.lo;
    };

    // Single-wave workgroup: LDS operations of one wave execute in issue order, so data exchanged through LDS
    // needs no hardware wait, only a compiler fence.  __syncthreads() would add s_waitcnt vmcnt(0), i.e. a full
    // round trip for the history / descriptor stores still in flight -- only paid when a row lives in the HBM pool.
    const bool meta_in_lds = max(X, OE) < kMetaRing;
    auto sync = [&](bool rows_in_lds) {
        if (rows_in_lds && meta_in_lds) asm volatile("" ::: "memory");
        else __syncthreads();
    };
    // direct mode: every pair of the batch; indirect mode: the pairs the lane kernel could not pack
    const uint32_t n_units = a.todo ? min(a.todo[0], a.n_pairs) : a.n_pairs;

    for (uint32_t it = 0;; ++it) {
        uint32_t pair;
        if (!xcd_unit(n_units, it, &pair)) break;   // past this block's slice: done
        if (a.todo) pair = a.todo[16 + pair];
        const aim_request_t rq = load_request(a, pair);
        const int plen = rq.pattern_len, tlen = rq.text_len;
        const uint32_t *gP = reinterpret_cast<const uint32_t *>(a.patterns + (uint64_t)pair * rs);
        const uint32_t *gT = reinterpret_cast<const uint32_t *>(a.texts + (uint64_t)pair * rs);
        char *ops = BT ? a.ops + (uint64_t)pair * 2 * rs : nullptr;

        __syncthreads();  // previous pair's LDS reads are done
        if (SEQ_LDS) {
            const int pw = (plen + 3) >> 2, tw = (tlen + 3) >> 2;
            for (int w = lane; w < pw; w += kWave) ldsP[w] = gP[w];
            for (int w = lane; w < tw; w += kWave) ldsT[w] = gT[w];
        }
        if (BT) {  // memset(cigar->operations, 'M', 2*READ_SIZE), wfa.c:465
            uint32_t *o4 = reinterpret_cast<uint32_t *>(ops);
            for (int w = lane; w < (rs >> 1); w += kWave) o4[w] = 0x4D4D4D4Du;
        }
        const int last_word = SEQ_LDS ? rsw + 1 : rsw - 1;

        // edit_cigar_allocate, wfa.c:57-67
        const int max_ops = plen + tlen;
        int begin_offset = max_ops - 1;
        const int end_offset = max_ops;
        int status = AIM_PAIR_OK;
        int final_score;
        const int ak = tlen - plen;   // alignment_k

        // wavefronts[0] = allocate_new_score(0,0,0,0); M[0] = 0   (wfa.c:347-348)
        int pool_used = 1;
        WfMeta cur;
        cur.klo = cur.khi = cur.lo = cur.hi = 0;
        cur.off_m = 0; cur.off_i = -1; cur.off_d = -1;
        cur.flags = WF_PRESENT | WF_INULL | WF_DNULL | (ring_slots > 0 ? WF_INLDS : 0);
        ctx.cur_score = 0;
        if (lane == 0) {
            pool[0] = 0;
            if (ring_slots > 0) slot(0, 0)[0] = 0;
        }
        wf_put_meta(ctx, 0, cur, lane);
        __syncthreads();

        int score = 0;
        for (;;) {
            const bool live = (cur.flags & WF_PRESENT) && !(cur.flags & WF_MNULL);
            // ---- affine_wfa_extend, wfa.c:186-208 -------------------------------
            if (live) {
                const bool inlds = cur.flags & WF_INLDS;
                // one body, two call sites: after inlining the LDS call site compiles to ds_read/ds_write and the
                // pool call site to global loads/stores (a merged pointer would force flat accesses)
                auto extend_row = [&](awf_t *mrow) {
                    for (int k = cur.klo + lane; k <= cur.khi; k += kWave) {
                        const int moff = mrow[k - cur.lo];
                        if (moff >= 0) {
                            const int cnt = SEQ_LDS
                                ? wf_extend_count((const uint32_t *)ldsP, (const uint32_t *)ldsT, moff - k, moff, plen, tlen, last_word)
                                : wf_extend_count(gP, gT, moff - k, moff, plen, tlen, last_word);
                            if (cnt) mrow[k - cur.lo] = (awf_t)(moff + cnt);
                            if (BT && inlds) pool[cur.off_m + (k - cur.lo)] = (awf_t)(moff + cnt);   // HBM history for the traceback
                        } else if (BT && inlds) {
                            pool[cur.off_m + (k - cur.lo)] = (awf_t)moff;
                        }
                    }
                };
                if (inlds) extend_row(slot(score, 0));
                else extend_row(pool + cur.off_m);
                sync(inlds);
            }
            // ---- affine_wfa_reduce_wvs (WFA-adaptive), wfa.c:69-140 --------------
            if (REDUCE && live && (cur.khi - cur.klo + 1) >= 10) {
                auto reduce_row = [&](const awf_t *mk) {
                int mind = max(plen, tlen);
                {
                    int d = 0x7fffffff;   // per-lane minimum over its diagonals, then ONE wave reduction
                    for (int k = cur.klo + lane; k <= cur.khi; k += kWave) {
                        const int off = mk[k];
                        d = min(d, max(plen - (off - k), tlen - off));
                    }
                    mind = min(mind, wave_min_i32(d));
                }
                int nklo = cur.klo, nkhi = cur.khi;
                const int top_limit = min(ak - 1, cur.khi);
                if (cur.klo < top_limit) {
                    bool found = false;
                    for (int base = cur.klo; base < top_limit && !found; base += kWave) {
                        const int k = base + lane;
                        bool ok = false;
                        if (k < top_limit) {
                            const int off = mk[k];
                            ok = (max(plen - (off - k), tlen - off) - mind) <= 50;
                        }
                        const unsigned long long mask = __ballot(ok);
                        if (mask) { nklo = base + __builtin_ctzll(mask); found = true; }
                    }
                    if (!found) nklo = top_limit;
                }
                const int bottom_limit = max(ak + 1, nklo);
                if (cur.khi > bottom_limit) {
                    bool found = false;
                    for (int top = cur.khi; top > bottom_limit && !found; top -= kWave) {
                        const int k = top - lane;
                        bool ok = false;
                        if (k > bottom_limit) {
                            const int off = mk[k];
                            ok = (max(plen - (off - k), tlen - off) - mind) <= 50;
                        }
                        const unsigned long long mask = __ballot(ok);
                        if (mask) { nkhi = top - __builtin_ctzll(mask); found = true; }
                    }
                    if (!found) nkhi = bottom_limit;
                }
                if (nklo > nkhi) {
                    cur.flags |= WF_MNULL | WF_INULL | WF_DNULL;   // wfa.c:131-139
                } else {
                    cur.klo = nklo;
                    cur.khi = nkhi;
                }
                wf_put_meta(ctx, score, cur, lane);
                sync(true);
                };
                if (cur.flags & WF_INLDS) reduce_row(slot(score, 0) - cur.lo);
                else reduce_row(pool + cur.off_m - cur.lo);
            }
            // ---- affine_wfa_end_reached, wfa.c:210-230 ---------------------------
            bool done = false;
            if ((cur.flags & WF_PRESENT) && !(cur.flags & WF_MNULL) && cur.klo <= ak && cur.khi >= ak) {
                const int off = (cur.flags & WF_INLDS) ? (int)slot(score, 0)[ak - cur.lo] : (int)pool[cur.off_m + (ak - cur.lo)];
                done = off >= tlen;
            }
            if (done) { final_score = score; break; }
            ++score;
            if (score > MS) { final_score = score; break; }   // wfa.c:368-376
            ctx.cur_score = score;

            // ---- affine_wfa_compute_next, wfa.c:268-340 --------------------------
            const int s_sub = score - X, s_o = score - OE, s_e = score - E;
            WfMeta ms, mo, me;
            ms.flags = mo.flags = me.flags = 0;
            if (s_sub >= 0) ms = wf_get_meta(ctx, s_sub);
            if (s_o >= 0) mo = wf_get_meta(ctx, s_o);
            if (s_e >= 0) me = wf_get_meta(ctx, s_e);
            const bool m_sub_null = (s_sub < 0) || !(ms.flags & WF_PRESENT) || (ms.flags & WF_MNULL);
            const bool m_o_null = (s_o < 0) || !(mo.flags & WF_PRESENT) || (mo.flags & WF_MNULL);
            const bool i_e_null = (s_e < 0) || !(me.flags & WF_PRESENT) || !(me.flags & WF_HASI) || (me.flags & WF_INULL);
            const bool d_e_null = (s_e < 0) || !(me.flags & WF_PRESENT) || !(me.flags & WF_HASD) || (me.flags & WF_DNULL);
            const bool i_out_null = m_o_null && i_e_null;
            const bool d_out_null = m_o_null && d_e_null;
            if (m_sub_null && i_out_null && d_out_null) {
                cur.flags = 0;   // wavefronts[score] = NULL
                cur.klo = cur.lo = 0; cur.khi = cur.hi = -1;
                cur.off_m = cur.off_i = cur.off_d = -1;
                wf_put_meta(ctx, score, cur, lane);
                sync(true);
                continue;
            }
            const int sub_lo = m_sub_null ? 1 : ms.klo, sub_hi = m_sub_null ? -1 : ms.khi;
            const int o_lo = m_o_null ? 1 : mo.klo, o_hi = m_o_null ? -1 : mo.khi;
            const bool e_none = i_e_null && d_e_null;
            const int e_lo = e_none ? 1 : me.klo, e_hi = e_none ? -1 : me.khi;
            const int lo = min(min(sub_lo, o_lo), e_lo) - 1;
            const int hi = max(max(sub_hi, o_hi), e_hi) + 1;
            const int len = hi - lo + 1;
            const int narr = 1 + (d_out_null ? 0 : 1) + (i_out_null ? 0 : 1);
            // allocate_new_score, wfa.c:143-183: an LDS slot when the wavefront fits one; an HBM pool
            // region when it does not, and always with BACKTRACE (history)
            const bool inlds = ring_slots > 0 && len <= slot_w;
            const bool in_pool = BT || !inlds;
            if (in_pool && pool_used + len * narr > pool_cap) {
                if (BT) {   // allocate_new(): "out of memory" + exit(1), dpu_allocator_wram.c:19-23
                    status = AIM_PAIR_NOMEM;
                    final_score = score;
                    break;
                }
                pool_used = 0;   // score-only: the pool is a ring sized for the live window
                // The plan never hands a score-only wavefront less than the live window ((R+2) * 3 * (2*MAX_SCORE+3)
                // entries, make_plan); a wavefront that still does not fit after the wrap would run over the next
                // workgroup's scratch, so it is reported like the DPU arena's "out of memory" instead.
                if ((uint32_t)(len * narr) > pool_cap) {
                    status = AIM_PAIR_NOMEM;
                    final_score = score;
                    break;
                }
            }
            cur.flags = WF_PRESENT | (i_out_null ? WF_INULL : WF_HASI) | (d_out_null ? WF_DNULL : WF_HASD) | (inlds ? WF_INLDS : 0);
            cur.klo = cur.lo = lo;
            cur.khi = cur.hi = hi;
            cur.off_m = in_pool ? pool_used : -1;
            cur.off_d = (d_out_null || !in_pool) ? -1 : pool_used + len;
            cur.off_i = (i_out_null || !in_pool) ? -1 : pool_used + len * (d_out_null ? 1 : 2);
            if (in_pool) pool_used += len * narr;
            wf_put_meta(ctx, score, cur, lane);
            // affine_wfa_compute_offsets, wfa.c:231-266 -- one body, LDS-typed and generic call sites (see extend)
            auto compute_row = [&](const awf_t *r_mo, const awf_t *r_ie, const awf_t *r_de, const awf_t *r_ms, awf_t *om, awf_t *oi,
                                   awf_t *od) {
                for (int k = lo + lane; k <= hi; k += kWave) {
                    int ins = -10;
                    if (!i_out_null) {
                        const int ins_g = (!m_o_null && o_lo <= k - 1 && k - 1 <= o_hi) ? (int)r_mo[k - 1] : kAwfNull;
                        const int ins_i = (!i_e_null && e_lo <= k - 1 && k - 1 <= e_hi) ? (int)r_ie[k - 1] : kAwfNull;
                        ins = (ins_g == kAwfNull && ins_i == kAwfNull) ? kAwfNull : (int)(awf_t)(max(ins_g, ins_i) + 1);
                        oi[k - lo] = (awf_t)ins;
                        if (BT && inlds) pool[cur.off_i + (k - lo)] = (awf_t)ins;
                    }
                    int del = -10;
                    if (!d_out_null) {
                        const int del_g = (!m_o_null && o_lo <= k + 1 && k + 1 <= o_hi) ? (int)r_mo[k + 1] : kAwfNull;
                        const int del_d = (!d_e_null && e_lo <= k + 1 && k + 1 <= e_hi) ? (int)r_de[k + 1] : kAwfNull;
                        del = max(del_g, del_d);
                        od[k - lo] = (awf_t)del;
                        if (BT && inlds) pool[cur.off_d + (k - lo)] = (awf_t)del;
                    }
                    int sub = -10;
                    if (!m_sub_null) sub = (sub_lo <= k && k <= sub_hi) ? (int)(awf_t)(r_ms[k] + 1) : kAwfNull;
                    om[k - lo] = (awf_t)max(del, max(sub, ins));
                }
            };
            const bool all_lds = inlds && (m_o_null || (mo.flags & WF_INLDS)) && (e_none || (me.flags & WF_INLDS)) &&
                                 (m_sub_null || (ms.flags & WF_INLDS));
            if (all_lds) {
                compute_row(slot(s_o < 0 ? 0 : s_o, 0) - mo.lo, slot(s_e < 0 ? 0 : s_e, 1) - me.lo, slot(s_e < 0 ? 0 : s_e, 2) - me.lo,
                            slot(s_sub < 0 ? 0 : s_sub, 0) - ms.lo, slot(score, 0), slot(score, 1), slot(score, 2));
            } else {
                compute_row(m_o_null ? nullptr : rowp(mo, s_o, 0), i_e_null ? nullptr : rowp(me, s_e, 1),
                            d_e_null ? nullptr : rowp(me, s_e, 2), m_sub_null ? nullptr : rowp(ms, s_sub, 0),
                            inlds ? slot(score, 0) : pool + cur.off_m, inlds ? slot(score, 1) : pool + cur.off_i,
                            inlds ? slot(score, 2) : pool + cur.off_d);
            }
            sync(inlds);
        }
        __syncthreads();   // history and descriptors in HBM are complete before the traceback reads them

        // ---- affine_wavefronts_backtrace, wfa_backtracing.c:210-351 ---------------
        // Wave-uniform scalar walk; the ops row is pre-filled with 'M' so match
        // runs only move begin_offset, and gap runs are filled by all lanes.
        if (BT && status == AIM_PAIR_OK && final_score <= MS) {
            enum { BT_M = 0, BT_I = 1, BT_D = 2 };
            const int ops_cap = 2 * rs;
            int sc = final_score, k = ak;
            WfMeta m0 = ctx.gmeta[sc];
            int offset = pool[m0.off_m + (k - m0.lo)];
            auto valid_loc = [&](int kk, int off) {
                const int v = off - kk, h = off;
                return v > 0 && v <= plen && h > 0 && h <= tlen;
            };
            auto put_run = [&](char ch, int count) {   // ops[begin--] = ch, count times
                for (int i = lane; i < count; i += kWave) {
                    const int at = begin_offset - i;
                    if (at >= 0 && at < ops_cap) ops[at] = ch;
                }
                if (count > 0) begin_offset -= count;
            };
            bool valid = valid_loc(k, offset);
            int bt = BT_M;
            int v = offset - k, h = offset;
            while (v > 0 && h > 0 && sc > 0) {
                if (!valid) {
                    valid = valid_loc(k, offset);
                    if (valid) {   // add_trailing_gap, wfa_backtracing.c:48-69
                        if (k < ak) put_run('I', ak - k);
                        else if (k > ak) put_run('D', k - ak);
                    }
                }
                const int s_o = sc - OE, s_e = sc - E, s_x = sc - X;
                WfMeta mo, me, mx;
                mo.flags = me.flags = mx.flags = 0;
                if (s_o >= 0) mo = ctx.gmeta[s_o];
                if (s_e >= 0) me = ctx.gmeta[s_e];
                if (s_x >= 0 && bt == BT_M) mx = ctx.gmeta[s_x];
                int del_ext = kAwfNull, del_open = kAwfNull, ins_ext = kAwfNull, ins_open = kAwfNull, misms = kAwfNull;
                if (bt != BT_I) {
                    if ((me.flags & WF_PRESENT) && !(me.flags & WF_DNULL) && me.klo <= k + 1 && k + 1 <= me.khi)
                        del_ext = pool[me.off_d + (k + 1 - me.lo)];
                    if ((mo.flags & WF_PRESENT) && mo.klo <= k + 1 && k + 1 <= mo.khi)
                        del_open = pool[mo.off_m + (k + 1 - mo.lo)];
                }
                if (bt != BT_D) {
                    if ((me.flags & WF_PRESENT) && (me.flags & WF_HASI) && me.klo <= k - 1 && k - 1 <= me.khi)
                        ins_ext = (awf_t)(pool[me.off_i + (k - 1 - me.lo)] + 1);
                    if ((mo.flags & WF_PRESENT) && mo.klo <= k - 1 && k - 1 <= mo.khi)
                        ins_open = (awf_t)(pool[mo.off_m + (k - 1 - mo.lo)] + 1);
                }
                if (bt == BT_M) {
                    if ((mx.flags & WF_PRESENT) && mx.klo <= k && k <= mx.khi)
                        misms = (awf_t)(pool[mx.off_m + (k - mx.lo)] + 1);
                }
                const int max_del = max(del_ext, del_open);
                const int max_ins = max(ins_ext, ins_open);
                const int max_all = max(misms, max(max_ins, max_del));
                if (bt == BT_M) {
                    const int num_matches = offset - max_all;
                    if (num_matches > 0) begin_offset -= num_matches;   // 'M' already in place
                    offset = max_all;
                    v = offset - k;
                    h = offset;
                    if (v <= 0 || h <= 0) break;
                }
                char op;
                if (max_all == del_ext) { op = 'D'; sc = s_e; ++k; bt = BT_D; }
                else if (max_all == del_open) { op = 'D'; sc = s_o; ++k; bt = BT_M; }
                else if (max_all == ins_ext) { op = 'I'; sc = s_e; --k; offset = (awf_t)(offset - 1); bt = BT_I; }
                else if (max_all == ins_open) { op = 'I'; sc = s_o; --k; offset = (awf_t)(offset - 1); bt = BT_M; }
                else if (max_all == misms) { op = 'X'; sc = s_x; offset = (awf_t)(offset - 1); }
                else { status = AIM_PAIR_WFA_NO_LINK; break; }
                if (valid) {
                    if (lane == 0 && begin_offset >= 0 && begin_offset < ops_cap) ops[begin_offset] = op;
                    --begin_offset;
                }
                v = offset - k;
                h = offset;
            }
            if (status == AIM_PAIR_OK) {
                if (sc == 0) {
                    if (offset > 0) begin_offset -= offset;
                } else {
                    if (v > 0) put_run('D', v);
                    if (h > 0) put_run('I', h);
                }
                ++begin_offset;
            }
        }

        if (lane == 0) {
            aim_result_t r;
            r.max_operations = max_ops;
            r.begin_offset = begin_offset;
            r.end_offset = end_offset;
            r.score = final_score;
            r.status = status;
            r.idx = rq.idx;
            store_result(a, pair, r);
        }
    }
}

// Kernels are instantiated in ONE translation unit (tu_*.hip defines AIM_TU_WFA_WAVE); every other includer sees the declaration only.
#ifdef AIM_TU_WFA_WAVE
void wfa_wave_launch(bool bt, bool red, bool seq_lds, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s)
{
#define AIM_WFW(BTV, REDV)                                                                                              \
    do {                                                                                                                \
        if (seq_lds) hipLaunchKernelGGL((wfa_wave_kernel<BTV, REDV, true>), dim3(grid), dim3(kWave), lds, s, ka);       \
        else hipLaunchKernelGGL((wfa_wave_kernel<BTV, REDV, false>), dim3(grid), dim3(kWave), lds, s, ka);              \
    } while (0)
    if (bt && red) AIM_WFW(true, true);
    else if (bt) AIM_WFW(true, false);
    else if (red) AIM_WFW(false, true);
    else AIM_WFW(false, false);
#undef AIM_WFW
}
#else
void wfa_wave_launch(bool bt, bool red, bool seq_lds, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s);
#endif

}  // namespace aim
