// wfa_diag.hpp -- WFA / WFA-adaptive for long reads and large MAX_SCORE: ONE PAIR PER WAVEFRONT with a FIXED
// diagonal -> lane mapping and register-resident wavefronts.
//
// Same results as affine_wfa_compute / affine_wavefronts_backtrace (WFA/DPU-WRAM/dpu/wfa.c:342-379,
// wfa_backtracing.c:210-351).  Diagonal k has the fixed home idx = k + MAX_SCORE + 1: lane = idx & 63,
// slot = idx >> 6 (<= 8 slots, i.e. MAX_SCORE <= 254).  Consequences:
//   * a lane keeps the M / I / D offsets of ITS diagonals for the current score in VGPRs across
//     compute_next -> extend -> reduce -> end test (wfa.c:354-377): no LDS round trip between the phases;
//   * the k-1 / k+1 neighbours of the gap recurrences (wfa.c:240-253: I[s-e][k-1], D[s-e][k+1]) are the
//     neighbouring LANES: one DPP wave shift per slot (gap_e == 1, so s-e is the score just computed);
//   * only the older M wavefronts (s-x, s-o-e) come from the LDS window (fixed homes, no allocator), all
//     reads of a step issued together; the reduction (wfa.c:69-140) is register + DPP work.
// PMC on BASELINE config 3 showed the LDS-resident group kernel parked 44 % of its wave cycles on ~10 dependent
// LDS round trips per score step; this layout leaves the sequence reads of extend as the only dependent ones.
// Sequence staging, A/C/G/T validation + 2-bit packing, the to-do list for other bytes and the per-pair HBM
// history for the traceback are those of wfa_group.hpp.
#pragma once

#include "aim_device.hpp"
#include "wfa_group.hpp"

#ifndef AIM_DIAG_STAMPS
#define AIM_DIAG_STAMPS 0   // diagnostic build: s_memtime per phase of a score step, summed per wave into the scratch tail
#endif
#if AIM_DIAG_STAMPS
#define AIM_DSTAMP(i) do { unsigned long long t_; __builtin_amdgcn_sched_barrier(0); \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); __builtin_amdgcn_sched_barrier(0); \
        dsum[i] += t_ - dlast; dlast = t_; } while (0)
#else
#define AIM_DSTAMP(i) do { } while (0)
#endif

namespace aim {

constexpr int kDiagSlots = 8;

template <bool REDUCE, bool BT>
__global__ __launch_bounds__(64) void wfa_diag_kernel(KArgs a, GroupCfg c)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int rs = a.p.read_size;
    const int rows_dw = ((rs + 15) / 16) * 4;
    uint32_t *rowsP = reinterpret_cast<uint32_t *>(smem);
    uint32_t *rowsT = rowsP + rows_dw;
    uint32_t *mine = rowsT + rows_dw + 1;
    int16_t *Mw = reinterpret_cast<int16_t *>(mine);                        // [ring_m][wcap]
    int16_t *meta = Mw + c.ring_m * c.wcap;                                  // [ring_m][4] = klo, khi, flags, -
    uint32_t *packed = mine + (c.ring_m * c.wcap * 2 + c.ring_m * 8 + 3) / 4;
    uint32_t *pkP = packed, *pkT = packed + c.np;

    const int X = a.p.mismatch, OE = a.p.gap_o + a.p.gap_e, MS = a.p.max_score;   // gap_e == 1 (planner)
    const int kb = c.kbias;
    uint32_t *todo = reinterpret_cast<uint32_t *>(a.scratch);
    int16_t *hist = BT ? reinterpret_cast<int16_t *>(a.scratch + a.scratch_per_wave) + (size_t)blockIdx.x * c.hist_stride : nullptr;
    const int hrow = 3 * c.wcap + 4;
    auto hM = [&](int s) { return hist + (size_t)s * hrow + kb; };
    auto hI = [&](int s) { return hist + (size_t)s * hrow + c.wcap + kb; };
    auto hD = [&](int s) { return hist + (size_t)s * hrow + 2 * c.wcap + kb; };
    auto hMeta = [&](int s) { return hist + (size_t)s * hrow + 3 * c.wcap; };
    const int nchunk_total = (rs + 15) / 16;

    auto dma = [&](uint32_t pair) {
        const char *gp = a.patterns + (uint64_t)pair * rs, *gt = a.texts + (uint64_t)pair * rs;
        for (int base = 0; base < nchunk_total; base += kWave) {
            const int ch = base + lane;
            if (ch < nchunk_total) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp + (size_t)ch * 16),
                                                 (__attribute__((address_space(3))) void *)(rowsP + base * 4), 16, 0, AIM_LANE_DMA_AUX);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gt + (size_t)ch * 16),
                                                 (__attribute__((address_space(3))) void *)(rowsT + base * 4), 16, 0, AIM_LANE_DMA_AUX);
            }
        }
    };
    auto mrow = [&](int s) { return Mw + (s & (c.ring_m - 1)) * c.wcap + kb; };   // row[k]
    auto fence = [&]() { asm volatile("" ::: "memory"); };
    // my diagonal in slot j
    auto kof = [&](int j) { return (j << 6) + lane - kb; };

#if AIM_DIAG_STAMPS
    unsigned long long dsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dlast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dlast) :: "memory");
#endif
    uint32_t pair;
    bool have = xcd_unit(a.n_pairs, 0, &pair);
    aim_request_t rq_next;
    rq_next.pattern_len = rq_next.text_len = 0; rq_next.padding = 0; rq_next.idx = 0;
    if (have) { dma(pair); rq_next = a.req[pair]; }
    for (uint32_t it = 0; have; ++it) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        const aim_request_t rq = rq_next;
        // every lane loaded the same descriptor: make the lengths wave-uniform so the bookkeeping runs on the scalar unit
        const int plen = __builtin_amdgcn_readfirstlane(rq.pattern_len), tlen = __builtin_amdgcn_readfirstlane(rq.text_len);
        // ---- validate + pack (as wfa_group.hpp, all 64 lanes on one pair) ---------------------------------------
        uint32_t bad = 0;
        {
            const int npw = (rs + 15) / 16;
            for (int j = lane; j < npw; j += kWave) {
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const uint32_t *r = side ? rowsT : rowsP;
                    const int len = side ? tlen : plen;
                    uint32_t out = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int w = 4 * j + i;
                        const uint32_t av = (4 * w < rs) ? r[w] : 0u;
                        const uint32_t t = (av >> 1) & 0x03030303u;
                        const uint32_t rec = __builtin_amdgcn_perm(0u, 0x47544341u, t);
                        const int rem = len - 4 * w;
                        const uint32_t mask = rem >= 4 ? ~0u : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
                        bad |= (rec ^ av) & mask;
                        out |= __builtin_amdgcn_udot4(t, 0x40100401u, 0u, false) << (8 * i);
                    }
                    (side ? pkT : pkP)[j] = out;
                }
            }
            if (lane == 0) { pkP[c.np - 1] = 0u; pkT[c.np - 1] = 0u; }
        }
        bad = __ballot(bad != 0u) ? 1u : 0u;
        uint32_t npair = 0;
        const bool nhave = xcd_unit(a.n_pairs, it + 1, &npair);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __syncthreads();
        if (nhave) { dma(npair); rq_next = a.req[npair]; }
        __builtin_amdgcn_sched_barrier(0);

        auto extend = [&](int k, int off) -> int {   // affine_wfa_extend on packed words, wfa.c:186-208
            int v = off - k, h = off;
            if (off < 0 || v < 0) return off;
            int rem = min(plen - v, tlen - h);
            while (rem > 0) {
                const int wp = v >> 4, wt = h >> 4;
                const uint32_t pw = __builtin_amdgcn_alignbit(pkP[wp + 1], pkP[wp], (uint32_t)((v & 15) * 2));
                const uint32_t tw = __builtin_amdgcn_alignbit(pkT[wt + 1], pkT[wt], (uint32_t)((h & 15) * 2));
                const uint32_t x = pw ^ tw;
                const int n = x ? (__builtin_ctz(x) >> 1) : 16;
                if (n >= rem) { h += rem; break; }
                v += n; h += n; rem -= n;
                if (n < 16) break;
            }
            return h;
        };

        const int ak = tlen - plen;
        int score = 0, final_score = -1, status = AIM_PAIR_OK;
        int begin_offset = plen + tlen - 1;
        bool done = bad != 0u;
        int Mc[kDiagSlots], Ic[kDiagSlots], Dc[kDiagSlots];
#pragma unroll
        for (int j = 0; j < kDiagSlots; ++j) { Mc[j] = 0; Ic[j] = kGrpNull; Dc[j] = kGrpNull; }   // M[0][0] = 0 (wfa.c:348)
        int klo = 0, khi = 0, flags = GF_PRESENT | GF_INULL | GF_DNULL;
        if (lane == 0) { meta[0] = 0; meta[1] = 0; meta[2] = (int16_t)flags; }
        if (BT && !done) {   // memset(cigar->operations, 'M', 2*READ_SIZE), wfa.c:465
            uint4 *orow = reinterpret_cast<uint4 *>(a.ops + (uint64_t)pair * 2 * rs);
            const uint4 mm = make_uint4(0x4D4D4D4Du, 0x4D4D4D4Du, 0x4D4D4D4Du, 0x4D4D4D4Du);
            for (int j = lane; j < (2 * rs) / 16; j += kWave) orow[j] = mm;
        }
        fence();
        AIM_DSTAMP(0);   // staging: wait DMA + pack + next DMA issue
        while (!done) {
            const bool live = (flags & GF_PRESENT) && !(flags & GF_MNULL);
            const int jlo = (klo + kb) >> 6, jhi = (khi + kb) >> 6;
            if (live) {
                int16_t *mr = mrow(score);
                // first 16-base window of every active slot is fetched before any of them is examined (one LDS
                // round trip instead of one per slot); the rare longer runs continue in extend()
                uint32_t xw[kDiagSlots];
                int remv[kDiagSlots];
#pragma unroll
                for (int j = 0; j < kDiagSlots; ++j) {
                    xw[j] = 0u; remv[j] = 0;
                    if (j < jlo || j > jhi) continue;
                    const int k = kof(j), off = Mc[j];
                    const int v = off - k;
                    const bool go = k >= klo && k <= khi && off >= 0 && v >= 0;
                    const int rem = go ? min(plen - v, tlen - off) : 0;
                    remv[j] = rem;
                    const int vv = rem > 0 ? v : 0, hh = rem > 0 ? off : 0;     // clamped: always a valid packed word
                    const uint32_t pw = __builtin_amdgcn_alignbit(pkP[(vv >> 4) + 1], pkP[vv >> 4], (uint32_t)((vv & 15) * 2));
                    const uint32_t tw = __builtin_amdgcn_alignbit(pkT[(hh >> 4) + 1], pkT[hh >> 4], (uint32_t)((hh & 15) * 2));
                    xw[j] = pw ^ tw;
                }
#pragma unroll
                for (int j = 0; j < kDiagSlots; ++j) {
                    if (j < jlo || j > jhi) continue;
                    const int k = kof(j);
                    if (k >= klo && k <= khi) {
                        int off = Mc[j];
                        const int rem = remv[j];
                        if (rem > 0) {
                            const int n = xw[j] ? (__builtin_ctz(xw[j]) >> 1) : 16;
                            if (n >= rem) off += rem;
                            else { off += n; if (n == 16) off = extend(k, off); }
                        }
                        Mc[j] = off;
                        mr[k] = (int16_t)off;                   // visible to the scores that read M[s] later
                        if (BT) hM(score)[k] = (int16_t)off;
                    }
                }
                fence();
            }
            AIM_DSTAMP(1);   // extend
            if (REDUCE && live && (khi - klo + 1) >= 10) {   // affine_wfa_reduce_wvs, wfa.c:69-140, registers + DPP
                int dist[kDiagSlots];                        // distance-to-target of my diagonals, computed once
                int part = 0x7fffffff;
#pragma unroll
                for (int j = 0; j < kDiagSlots; ++j) {
                    dist[j] = 0x7fffffff;
                    if (j < jlo || j > jhi) continue;
                    const int k = kof(j);
                    if (k >= klo && k <= khi) { dist[j] = max(plen - (Mc[j] - k), tlen - Mc[j]); part = min(part, dist[j]); }
                }
                const int mind = min(max(plen, tlen), wave_min_i32(part));
                int nklo = klo, nkhi = khi;
                const int top_limit = min(ak - 1, khi);
                const int thr = mind + 50;                    // keep while distance - min_distance <= 50
                if (klo < top_limit) {
                    int first = top_limit;
#pragma unroll
                    for (int j = kDiagSlots - 1; j >= 0; --j) {   // descending slots: the lowest qualifying k wins
                        if (j < jlo || j > jhi) continue;
                        const int k = kof(j);
                        if (k < top_limit && dist[j] <= thr) first = k;   // dist is INT_MAX outside [klo, khi]
                    }
                    nklo = wave_min_i32(first);
                }
                const int bottom_limit = max(ak + 1, nklo);
                if (khi > bottom_limit) {
                    int last = bottom_limit;
#pragma unroll
                    for (int j = 0; j < kDiagSlots; ++j) {         // ascending slots: the highest qualifying k wins
                        if (j < jlo || j > jhi) continue;
                        const int k = kof(j);
                        if (k > bottom_limit && dist[j] <= thr) last = k;
                    }
                    nkhi = -wave_min_i32(-last);
                }
                if (nklo > nkhi) flags |= GF_MNULL | GF_INULL | GF_DNULL;
                else { klo = nklo; khi = nkhi; }
            }
            AIM_DSTAMP(2);   // reduce
            if (lane == 0) {   // final descriptor of this score (after reduction)
                int16_t *me = meta + (score & (c.ring_m - 1)) * 4;
                me[0] = (int16_t)klo; me[1] = (int16_t)khi; me[2] = (int16_t)flags;
                if (BT) { int16_t *hm = hMeta(score); hm[0] = (int16_t)klo; hm[1] = (int16_t)khi; hm[2] = (int16_t)flags; }
            }
            fence();
            // affine_wfa_end_reached, wfa.c:210-230: M[ak] lives in lane (ak+kb)&63, slot (ak+kb)>>6
            if ((flags & GF_PRESENT) && !(flags & GF_MNULL) && klo <= ak && khi >= ak) {
                const int ia = ak + kb, ja = ia >> 6, la = ia & 63;
                int mend = 0;
#pragma unroll
                for (int j = 0; j < kDiagSlots; ++j)
                    if (j == ja) mend = __builtin_amdgcn_readlane(Mc[j], la);
                if (mend >= tlen) { done = true; final_score = score; break; }
            }
            if (score + 1 > MS) { done = true; final_score = score + 1; break; }   // wfa.c:368-376
            ++score;
            AIM_DSTAMP(3);   // descriptor write + end test

            // ---- affine_wfa_compute_next, wfa.c:268-340 ---------------------------------------------------------
            const int s_sub = score - X, s_o = score - OE;
            int sub_f = 0, o_f = 0, sub_lo = 1, sub_hi = -1, o_lo = 1, o_hi = -1;
            if (s_sub >= 0) {
                const int16_t *m = meta + (s_sub & (c.ring_m - 1)) * 4;
                sub_lo = __builtin_amdgcn_readfirstlane((int)m[0]); sub_hi = __builtin_amdgcn_readfirstlane((int)m[1]);
                sub_f = __builtin_amdgcn_readfirstlane((int)m[2]);
            }
            if (s_o >= 0) {
                const int16_t *m = meta + (s_o & (c.ring_m - 1)) * 4;
                o_lo = __builtin_amdgcn_readfirstlane((int)m[0]); o_hi = __builtin_amdgcn_readfirstlane((int)m[1]);
                o_f = __builtin_amdgcn_readfirstlane((int)m[2]);
            }
            const int e_f = flags;                       // s - gap_e is the score just finished (gap_e == 1)
            int e_lo = klo, e_hi = khi;
            const bool m_sub_null = (s_sub < 0) || !(sub_f & GF_PRESENT) || (sub_f & GF_MNULL);
            const bool m_o_null = (s_o < 0) || !(o_f & GF_PRESENT) || (o_f & GF_MNULL);
            const bool i_e_null = !(e_f & GF_PRESENT) || !(e_f & GF_HASI) || (e_f & GF_INULL);
            const bool d_e_null = !(e_f & GF_PRESENT) || !(e_f & GF_HASD) || (e_f & GF_DNULL);
            const bool i_out_null = m_o_null && i_e_null, d_out_null = m_o_null && d_e_null;
            if (m_sub_null && i_out_null && d_out_null) {
                flags = 0; klo = 0; khi = -1;
                continue;
            }
            if (m_sub_null) { sub_lo = 1; sub_hi = -1; }
            if (m_o_null) { o_lo = 1; o_hi = -1; }
            if (i_e_null && d_e_null) { e_lo = 1; e_hi = -1; }
            const int lo = min(min(sub_lo, o_lo), e_lo) - 1;
            const int hi = max(max(sub_hi, o_hi), e_hi) + 1;
            const int njlo = (lo + kb) >> 6, njhi = (hi + kb) >> 6;
            const int16_t *r_ms = mrow(s_sub < 0 ? 0 : s_sub), *r_mo = mrow(s_o < 0 ? 0 : s_o);
            // neighbours of the just-finished score: I at k-1 = lane-1 (lane 0: lane 63 of the slot below),
            // D at k+1 = lane+1 (lane 63: lane 0 of the slot above)
            int shI[kDiagSlots], shD[kDiagSlots];
#pragma unroll
            for (int j = 0; j < kDiagSlots; ++j) {
                if (j < njlo || j > njhi) { shI[j] = kGrpNull; shD[j] = kGrpNull; continue; }
                const int wrapI = (j > 0) ? __builtin_amdgcn_readlane(Ic[j - 1], 63) : kGrpNull;
                const int wrapD = (j + 1 < kDiagSlots) ? __builtin_amdgcn_readlane(Dc[j + 1], 0) : kGrpNull;
                shI[j] = __builtin_amdgcn_update_dpp(wrapI, Ic[j], 0x138, 0xf, 0xf, false);   // wave_shr:1
                shD[j] = __builtin_amdgcn_update_dpp(wrapD, Dc[j], 0x130, 0xf, 0xf, false);   // wave_shl:1
            }
            // the three M reads of every active slot are unconditional (index clamped into the row) and issued
            // together; the range tests of AFFINE_WAVEFRONT_COND_FETCH (common.h:121-124) become selects
            int vml[kDiagSlots], vmr[kDiagSlots], vms[kDiagSlots];
#pragma unroll
            for (int j = 0; j < kDiagSlots; ++j) {
                vml[j] = vmr[j] = vms[j] = kGrpNull;
                if (j < njlo || j > njhi) continue;
                const int k = kof(j);
                const int kc = min(max(k, -kb), kb), km = max(kc - 1, -kb), kp = min(kc + 1, kb);
                vml[j] = r_mo[km];
                vmr[j] = r_mo[kp];
                vms[j] = r_ms[kc];
            }
#pragma unroll
            for (int j = 0; j < kDiagSlots; ++j) {   // affine_wfa_compute_offsets, wfa.c:231-266
                if (j < njlo || j > njhi) continue;
                const int k = kof(j);
                int ins = -10;
                if (!i_out_null) {
                    const int ins_g = (!m_o_null && o_lo <= k - 1 && k - 1 <= o_hi) ? vml[j] : kGrpNull;
                    const int ins_i = (!i_e_null && e_lo <= k - 1 && k - 1 <= e_hi) ? shI[j] : kGrpNull;
                    ins = (ins_g == kGrpNull && ins_i == kGrpNull) ? kGrpNull : (int)(int16_t)(max(ins_g, ins_i) + 1);
                }
                int del = -10;
                if (!d_out_null) {
                    const int del_g = (!m_o_null && o_lo <= k + 1 && k + 1 <= o_hi) ? vmr[j] : kGrpNull;
                    const int del_d = (!d_e_null && e_lo <= k + 1 && k + 1 <= e_hi) ? shD[j] : kGrpNull;
                    del = max(del_g, del_d);
                }
                int sub = -10;
                if (!m_sub_null) sub = (sub_lo <= k && k <= sub_hi) ? (int)(int16_t)(vms[j] + 1) : kGrpNull;
                const bool in = k >= lo && k <= hi;
                Ic[j] = in ? ins : Ic[j];
                Dc[j] = in ? del : Dc[j];
                Mc[j] = in ? (int)(int16_t)max(del, max(sub, ins)) : Mc[j];
                if (BT && in) {
                    if (!i_out_null) hI(score)[k] = (int16_t)ins;
                    if (!d_out_null) hD(score)[k] = (int16_t)del;
                }
            }
            flags = GF_PRESENT | (i_out_null ? GF_INULL : GF_HASI) | (d_out_null ? GF_DNULL : GF_HASD);
            klo = lo; khi = hi;
            AIM_DSTAMP(4);   // compute_next
        }
        AIM_DSTAMP(5);
        // ---- traceback over the HBM history (lane 0), as in wfa_group.hpp ---------------------------------------
        if (BT) {
            __syncthreads();
            if (lane == 0 && bad == 0u && final_score <= MS) {
                char *ops = a.ops + (uint64_t)pair * 2 * rs;
                const int cap = 2 * rs, E = 1;
                enum { BT_M = 0, BT_I = 1, BT_D = 2 };
                auto put = [&](char ch) { if (begin_offset >= 0 && begin_offset < cap) ops[begin_offset] = ch; --begin_offset; };
                auto valid_loc = [&](int kk_, int off_) { const int v_ = off_ - kk_, h_ = off_; return v_ > 0 && v_ <= plen && h_ > 0 && h_ <= tlen; };
                int sc = final_score, k = ak;
                int offset = hM(sc)[k];
                bool valid = valid_loc(k, offset);
                int bt = BT_M;
                int v = offset - k, h = offset;
                while (v > 0 && h > 0 && sc > 0) {
                    if (!valid) {
                        valid = valid_loc(k, offset);
                        if (valid) {
                            if (k < ak) for (int i = k; i < ak; ++i) put('I');
                            else if (k > ak) for (int i = ak; i < k; ++i) put('D');
                        }
                    }
                    const int s_o = sc - OE, s_e = sc - E, s_x = sc - X;
                    int o_lo = 1, o_hi = -1, o_f = 0, e_lo = 1, e_hi = -1, e_f = 0, x_lo = 1, x_hi = -1, x_f = 0;
                    if (s_o >= 0) { const int16_t *m = hMeta(s_o); o_lo = m[0]; o_hi = m[1]; o_f = m[2]; }
                    if (s_e >= 0) { const int16_t *m = hMeta(s_e); e_lo = m[0]; e_hi = m[1]; e_f = m[2]; }
                    if (s_x >= 0 && bt == BT_M) { const int16_t *m = hMeta(s_x); x_lo = m[0]; x_hi = m[1]; x_f = m[2]; }
                    int del_ext = kGrpNull, del_open = kGrpNull, ins_ext = kGrpNull, ins_open = kGrpNull, misms = kGrpNull;
                    if (bt != BT_I) {
                        if ((e_f & GF_PRESENT) && !(e_f & GF_DNULL) && e_lo <= k + 1 && k + 1 <= e_hi) del_ext = hD(s_e)[k + 1];
                        if ((o_f & GF_PRESENT) && o_lo <= k + 1 && k + 1 <= o_hi) del_open = hM(s_o)[k + 1];
                    }
                    if (bt != BT_D) {
                        if ((e_f & GF_PRESENT) && (e_f & GF_HASI) && e_lo <= k - 1 && k - 1 <= e_hi) ins_ext = (int16_t)(hI(s_e)[k - 1] + 1);
                        if ((o_f & GF_PRESENT) && o_lo <= k - 1 && k - 1 <= o_hi) ins_open = (int16_t)(hM(s_o)[k - 1] + 1);
                    }
                    if (bt == BT_M) {
                        if ((x_f & GF_PRESENT) && x_lo <= k && k <= x_hi) misms = (int16_t)(hM(s_x)[k] + 1);
                    }
                    const int max_all = max(misms, max(max(ins_ext, ins_open), max(del_ext, del_open)));
                    if (bt == BT_M) {
                        const int num_matches = offset - max_all;
                        if (num_matches > 0) begin_offset -= num_matches;
                        offset = max_all;
                        v = offset - k;
                        h = offset;
                        if (v <= 0 || h <= 0) break;
                    }
                    char op;
                    if (max_all == del_ext) { op = 'D'; sc = s_e; ++k; bt = BT_D; }
                    else if (max_all == del_open) { op = 'D'; sc = s_o; ++k; bt = BT_M; }
                    else if (max_all == ins_ext) { op = 'I'; sc = s_e; --k; offset = (int16_t)(offset - 1); bt = BT_I; }
                    else if (max_all == ins_open) { op = 'I'; sc = s_o; --k; offset = (int16_t)(offset - 1); bt = BT_M; }
                    else if (max_all == misms) { op = 'X'; sc = s_x; offset = (int16_t)(offset - 1); }
                    else { status = AIM_PAIR_WFA_NO_LINK; break; }
                    if (valid) put(op);
                    v = offset - k;
                    h = offset;
                }
                if (status == AIM_PAIR_OK) {
                    if (sc == 0) { if (offset > 0) begin_offset -= offset; }
                    else { for (; v > 0; --v) put('D'); for (; h > 0; --h) put('I'); }
                    ++begin_offset;
                }
            }
        }
        if (lane == 0) {
            if (bad != 0u) {
                const uint32_t slot = atomicAdd(&todo[LANE_TODO_COUNT], 1u);
                todo[LANE_TODO_LIST + slot] = pair;
            } else {
                aim_result_t r;
                r.max_operations = plen + tlen;
                r.begin_offset = begin_offset;
                r.end_offset = plen + tlen;
                r.score = final_score;
                r.status = status;
                r.idx = rq.idx;
                a.res[pair] = r;
            }
        }
        have = nhave;
        pair = npair;
        AIM_DSTAMP(6);   // traceback + result
    }
#if AIM_DIAG_STAMPS
    if (lane == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(a.scratch + 256) + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 8; ++i) dbg[i] = dsum[i];
    }
#endif
}

// Eligible: gap_e == 1, MAX_SCORE <= 254 (8 slots of 64 diagonals), READ_SIZE <= 4096.
inline bool wfa_diag_plan(const aim_params_t &p, uint32_t n_pairs, GroupCfg *c, uint32_t *grid, size_t *lds, size_t *hist_bytes)
{
    if (p.algo != AIM_ALGO_WFA || p.gap_e != 1) return false;
    if (2 * p.max_score + 3 > kDiagSlots * kWave || p.read_size > 4096) return false;
    const int R = p.mismatch > p.gap_o + p.gap_e ? p.mismatch : p.gap_o + p.gap_e;
    int ring_m = 1;
    while (ring_m <= R) ring_m *= 2;
    if (ring_m > 32) return false;
    c->kbias = p.max_score + 1;
    c->wcap = 2 * p.max_score + 3;
    c->ring_m = ring_m;
    c->ring_e = 2;
    c->np = (p.read_size + 15) / 16 + 1;
    c->pair_dwords = 0;
    c->rows_per_wave = 1;
    const size_t rows_bytes = (size_t)2 * (((size_t)p.read_size + 15) / 16) * 16 + 8;
    *lds = rows_bytes + (size_t)((ring_m * c->wcap * 2 + ring_m * 8 + 3) / 4 + 2 * c->np) * 4 + 64;
    uint32_t per_cu = (uint32_t)std::min<size_t>(12, lds_workgroups_per_cu(*lds));
    if (const char *e = getenv("AIM_DIAG_PER_CU")) per_cu = (uint32_t)std::max(1, atoi(e));
    uint32_t gr = 256 * per_cu;
    const uint32_t need = ((n_pairs + 7u) / 8u) * 8u;
    if (gr > need) gr = need < 8u ? 8u : need;
    *grid = gr;
    c->hist_stride = (p.max_score + 2) * (3 * c->wcap + 4);
    *hist_bytes = (p.flags & AIM_FLAG_BACKTRACE) ? (((size_t)gr * c->hist_stride * 2 + 255) & ~(size_t)255) : 0;
    return true;
}

inline void wfa_diag_launch(const aim_params_t &p, const GroupCfg &c, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s)
{
    const bool red = p.flags & AIM_FLAG_REDUCE, bt = p.flags & AIM_FLAG_BACKTRACE;
    if (red && bt) hipLaunchKernelGGL((wfa_diag_kernel<true, true>), dim3(grid), dim3(kWave), lds, s, ka, c);
    else if (red) hipLaunchKernelGGL((wfa_diag_kernel<true, false>), dim3(grid), dim3(kWave), lds, s, ka, c);
    else if (bt) hipLaunchKernelGGL((wfa_diag_kernel<false, true>), dim3(grid), dim3(kWave), lds, s, ka, c);
    else hipLaunchKernelGGL((wfa_diag_kernel<false, false>), dim3(grid), dim3(kWave), lds, s, ka, c);
}

}  // namespace aim
