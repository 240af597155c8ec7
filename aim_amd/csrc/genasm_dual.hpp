// genasm_dual.hpp -- GenASM for long reads, TWO PAIRS PER WAVEFRONT (round 4).
//
// PARITY UNPINNED like genasm_wave.hpp (same published algorithm, same [spec] choices, oracle/genasm_oracle.c).
//
// genasm_wave_kernel<.., LONG> keeps one pair per wavefront and sweeps GenASM-DC with lanes 0..15 (one 16-lane DPP row = the 16 error
// levels of the fast path): 90 % of its 3.64 M VALU instructions per 100-kb pair run with 16 of 64 lanes, and at 16 wavefronts per CU it is
// VALU-issue-bound (0.97). Here lanes 0..31 belong to pair A and lanes 32..63 to pair B: the sweep runs in DPP rows 0 and 2 with ONE
// instruction stream for both pairs, the traceback looks at 32 candidate cells per pair and iteration instead of 64 (a window commits 40
// characters), and everything that was wave-uniform per window -- level, consumed characters, op count -- is a pair of scalars.
//
// Only REGULAR windows take this path: m = n = 64 characters and not the pair's last window (for 100-kb reads: 2 499 of 2 500). A window that is
// irregular, or finds no alignment within 15 edits, runs alone through ga_window_wide (all 64 levels over the whole wavefront, columns in the
// wavefront's HBM slab: genasm_wave_kernel's slow path, verbatim) while the other half waits.
#pragma once

#include "genasm_wave.hpp"

namespace aim {

constexpr int kGdHalf = kGlCols * 16 + kGlPm;          // uint64 per half: kept columns [56 .. -15] x 16 levels + pattern masks of columns 79 .. -16
constexpr size_t kGdLdsBytes = (size_t)2 * kGdHalf * 8 + 2 * 2 * kGaW;   // two halves + their window characters (p, t)

// One window of ONE pair over the whole wavefront, 64 levels, columns in the HBM slab Rg: the slow path of genasm_wave_kernel. Emits the
// window's ops at ops[nops ..] and returns consumed text / pattern characters, op count and edits through the references.
template <bool BT>
__device__ __forceinline__ void ga_window_wide(const unsigned char *gP, const unsigned char *gT, int pi, int ti, int plen, int tlen, char *ops, int nops, int cap,
                                               uint64_t *Rg, int lane, int &ca, int &cb, int &wn, int &edits, int &status)
{
    constexpr uint64_t ONES = ~0ull;
    const int m = min(kGaW, plen - pi), n = min(kGaW, tlen - ti);
    const bool last = (m == plen - pi) && (n == tlen - ti);
    const int prev = lane < m ? (int)gP[pi + m - 1 - lane] : 0x100;
    const int pfwd = lane < m ? (int)gP[pi + lane] : 0x200;
    const int tfwd = lane < n ? (int)gT[ti + lane] : 0x300;
    uint64_t mypm = ONES;
    for (uint64_t rest = __ballot(lane < n); rest;) {
        const int c = __builtin_amdgcn_readlane(tfwd, (int)__builtin_ctzll(rest));
        const uint64_t pm = ~__ballot(prev == c);
        if (tfwd == c) mypm = pm;
        rest &= ~__ballot(tfwd == c);
    }
    const uint64_t R = ga_dc<true>(n, lane, mypm, Rg);
    const uint64_t hit = __ballot(!((R >> (m - 1)) & 1ull));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront reads its own slab back below
    int d = hit ? (int)__builtin_ctzll(hit) : -1;
    ca = cb = wn = 0;
    edits = 0;
    int opsA = 'M', opsB = 'M';
    auto put = [&](int ch) {
        opsA = lane == wn ? ch : opsA;
        opsB = lane + 64 == wn ? ch : opsB;
        ++wn;
    };
    if (d < 0) {   // [spec] no alignment of this window within 63 edits: diagonal steps
        const int steps = min(min(m, n), kGaCommit);
        const bool x = lane < steps && pfwd != tfwd;
        opsA = x ? 'X' : opsA;
        edits += __builtin_popcountll(__ballot(x));
        ca = cb = wn = steps;
    } else {
        for (;;) {
            const int ai = ca + lane, bi = cb + lane;
            const bool inr = bi < m && ai < n && (last || (ai < kGaCommit && bi < kGaCommit));
            const int aic = min(ai, n - 1), bic = min(bi, kGaW - 1);
            const int dm1 = d > 0 ? d - 1 : 0;
            const uint64_t rn_d = Rg[(aic + 1) * 64 + d], rn_dm1 = Rg[(aic + 1) * 64 + dm1], rc_dm1 = Rg[aic * 64 + dm1];
            const int pch = __builtin_amdgcn_ds_bpermute(bic << 2, pfwd), tch = __builtin_amdgcn_ds_bpermute(aic << 2, tfwd);
            const int q1 = m - 2 - bi, q0 = m - 1 - bi;
            auto clr = [&](uint64_t r, int q) -> bool { return q < 0 || !((r >> (q & 63)) & 1ull); };
            const bool cm = inr && pch == tch && clr(rn_d, q1);
            int code = 0;
            if (d > 0) code = clr(rn_dm1, q1) ? 'X' : clr(rc_dm1, q1) ? 'D' : clr(rn_dm1, q0) ? 'I' : 0;
            const uint64_t bad = ~__ballot(cm);
            const int run = bad ? (int)__builtin_ctzll(bad) : 64;
            wn += run; ca += run; cb += run;
            if (cb == m) break;
            if (!last && (ca >= kGaCommit || cb >= kGaCommit)) break;
            if (ca == n) { put('D'); ++cb; --d; ++edits; continue; }
            const int op = __builtin_amdgcn_readlane(code, run);
            if (op == 0) { status = AIM_PAIR_WFA_NO_LINK; break; }
            put(op);
            ca += op != 'D';
            cb += op != 'I';
            --d; ++edits;
        }
    }
    if (BT) {
        if (lane < wn && nops + lane < cap) ops[nops + lane] = (char)opsA;
        if (lane + 64 < wn && nops + 64 + lane < cap) ops[nops + 64 + lane] = (char)opsB;
    }
}

template <bool BT>
__global__ __launch_bounds__(64) void genasm_dual_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    constexpr uint64_t ONES = ~0ull;
    constexpr int N = kGaW;                               // a regular window: m = n = 64
    const int lane = threadIdx.x, half = lane >> 5, hl = lane & 31;
    const int rs = a.p.read_size, cap = 2 * rs;
    uint64_t *Rh = reinterpret_cast<uint64_t *>(smem) + half * kGdHalf;          // this lane's half: kept columns + pattern masks
    unsigned char *cwP = reinterpret_cast<unsigned char *>(smem) + (size_t)2 * kGdHalf * 8 + half * 2 * kGaW, *cwT = cwP + kGaW;   // its window's characters
    uint64_t *Rg = reinterpret_cast<uint64_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave);

    // per half (wave-uniform scalars)
    uint32_t pair[2] = {0, 0};
    int plen[2] = {0, 0}, tlen[2] = {0, 0}, pi[2] = {0, 0}, ti[2] = {0, 0}, nops[2] = {0, 0}, dist[2] = {0, 0}, status[2] = {0, 0};
    uint32_t idx[2] = {0, 0};
    bool live[2] = {false, false};
    uint32_t it = 0;
    bool more = true;
    auto fetch = [&](int h) {
        live[h] = false;
        while (more) {
            uint32_t pr;
            more = xcd_unit(a.n_pairs, it, &pr);
            if (!more) break;
            ++it;
            const aim_request_t rq = load_request(a, pr);
            pair[h] = pr;
            plen[h] = __builtin_amdgcn_readfirstlane(rq.pattern_len);
            tlen[h] = __builtin_amdgcn_readfirstlane(rq.text_len);
            idx[h] = (uint32_t)__builtin_amdgcn_readfirstlane((int)rq.idx);
            pi[h] = ti[h] = nops[h] = dist[h] = 0;
            status[h] = AIM_PAIR_OK;
            live[h] = true;
            break;
        }
    };
    auto seqP = [&](int h) { return reinterpret_cast<const unsigned char *>(a.patterns + (uint64_t)pair[h] * rs); };
    auto seqT = [&](int h) { return reinterpret_cast<const unsigned char *>(a.texts + (uint64_t)pair[h] * rs); };
    auto opsof = [&](int h) { return BT ? a.ops + (uint64_t)pair[h] * 2 * rs : nullptr; };
    auto finish = [&](int h) {   // one sequence is exhausted (or the pair failed): the rest of the other is gaps; result; next pair
        if (status[h] == AIM_PAIR_OK) {
            const int rp = plen[h] - pi[h], rt = tlen[h] - ti[h];
            if (BT) {
                char *ops = opsof(h);
                for (int i = lane; i < rp; i += kWave) if (nops[h] + i < cap) ops[nops[h] + i] = 'D';
                for (int i = lane; i < rt; i += kWave) if (nops[h] + rp + i < cap) ops[nops[h] + rp + i] = 'I';
            }
            nops[h] += rp + rt;
            dist[h] += rp + rt;
        }
        if (lane == 0) {
            aim_result_t r;
            r.max_operations = plen[h] + tlen[h];
            r.begin_offset = 0;
            r.end_offset = nops[h];
            r.score = dist[h];
            r.status = status[h];
            r.idx = idx[h];
            store_result(a, pair[h], r);
        }
        fetch(h);
    };
    fetch(0);
    fetch(1);

    while (live[0] || live[1]) {
        // pairs that are done (a sequence exhausted, or failed) leave; irregular windows run alone
        bool again = false;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!live[h]) continue;
            if (status[h] != AIM_PAIR_OK || pi[h] >= plen[h] || ti[h] >= tlen[h]) { finish(h); again = true; continue; }
            const int m = min(kGaW, plen[h] - pi[h]), n = min(kGaW, tlen[h] - ti[h]);
            const bool last = (m == plen[h] - pi[h]) && (n == tlen[h] - ti[h]);
            if (m != N || n != N || last) {
                int ca, cb, wn, ed;
                ga_window_wide<BT>(seqP(h), seqT(h), pi[h], ti[h], plen[h], tlen[h], opsof(h), nops[h], cap, Rg, lane, ca, cb, wn, ed, status[h]);
                nops[h] += wn; pi[h] += cb; ti[h] += ca; dist[h] += ed;
                again = true;
            }
        }
        if (again) continue;
        const bool on = live[half];                       // (both live halves hold a regular window here)

        // ---- window characters: lane hl holds positions hl and hl + 32 of its half's window
        {
            const unsigned char *gp = (half ? seqP(1) : seqP(0)) + (half ? pi[1] : pi[0]);
            const unsigned char *gt = (half ? seqT(1) : seqT(0)) + (half ? ti[1] : ti[0]);
            __syncthreads();                              // (single wavefront: the previous window's walk is done with the character arrays)
            if (on) {
                cwP[hl] = gp[hl]; cwP[hl + 32] = gp[hl + 32];
                cwT[hl] = gt[hl]; cwT[hl + 32] = gt[hl + 32];
            }
            __syncthreads();
        }
        const int tf0 = on ? (int)cwT[hl] : 0x300, tf1 = on ? (int)cwT[hl + 32] : 0x301;
        const int pv0 = on ? (int)cwP[N - 1 - hl] : 0x100, pv1 = on ? (int)cwP[N - 1 - (hl + 32)] : 0x101;   // reversed pattern: positions hl, hl + 32
        // ---- pattern masks of my two text columns: bit i = 0 <=> p[m-1-i] == t[column]; one pass per distinct character of either window
        uint64_t mypm0 = ONES, mypm1 = ONES;
        {
            bool a0 = !on, a1 = !on;
            for (;;) {
                const uint64_t r0 = __ballot(!a0), r1 = __ballot(!a1);
                if (!(r0 | r1)) break;
                const int c = r0 ? __builtin_amdgcn_readlane(tf0, (int)__builtin_ctzll(r0)) : __builtin_amdgcn_readlane(tf1, (int)__builtin_ctzll(r1));
                const uint64_t bLo = __ballot(pv0 == c), bHi = __ballot(pv1 == c);      // bit of lane l: position (l & 31) [+ 32] of half (l >> 5)
                const uint64_t pmA = ~((bLo & 0xffffffffull) | (bHi << 32)), pmB = ~((bLo >> 32) | (bHi & 0xffffffff00000000ull));
                const uint64_t pm = half ? pmB : pmA;
                if (tf0 == c) { mypm0 = pm; a0 = true; }
                if (tf1 == c) { mypm1 = pm; a1 = true; }
            }
        }
        // ---- GenASM-DC, levels 0..15 of both halves (ga_dc16_long with n = m = 64; lanes hl < 16 of each half = DPP rows 0 and 2)
        int dlev[2] = {-1, -1};
        {
            if (on) { Rh[gl_pm(hl)] = mypm0; Rh[gl_pm(hl + 32)] = mypm1; }
            const bool sw = hl < 16;
            if (sw) Rh[gl_pm(N + hl)] = ONES;
            uint64_t cur = ONES << hl;                   // R_n[d]
            const uint64_t lane0 = hl == 0 ? ONES : 0ull;
            const uint64_t endbit = 1ull << (N - 1);
            constexpr int nA = N - 1 - kGlTop;            // steps before level 0 reaches column 41
            uint64_t *rp = Rh + gl_slot(N - 1 - nA + hl) + hl;
            const uint64_t *pp = Rh + gl_pm(N - 1 + hl);
            auto shr1 = [](uint64_t v) -> uint64_t {
                const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x111, 0xf, 0xf, true);
                const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x111, 0xf, 0xf, true);
                return ((uint64_t)hi << 32) | lo;
            };
            if (sw) {
                auto step = [&](const uint64_t &nb_prev, uint64_t &nb_cur, const uint64_t &pm_now, uint64_t &pm_next, int k, auto st) {
                    pm_next = pp[k + 1];
                    nb_cur = shr1(cur);
                    const uint64_t t = (((nb_prev & nb_cur) << 1) & nb_prev) | lane0;
                    cur = ((cur << 1) | pm_now) & t;
                    if constexpr (decltype(st)::value) rp[k * 16] = cur;
                };
                uint64_t nbA = shr1(cur), nbB, pmA = pp[0], pmB;
                int u = 0;
                for (; u + 2 <= nA; u += 2) {
                    step(nbA, nbB, pmA, pmB, 0, std::false_type{});
                    step(nbB, nbA, pmB, pmA, 1, std::false_type{});
                    pp += 2;
                }
                if (u < nA) {
                    step(nbA, nbB, pmA, pmB, 0, std::false_type{});
                    nbA = nbB; pmA = pmB;
                    pp += 1;
                }
                for (u = nA; u + 2 <= N - 1; u += 2) {
                    step(nbA, nbB, pmA, pmB, 0, std::true_type{});
                    step(nbB, nbA, pmB, pmA, 1, std::true_type{});
                    rp += 2 * 16;
                    pp += 2;
                }
                if (u < N - 1) {
                    step(nbA, nbB, pmA, pmB, 0, std::true_type{});
                    nbA = nbB; pmA = pmB;
                    rp += 16; pp += 1;
                }
                bool need0 = live[0], need1 = live[1];
                for (u = N - 1; u < N + 15; ++u) {       // level u - (n-1) completes column 0 in this step
                    step(nbA, nbB, pmA, pmB, 0, std::true_type{});
                    nbA = nbB; pmA = pmB;
                    rp += 16; pp += 1;
                    const uint64_t hb = __ballot(hl == u - (N - 1) && !(cur & endbit));
                    if (need0 && (hb & 0xffffull)) { dlev[0] = u - (N - 1); need0 = false; }
                    if (need1 && (hb >> 32)) { dlev[1] = u - (N - 1); need1 = false; }
                    if (!need0 && !need1) break;
                }
            }
            dlev[0] = __builtin_amdgcn_readfirstlane(dlev[0]);
            dlev[1] = __builtin_amdgcn_readfirstlane(dlev[1]);
        }
        // a window that needs more than 15 edits: alone through the 64-level path (its half sits out the walk below)
        bool walkh[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            walkh[h] = live[h] && dlev[h] >= 0;
            if (live[h] && dlev[h] < 0) {
                int ca, cb, wn, ed;
                ga_window_wide<BT>(seqP(h), seqT(h), pi[h], ti[h], plen[h], tlen[h], opsof(h), nops[h], cap, Rg, lane, ca, cb, wn, ed, status[h]);
                nops[h] += wn; pi[h] += cb; ti[h] += ca; dist[h] += ed;
            }
        }
        if (!walkh[0] && !walkh[1]) continue;
        // ---- GenASM-TB of both halves: lane hl looks at the cell its half's walk would reach after hl matches (genasm_wave.hpp's walk, 32 cells per
        // half and iteration). A window's ops: three registers per lane (ops hl, 32 + hl, 64 + hl), pre-set to 'M'.
        int d[2] = {dlev[0], dlev[1]}, ca[2] = {0, 0}, cb[2] = {0, 0}, wn[2] = {0, 0};
        bool fin[2] = {!walkh[0], !walkh[1]};
        int o0 = 'M', o1 = 'M', o2 = 'M';
        auto put = [&](int h, int ch) {
            const int at = h * 32 + (wn[h] & 31), reg = wn[h] >> 5;
            o0 = (lane == at && reg == 0) ? ch : o0;
            o1 = (lane == at && reg == 1) ? ch : o1;
            o2 = (lane == at && reg == 2) ? ch : o2;
            ++wn[h];
        };
        while (!(fin[0] && fin[1])) {
            const int cav = half ? ca[1] : ca[0], cbv = half ? cb[1] : cb[0], dv = half ? d[1] : d[0];
            const bool act = half ? !fin[1] : !fin[0];
            const int ai = cav + hl, bi = cbv + hl;
            const bool inr = act && bi < N && ai < N && ai < kGaCommit && bi < kGaCommit;
            const int aic = min(ai, kGaCommit), bic = min(bi, N - 1);            // (columns above 41 were not kept; lanes clamped there are outside the commit range)
            const int dm1 = dv > 0 ? dv - 1 : 0;
            const uint64_t rn_d = Rh[gl_slot(aic + 1) + dv], rn_dm1 = Rh[gl_slot(aic + 1) + dm1], rc_dm1 = Rh[gl_slot(aic) + dm1];
            const int pch = cwP[bic], tch = cwT[min(aic, N - 1)];
            const int q1 = N - 2 - bi, q0 = N - 1 - bi;
            auto clr = [&](uint64_t r, int q) -> bool { return q < 0 || !((r >> (q & 63)) & 1ull); };
            const bool cm = inr && pch == tch && clr(rn_d, q1);
            int code = 0;
            if (dv > 0) code = clr(rn_dm1, q1) ? 'X' : clr(rc_dm1, q1) ? 'D' : clr(rn_dm1, q0) ? 'I' : 0;
            const uint64_t bm = __ballot(cm);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (fin[h]) continue;
                const uint32_t bad = ~(uint32_t)(bm >> (32 * h));
                const int run = bad ? (int)__builtin_ctz(bad) : 32;
                wn[h] += run; ca[h] += run; cb[h] += run;
                if (cb[h] == N) { fin[h] = true; continue; }
                if (ca[h] >= kGaCommit || cb[h] >= kGaCommit) { fin[h] = true; continue; }
                if (ca[h] == N) { put(h, 'D'); ++cb[h]; --d[h]; ++dist[h]; continue; }
                if (run == 32) continue;                  // 32 matches and still inside the commit range: the run goes on
                const int op = __builtin_amdgcn_readlane(code, 32 * h + run);
                if (op == 0) { status[h] = AIM_PAIR_WFA_NO_LINK; fin[h] = true; continue; }   // cannot happen (the recurrence guarantees one rule applies)
                put(h, op);
                ca[h] += op != 'D';
                cb[h] += op != 'I';
                --d[h]; ++dist[h];
            }
        }
        // ---- the windows' ops leave as coalesced byte stores; advance
        if (BT) {
            char *ops = half ? opsof(1) : opsof(0);
            const int base = half ? nops[1] : nops[0], w = half ? wn[1] : wn[0];
            const bool wr = half ? walkh[1] : walkh[0];
            if (wr && hl < w && base + hl < cap) ops[base + hl] = (char)o0;
            if (wr && hl + 32 < w && base + 32 + hl < cap) ops[base + 32 + hl] = (char)o1;
            if (wr && hl + 64 < w && base + 64 + hl < cap) ops[base + 64 + hl] = (char)o2;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (walkh[h]) { nops[h] += wn[h]; pi[h] += cb[h]; ti[h] += ca[h]; }
    }
}

// The LONG variant pays one 64-level pass per pair (the last window) for 16 instead of 11 wavefronts per CU. Same box, kernel ms of the
// standard variant at 8 / at 11 per CU / LONG at 16 (tools/ga_sweep.py, e = 10 %): l=100 2.12 / 1.86 / 4.83; l=300 3.60 / 3.19 / 3.16;
// l=1000 6.07 / 5.46 / 4.48; l=2000 6.19 / 5.38 / 4.33; l=3000 4.69 / 4.31 / 3.24; l=5000 7.74 / 7.12 / 5.27 -- LONG from ~15 windows per pair up.
inline bool genasm_long(const aim_params_t &p, const Knobs &kn)
{
    return kn.ga_long >= 0 ? kn.ga_long != 0 : p.read_size >= 640;
}

// Two pairs per wavefront (genasm_dual_kernel) wherever the LONG variant is used; AIM_GA_DUAL=0 keeps one pair per wavefront (A/B).
inline bool genasm_dual(const aim_params_t &p, const Knobs &kn) { return genasm_long(p, kn) && kn.ga_dual != 0; }

inline void genasm_plan(const aim_params_t &p, const Knobs &kn, uint32_t n_pairs, uint32_t *grid, uint32_t *block, size_t *lds)
{
    const bool lg = genasm_long(p, kn);
    *block = kWave;
    if (genasm_dual(p, kn)) {
        *lds = kGdLdsBytes;   // 20 224 B = 16 LDS granules: exactly 8 wavefronts = 16 pairs per CU (4 096 pairs resident on the chip)
        uint32_t per_cu = (uint32_t)std::min<size_t>(16, lds_workgroups_per_cu(*lds));
        if (kn.ga_per_cu > 0) per_cu = (uint32_t)std::min<size_t>((size_t)kn.ga_per_cu, lds_workgroups_per_cu(*lds));
        uint32_t g = resident_grid(kn, per_cu);
        const uint32_t need = ((((n_pairs + 1u) / 2u) + 7u) / 8u) * 8u;
        if (g > need) g = need < 8u ? 8u : need;
        *grid = g;
        return;
    }
    *lds = lg ? (size_t)(kGlCols * 16 + kGlPm) * 8 + 64 : (size_t)kGaCols * kGaSlots * 8 + 64;
    uint32_t per_cu = (uint32_t)std::min<size_t>(16, lds_workgroups_per_cu(*lds));   // (the standard variant's 13 KB: 11; capped at 8 until round 3)
    if (kn.ga_per_cu > 0) per_cu = (uint32_t)std::min<size_t>((size_t)kn.ga_per_cu, lds_workgroups_per_cu(*lds));   // residency sweeps
    uint32_t g = resident_grid(kn, per_cu);
    const uint32_t need = ((n_pairs + 7u) / 8u) * 8u;
    if (g > need) g = need < 8u ? 8u : need;
    *grid = g;
}

// Kernels are instantiated in ONE translation unit (tu_*.hip defines AIM_TU_GENASM); every other includer sees the declaration only.
#ifdef AIM_TU_GENASM
void genasm_launch(const aim_params_t &p, const Knobs &kn, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s)
{
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    if (genasm_dual(p, kn)) {
        if (bt) hipLaunchKernelGGL((genasm_dual_kernel<true>), dim3(grid), dim3(kWave), lds, s, ka);
        else hipLaunchKernelGGL((genasm_dual_kernel<false>), dim3(grid), dim3(kWave), lds, s, ka);
    } else if (genasm_long(p, kn)) {
        if (bt) hipLaunchKernelGGL((genasm_wave_kernel<true, true>), dim3(grid), dim3(kWave), lds, s, ka);
        else hipLaunchKernelGGL((genasm_wave_kernel<false, true>), dim3(grid), dim3(kWave), lds, s, ka);
    } else {
        if (bt) hipLaunchKernelGGL((genasm_wave_kernel<true, false>), dim3(grid), dim3(kWave), lds, s, ka);
        else hipLaunchKernelGGL((genasm_wave_kernel<false, false>), dim3(grid), dim3(kWave), lds, s, ka);
    }
}
#else
void genasm_launch(const aim_params_t &p, const Knobs &kn, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s);
#endif

}  // namespace aim
