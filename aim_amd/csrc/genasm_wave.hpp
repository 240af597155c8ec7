// genasm_wave.hpp -- GenASM (bit-vector approximate string matching, windowed traceback): ONE PAIR PER WAVEFRONT, one window at a
// time. BASELINE config 5.
//
// PARITY UNPINNED: AIM's GenASM is an un-vendored, un-pinned submodule (/root/reference/.gitmodules:1-3, empty directory,
// no call sites or tests).  This kernel implements the PUBLISHED algorithm (Senol Cali et al., MICRO 2020: GenASM-DC,
// Algorithm 1; GenASM-TB and the W = 64 / O = 24 windows, Section 6) exactly as oracle/genasm_oracle.c restates it --
// every open choice is fixed there and marked [spec] -- and is tested bit for bit against that restatement.
//
// A pair of 100 kb reads is ~2 500 dependent windows (the next window starts where this one's traceback committed), so what
// matters is the latency of one window and the parallelism has to come from inside it. GenASM-DC (Algorithm 1), columns a = n-1 .. 0:
//     R_a[d] = ((R_{a+1}[d] << 1) | PM[t_a])  &  (R_{a+1}[d-1] << 1)  &  R_{a+1}[d-1]  &  (R_a[d-1] << 1)
//              match                             substitution            text-only edit   pattern-only edit
// Almost every window aligns within 15 edits (at e = 10 % a 64-character window carries ~6). Those windows take the FAST PATH
// (rounds 3-4): banded 32-bit words, the lanes are the window's text COLUMNS and an error level is one scan over the wavefront;
// the traceback keeps the lanes on their columns. A window that needs 16 .. 63 edits, or none at all [spec], takes the FULL-WIDTH
// PATH: the same mapping on full 64-bit vectors (the composition carries a shift), rows of levels in a per-wavefront slab of HBM
// scratch. (Until round 4 it was round 2's design: lanes = error levels, a skewed sweep with one DPP wave_shr per column.)
//
// FAST PATH.
// (1) BANDED bit-vectors. The fast path stops at 15 edits, and an alignment of the window with <= 15 edits never leaves the
// diagonals |a - i| <= 15 (a = text column, i = pattern position; the window's walk starts at (0, 0) and every edit moves it one diagonal
// and costs one level). Follow one bit of the recurrence: bit q (pattern position i = m-1-q) of R_a[d] reads bit q-1 of column a+1 (match,
// substitution: the SAME diagonal), bit q-1 of R_a[d-1] and bit q of R_{a+1}[d-1] (one diagonal away, one level down). So a bit on diagonal k
// of level l depends on diagonals k-j .. k+j of level l-j only, and what the hit test (diagonal 0, level d0 <= 15) and the traceback (a cell
// on diagonal k with |k| <= d0 - d at level d, its neighbours on k +- 1 at level d-1) read is a function of diagonals -15 .. +15 -- 31 bits.
// The sweep therefore keeps, per column a, ONE 32-bit word whose bit b is bit q = b + (m-16-a) of the full vector (diagonal b - 15, i.e.
// pattern position i = a + 15 - b): in these coordinates the "<< 1" of the terms that come from column a+1 disappears (the band slides with
// the column), the pattern-only edit keeps its "<< 1" and the text-only edit becomes ">> 1". Bits shifted in at either end are garbage that
// moves one bit inwards per level -- at level l bits b < l and b > 31 - l -- and the bits read at level l are l .. 30 - l (shown above).
// Positions past the pattern's end (i >= m) are in the band for columns > m - 16; they hold 0 ("matches") as in the full vector, where
// "<< 1" shifts them in; positions before its start (i < 0) only ever receive, never give (information flows towards lower i and lower a).
// The words are kept COMPLEMENTED (c = ~R, bit set = "aligns"): AND-of-ORs becomes OR-of-ANDs, which the ISA has three-operand forms for:
//     c_a[d] = (c_{a+1}[d] & eq_a)  |  c_{a+1}[d-1]  |  (c_a[d-1] << 1)  |  (c_{a+1}[d-1] >> 1),   eq_a = ~PM[t_a] in band coordinates
// Results are bit-identical to the full-width sweep wherever anything reads them (tests/test_genasm.py, tools/fuzz_parity.py --focus genasm).
// (2) The sweep as a SCAN over columns. With no shift on the match term one level is a bitwise linear recurrence along the columns,
//     c_a = (c_{a+1} & eq_a) | g_a   with   g_a[d] = c_{a+1}[d-1] | (c_a[d-1] << 1) | (c_{a+1}[d-1] >> 1),
// i.e. a composition of the maps x -> (x & e) | g, which is associative: (e1, g1) then (e2, g2) = (e1 & e2, (g1 & e2) | g2). Lane j owns column
// n-1-j and a level is ONE inclusive scan over the wavefront -- 4 DPP row_shr steps, row_bcast:15, row_bcast:31 -- instead of a walk down the
// columns; the products of the eq words that the steps need do not depend on the level and are computed once per window (6 registers), so a
// level costs: the neighbour column's word of the level below (one DPP wave_shr), g (3), 6 x (DPP move + and-or), the initial column folded in
// (1), one store of the 64 columns, the hit test on column 0's lane -- ~29 instructions with all lanes busy -- and levels beyond the first hit
// are never computed. 16 levels x 64 columns x 4 B = 4 KB of LDS per wavefront.
// (3) GenASM-TB with the lanes still on their columns: see walk_cols in the kernel.
// History of the fast path, 4 096 pairs of 100 kb (profiles/NOTES.md R4.5): 64-bit skewed 16-lane sweep 25.4 ms (round 3) -> banded words
// 13.9 -> scan 9.1 -> column-bound traceback 7.9 ms. Round 4 also made it the path of EVERY window (irregular m != n, the pair's last):
// round 3's standard variant (64-bit words, lanes = levels, 13 KB of LDS) is gone.
// Integer / bit work only; HBM sees each sequence byte once and the ops once.
#pragma once

#include <type_traits>
#include "aim_device.hpp"

namespace aim {

constexpr int kGaW = 64;        // window
constexpr int kGaCommit = 40;   // W - O

// Lanes of the one wavefront hand values to each other through LDS / the HBM slab (columns written by one lane are read by another). The
// hardware executes a wavefront's LDS instructions in order, but the COMPILER reasons per thread and may move a load above a store to a
// different address of the same thread: this stops it (no instruction is emitted).
__device__ __forceinline__ void ga_lds_order() { asm volatile("" ::: "memory"); }

// ---- full-width path (16 .. 63 edits): lanes = text columns like the fast path, full 64-bit complemented vectors (no band: an alignment with
// up to 63 edits can sit anywhere). In full coordinates the match term keeps its shift, and one column is the map
//     x -> (((x << 1) | 1) & EQ_a) | G_a,      G_a[d] = (C_{a+1}[d-1] << 1) | C_{a+1}[d-1] | (C_a[d-1] << 1) | 1   (d >= 1; G_a[0] = 0)
// (C = ~R; "~(X << 1) = (~X << 1) | 1"), a member of the family x -> ((x << k) & A) | B, which is closed under composition:
//     (k1, A1, B1) then (k2, A2, B2)  =  (k1 + k2, (A1 << k2) & A2, ((B1 << k2) & A2) | B2).
// So a level is again ONE inclusive scan over the wavefront; the shift of a step is the length of the lane's own segment at that step -- 1, 2,
// 4, 8 for the row_shr steps, (lane & 15) + 1 and (lane & 31) + 1 for the two row broadcasts -- and the A parts do not depend on the level
// (computed once per window). C_a[d] = ((C_n[d] << (j + 1)) & A_j) | B_j for lane j = column n-1-j, C_n[d] = the d lowest bits. Levels are
// computed until the first one whose column 0 has bit m-1 set (random sequences: ~30 of 64) and leave as one coalesced 512-B store each into
// this wavefront's HBM slab ([level][lane]); the walk reads a level's row back with one coalesced load per edit.
// Round 2-4's version (lanes = error levels, skewed sweep, n + 63 steps whatever the hit level, the walk gathering three slab columns per
// iteration) needed 54 k cycles per window against 5 k on the fast path: a pair that loses the diagonal -- 1 in ~4 000 at l = 100 000, e = 10 %,
// every later window random against random -- kept ONE wavefront busy for 45 ms next to a batch that takes 8 (profiles/NOTES.md R4.6).
template <int CTRL>
__device__ __forceinline__ uint32_t gl_dppz(uint32_t v)   // all four rows, lanes without a source receive 0 (bound_ctrl: no fill register to set up)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ uint64_t gw_dppz(uint64_t v)
{
    return ((uint64_t)gl_dppz<CTRL>((uint32_t)(v >> 32)) << 32) | gl_dppz<CTRL>((uint32_t)v);
}
template <int CTRL, int ROWS>
__device__ __forceinline__ uint64_t gw_dpp(uint64_t fill, uint64_t v)   // lanes without a source (or outside the row mask) receive `fill`
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)fill, (int)(uint32_t)v, CTRL, ROWS, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(fill >> 32), (int)(uint32_t)(v >> 32), CTRL, ROWS, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t gw_readlane(uint64_t v, int src)   // src wave-uniform
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t gw_bpermute(int byte_addr, uint64_t v)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(byte_addr, (int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// One window, all 64 levels if need be: eqcol = full "equal" mask of text column `lane` (bit q set <=> p[m-1-q] == t[lane]). Lane j works on
// column n-1-j. Returns the first level whose column 0 reports an alignment (bit m-1), -1 if none of the 64 does; levels 0 .. that one are in
// slab[level * 64 + lane].
template <bool REG>   // REG: m = n = 64 as constants
__device__ __forceinline__ int ga_dc64_scan(int lane, uint64_t eqcol, int n_, int m_, uint64_t *slab, uint64_t &eq_own, uint64_t &c_own)
{
    const int n = REG ? kGaW : n_, m = REG ? kGaW : m_;
    uint64_t e = gw_bpermute(max(n - 1 - lane, 0) << 2, eqcol);
    if (lane >= n) e = 0;                                // beyond column 0: these lanes only receive
    const int i16 = lane & 15, row = lane >> 4;
    const int k4 = i16 + 1, k5 = (lane & 31) + 1;        // a lane's own segment at the two broadcast steps
    // the A parts: ak[s] = this lane's own segment's A BEFORE step s; afull = the whole prefix's
    uint64_t ak[6], A = e;
    ak[0] = A; { const uint64_t t = (gw_dpp<0x111, 0xf>(0, A) << 1) & A; A = i16 >= 1 ? t : A; }
    ak[1] = A; { const uint64_t t = (gw_dpp<0x112, 0xf>(0, A) << 2) & A; A = i16 >= 2 ? t : A; }
    ak[2] = A; { const uint64_t t = (gw_dpp<0x114, 0xf>(0, A) << 4) & A; A = i16 >= 4 ? t : A; }
    ak[3] = A; { const uint64_t t = (gw_dpp<0x118, 0xf>(0, A) << 8) & A; A = i16 >= 8 ? t : A; }
    ak[4] = A; { const uint64_t t = (gw_dpp<0x142, 0xa>(0, A) << k4) & A; A = (row & 1) ? t : A; }
    ak[5] = A; { const uint64_t t = (gw_dpp<0x143, 0xc>(0, A) << k5) & A; A = (row & 2) ? t : A; }
    auto scan = [&](uint64_t B) -> uint64_t {            // lanes without a source receive 0: nothing is added
        B |= (gw_dppz<0x111>(B) << 1) & ak[0];
        B |= (gw_dppz<0x112>(B) << 2) & ak[1];
        B |= (gw_dppz<0x114>(B) << 4) & ak[2];
        B |= (gw_dppz<0x118>(B) << 8) & ak[3];
        B |= (gw_dpp<0x142, 0xa>(0, B) << k4) & ak[4];
        B |= (gw_dpp<0x143, 0xc>(0, B) << k5) & ak[5];
        return B;
    };
    uint64_t c = scan(e & 1ull);                         // level 0: C_n[0] = 0, G = 0
    slab[lane] = c;
    int d = 0;
    for (;;) {
        if (__builtin_amdgcn_readlane((int)(uint32_t)(c >> (m - 1)), n - 1) & 1) break;   // column 0, pattern position 0
        if (++d == 64) { d = -1; break; }
        const uint64_t cp = gw_dpp<0x138, 0xf>((1ull << (d - 1)) - 1ull, c);              // wave_shr:1: C_{a+1}[d-1]; lane 0's neighbour is column n
        const uint64_t B = scan((cp << 1) | cp | (c << 1) | 1ull);
        const uint64_t x = lane == 63 ? 0ull : ((1ull << d) - 1ull) << (lane + 1);       // the initial column C_n[d] through the whole prefix
        c = (x & A) | B;
        slab[d * 64 + lane] = c;
    }
    eq_own = e;
    c_own = c;
    return d;
}

// ---- fast path (<= 15 edits): lanes = text columns, banded complemented 32-bit words
constexpr int kGlDiag = 15;                                           // band bit of the main diagonal
constexpr size_t kGlLdsBytes = (size_t)16 * kGaW * 4 + 64;
__device__ __forceinline__ uint32_t gl_lowmask(int k) { return k <= 0 ? 0u : k >= 32 ? ~0u : (1u << k) - 1u; }   // the k lowest bits
// Band word of column `col` from the full 64-bit "equal" mask eq (bit q set <=> p[m-1-q] == t[col], bits >= m clear), complemented-PM form:
// bit b = eq bit (b + m-16-col); positions past the pattern's end (q < 0) read as set, positions before its start as clear.
__device__ __forceinline__ uint32_t gl_band_eq(uint64_t eq, int col, int m)
{
    const int s = m - 16 - col;                          // q of band bit 0
    const int up = -s;                                   // (1 .. 78 where it is used)
    const uint32_t dn = (uint32_t)(eq >> (s & 63));
    const uint32_t uw = up >= 32 ? ~0u : (uint32_t)((eq << (up & 31)) | ((1ull << (up & 31)) - 1ull));
    return s >= 0 ? dn : uw;
}
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t gl_dpp(uint32_t fill, uint32_t v)   // lanes without a source (or outside the row mask) receive `fill`
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, CTRL, ROWS, 0xf, false);
}
// One window: eqcol = band "equal" word of text column `lane` (lanes >= n: anything). Lane j works on column n-1-j -- the recurrence runs
// towards higher lanes, the direction DPP scans go; lanes >= n lie beyond column 0 and are never read. nm = 16 + n - m: the initial column
// c_n[d] = ~(~0 << d) is, in column n's band coordinates, the nm + d lowest bits. Returns the first level (0..15) whose column 0 reports an
// alignment, -1 if none does; levels 0 .. that one are in Rb[level * 64 + lane].
template <bool REG>   // REG: a regular window (m = n = 64): n and nm are compile-time constants
__device__ __forceinline__ int ga_dc16_scan(int lane, uint32_t eqcol, int n_, int nm_, uint32_t *Rb, uint32_t &eq_own, uint32_t &c_own)
{
    constexpr uint32_t ONES = ~0u;
    const int n = REG ? kGaW : n_, nm = REG ? 16 : nm_;
    uint32_t e = (uint32_t)__builtin_amdgcn_ds_bpermute(max(n - 1 - lane, 0) << 2, (int)eqcol);
    // inclusive AND-scan of eq; ek[k] = the product over this lane's segment BEFORE step k (what step k of the g-scan multiplies with)
    uint32_t ek[6];
    ek[0] = e; e &= gl_dpp<0x111, 0xf>(ONES, e);         // row_shr:1
    ek[1] = e; e &= gl_dpp<0x112, 0xf>(ONES, e);         // row_shr:2
    ek[2] = e; e &= gl_dpp<0x114, 0xf>(ONES, e);         // row_shr:4
    ek[3] = e; e &= gl_dpp<0x118, 0xf>(ONES, e);         // row_shr:8
    ek[4] = e; e &= gl_dpp<0x142, 0xa>(ONES, e);         // row_bcast:15 into rows 1 and 3
    ek[5] = e; e &= gl_dpp<0x143, 0xc>(ONES, e);         // row_bcast:31 into rows 2 and 3
    uint32_t c = e & gl_lowmask(nm);                     // level 0: c_a = c_n[0] & eq_{n-1} & .. & eq_a
    ga_lds_order();                                      // (the previous window's walk has read its rows: no store of this window moves above those loads)
    Rb[lane] = c;
    int d = 0;
    for (;;) {
        if ((__builtin_amdgcn_readlane((int)c, n - 1) >> kGlDiag) & 1) break;          // column 0, pattern position 0
        if (++d == 16) { d = -1; break; }
        const uint32_t cp = gl_dpp<0x138, 0xf>(gl_lowmask(nm + d - 1), c);              // wave_shr:1: c_{a+1}[d-1]; lane 0's neighbour is column n
        uint32_t g = ((c << 1) | cp) | (cp >> 1);        // pattern-only edit, substitution, text-only edit
        g |= gl_dppz<0x111>(g) & ek[0];
        g |= gl_dppz<0x112>(g) & ek[1];
        g |= gl_dppz<0x114>(g) & ek[2];
        g |= gl_dppz<0x118>(g) & ek[3];
        g |= gl_dpp<0x142, 0xa>(0u, g) & ek[4];
        g |= gl_dpp<0x143, 0xc>(0u, g) & ek[5];
        c = (e & gl_lowmask(nm + d)) | g;                // the initial column c_n[d] through the whole product, then everything added on the way
        Rb[d * kGaW + lane] = c;
        // (written out as v_and_b32_dpp + v_or_b32 per step -- 18 instead of 29 instructions per level -- this loop is SLOWER: 8.4 against 7.95 ms
        // per 4 096 pairs on the same box, profiles/NOTES.md R4.5. The compiler's v_mov / v_mov_dpp / v_and_or triplets stay.)
    }
    ga_lds_order();
    eq_own = ek[0];                                      // for the traceback: this lane's column's eq word and its word of the hit level
    c_own = c;
    return d;
}
#ifdef AIM_GA_STAMPS   // diagnostic builds only: s_memtime per phase of a window, summed per pair, dumped into the pair's ops row
#define AIM_GASTAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); gsum[i] += t_ - glast; glast = t_; } while (0)
#else
#define AIM_GASTAMP(i) do { } while (0)
#endif
template <bool BT>
__global__ __launch_bounds__(64) void genasm_wave_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    uint32_t *Rb = reinterpret_cast<uint32_t *>(smem);   // fast path: [16 levels][64 lanes] banded words
    uint64_t *Rg = reinterpret_cast<uint64_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave);   // full-width path: [64 levels][64 lanes]
    const int lane = threadIdx.x;
    const int rs = a.p.read_size;
    constexpr uint64_t ONES = ~0ull;

    for (uint32_t it = 0;; ++it) {
        uint32_t pair;
        if (!xcd_unit(a.n_pairs, it, &pair)) break;
        const aim_request_t rq = load_request(a, pair);
        const int plen = __builtin_amdgcn_readfirstlane(rq.pattern_len), tlen = __builtin_amdgcn_readfirstlane(rq.text_len);   // uniform
        const unsigned char *gP = reinterpret_cast<const unsigned char *>(a.patterns + (uint64_t)pair * rs);
        const unsigned char *gT = reinterpret_cast<const unsigned char *>(a.texts + (uint64_t)pair * rs);
        char *ops = BT ? a.ops + (uint64_t)pair * 2 * rs : nullptr;
        const int cap = 2 * rs;
        int pi = 0, ti = 0, nops = 0, dist = 0, status = AIM_PAIR_OK;   // wave-uniform
        int wide_run = 0;                                    // consecutive windows that needed more than 15 edits
#ifdef AIM_GA_STAMPS
        unsigned long long gsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, glast;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(glast) :: "memory");
#endif

        while (pi < plen && ti < tlen) {
            const int m = min(kGaW, plen - pi), n = min(kGaW, tlen - ti);
            const bool last = (m == plen - pi) && (n == tlen - ti);
            // lane j: p[m-1-j] (reversed, for the pattern masks) and t[j]
            const int prev = lane < m ? (int)gP[pi + m - 1 - lane] : 0x100;     // 0x100 never equals a byte
            const int tfwd = lane < n ? (int)gT[ti + lane] : 0x300;
            AIM_GASTAMP(0);   // window characters from HBM
            // lane j: PM of text column j -- bit i = 0 <=> p[m-1-i] == t[j] -- from one ballot per DISTINCT character of the window (any
            // byte values: the reference family compares raw bytes). The four bases first, straight-line (the data-dependent loop costs a scalar /
            // vector round trip per character); whatever else the window holds goes through the loop, which then usually does not run.
            uint64_t mypm = ONES;
            {
                const uint64_t eA = __ballot(prev == 'A'), eC = __ballot(prev == 'C'), eG = __ballot(prev == 'G'), eT = __ballot(prev == 'T');
                mypm = tfwd == 'A' ? ~eA : mypm;
                mypm = tfwd == 'C' ? ~eC : mypm;
                mypm = tfwd == 'G' ? ~eG : mypm;
                mypm = tfwd == 'T' ? ~eT : mypm;
            }
            for (uint64_t rest = __ballot(lane < n && tfwd != 'A' && tfwd != 'C' && tfwd != 'G' && tfwd != 'T'); rest;) {
                const int c = __builtin_amdgcn_readlane(tfwd, (int)__builtin_ctzll(rest));
                const uint64_t pm = ~__ballot(prev == c);
                if (tfwd == c) mypm = pm;
                rest &= ~__ballot(tfwd == c);
            }
            AIM_GASTAMP(1);   // pattern masks
            // FAST PATH: levels 0..15. Level d of a column depends on levels <= d only, so these are exactly the first 16 of the full
            // computation; if the window aligns within 15 edits the traceback never looks further.
            const bool regular = m == kGaW && n == kGaW && !last;   // 2 499 of the 2 500 windows of a 100-kb pair: the fast path is compiled twice, once with these as constants
            uint32_t eq_own = 0, c_own = 0;
            int d = -1;
            if (wide_run < 3) d = regular ? ga_dc16_scan<true>(lane, gl_band_eq(~mypm, lane, kGaW), kGaW, 16, Rb, eq_own, c_own)
                                        : ga_dc16_scan<false>(lane, gl_band_eq(~mypm, lane, m), n, 16 + n - m, Rb, eq_own, c_own);
            const bool slow = d < 0;           // wave-uniform
            AIM_GASTAMP(2);   // DC, 16 levels
            uint64_t weq = 0, wc = 0;
            if (slow) {
                // FULL-WIDTH PATH (a window that needs 16..63 edits): up to 64 levels of 64-bit words, rows to this wavefront's HBM slab
                d = regular ? ga_dc64_scan<true>(lane, ~mypm, kGaW, kGaW, Rg, weq, wc) : ga_dc64_scan<false>(lane, ~mypm, n, m, Rg, weq, wc);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront reads its own slab back below
            }
            // a pair that has lost the diagonal stays lost: after three wide windows in a row the next ones go straight to the full-width path, until one
            // of them aligns within 15 edits again (reads with ~20 % errors alternate between the two paths: they keep trying the fast one first)
            wide_run = (slow && (d < 0 || d > 15)) ? wide_run + 1 : 0;
            AIM_GASTAMP(3);   // DC, 64 levels (rare)
            int ca = 0, cb = 0, wn = 0;           // consumed text / pattern characters, ops of this window (uniform)
            // GenASM-TB. The window's ops live in two registers (lane i: ops i and 64 + i), pre-set to 'M': a run of matches costs
            // nothing to record and an edit is one compare + select.
            int opsA = 'M', opsB = 'M';
            auto put = [&](int ch) {
                opsA = lane == wn ? ch : opsA;
                opsB = lane + 64 == wn ? ch : opsB;
                ++wn;
            };
            if (d < 0) {   // [spec] no alignment of this window within 63 edits: diagonal steps
                const int steps = min(min(m, n), kGaCommit);
                const int pfwd = lane < m ? (int)gP[pi + lane] : 0x200;
                const bool x = lane < steps && pfwd != tfwd;
                opsA = x ? 'X' : opsA;
                dist += __builtin_popcountll(__ballot(x));
                ca = cb = wn = steps;
            } else if (slow) {
                // Full-width walk: walk_cols (below) in full coordinates. The cells of a run of matches, (ca + i, cb + i), sit on different bits
                // of their columns' words: lane of column a tests bit q = m-2-b of C_{a+1}[d] (ok(a+1, b+1, d); past the pattern's end: true) and
                // bit q+1 of EQ_a (p[b] == t[a]), b = cb + a - ca. One coalesced 512-B load of the level below per edit.
                auto walk_wide = [&](auto reg_tag) {
                    constexpr bool REG = decltype(reg_tag)::value;   // a regular window (m = n = 64, not the last): what a pair that lost the diagonal is made of
                    const int N = REG ? kGaW : n, M = REG ? kGaW : m;
                    const bool LAST = REG ? false : last;
                    uint64_t wd = wc;                                         // C_a[d] of my column
                    const int a_l = N - 1 - lane;                             // my column
                    // every edit moves exactly one level down: the rows of the next four levels are asked for ahead of their use (a slab round trip is ~600 cycles,
                    // an iteration ~150)
                    auto row = [&](int lvl) -> uint64_t { return Rg[max(lvl, 0) * 64 + lane]; };
                    uint64_t w1 = row(d - 1), w2 = row(d - 2), w3 = row(d - 3), w4 = row(d - 4);
                    for (;;) {
                        if (!REG && ca == N) {   // the window's text is used up: pattern-only edits to the end of the pattern (to the commit bound if this is not the last window)
                            const int k = LAST ? M - cb : (ca >= kGaCommit ? 0 : max(min(M, kGaCommit) - cb, 0));
                            opsA = (lane >= wn && lane < wn + k) ? 'D' : opsA;
                            opsB = (lane + 64 >= wn && lane + 64 < wn + k) ? 'D' : opsB;
                            wn += k; cb += k; d -= k; dist += k;
                            break;
                        }
                        const uint64_t wn1 = gw_dpp<0x138, 0xf>((1ull << d) - 1ull, wd);      // wave_shr:1: C_{a+1}[d]; lane 0's neighbour is column n
                        const int q = M - 2 - cb - (a_l - ca);
                        const bool okb = (!REG && q < 0) || ((wn1 >> (q & 63)) & 1ull);   // (a regular window's cells have b < 40: q >= 22)
                        const bool eqb = (weq >> ((q + 1) & 63)) & 1ull;
                        const uint64_t mm = __ballot(okb && eqb);
                        const uint64_t sh = ~(mm << (kGaW - N + ca));         // bit 63 = column ca, then ca + 1, ...: clear = the run goes on
                        int run = sh ? (int)__builtin_clzll(sh) : 64;
                        int lim = kGaCommit - max(ca, cb);                    // cells with a < 40 and b < 40 ...
                        if (!REG) { lim = min(N - ca, M - cb); if (!LAST) lim = min(lim, kGaCommit - max(ca, cb)); }   // ... or inside the window, whichever ends first
                        run = min(run, lim);
                        wn += run; ca += run; cb += run;
                        if (!REG && cb == M) break;
                        if (!LAST && (ca >= kGaCommit || cb >= kGaCommit)) break;
                        if (!REG && ca == N) continue;
                        if (d == 0) { status = AIM_PAIR_WFA_NO_LINK; break; }    // cannot happen (the recurrence guarantees one rule applies)
                        const uint64_t wl = w1;                               // C_a[d-1] of my column
                        w1 = w2; w2 = w3; w3 = w4; w4 = row(d - 5);
                        const uint64_t s1 = (!REG && ca + 1 == N) ? (1ull << (d - 1)) - 1ull : gw_readlane(wl, REG ? N - 2 - ca : max(N - 2 - ca, 0));   // column ca + 1
                        const uint64_t s0 = gw_readlane(wl, N - 1 - ca);                                                  // column ca
                        const int q1 = M - 2 - cb, q0 = M - 1 - cb;           // bits of pattern positions cb + 1 and cb
                        auto okq = [&](uint64_t w, int q_) -> bool { return (!REG && q_ < 0) || ((w >> (q_ & 63)) & 1ull); };
                        const int op = okq(s1, q1) ? 'X' : okq(s0, q1) ? 'D' : okq(s1, q0) ? 'I' : 0;
                        if (op == 0) { status = AIM_PAIR_WFA_NO_LINK; break; }   // cannot happen
                        put(op);
                        ca += op != 'D';
                        cb += op != 'I';
                        --d; ++dist;
                        wd = wl;
                    }
                };
                if (regular) walk_wide(std::true_type{}); else walk_wide(std::false_type{});
            } else {
                // walk_cols: the same walk over the banded words, the lanes still bound to their COLUMNS (lane j: column n-1-j, as in the sweep).
                // Every cell of a run of matches lies on the walk's diagonal k = ca - cb, whose band bit is kb = 15 + k in every column, and
                // "p[b] == t[a] and ok(a+1, b+1, d)" is bit kb of (c_{a+1}[d] & eq_a) -- the recurrence's own match term: one DPP shift of the
                // level's words (column n, the initial one, enters at lane 0), one AND, one ballot, and the run is the string of set bits from
                // column ca on (a count of leading ones), cut at the window's and the commit bounds. The edit after the run is decided by three
                // bits of level d-1 at the one cell where the run stopped: two v_readlane of that level's words, the rest scalar, in the
                // oracle's order. Words are complemented: bit set = ok(). One LDS read per edit (the next level's words), nothing per match.
                auto walk_reg = [&]() {   // a regular window: m = n = 64, not the last -- the walk ends on the commit bound, column 64 is never asked for, at most 55 ops
                    uint32_t wd = c_own;                                      // c_a[d] of my column
                    int kb = kGlDiag;                                         // wave-uniform, 0 .. 30
                    for (;;) {
                        const uint32_t wn1 = gl_dpp<0x138, 0xf>(0u, wd);      // wave_shr:1: c_{a+1}[d] (column 64 is never asked for: a < 40)
                        const uint64_t mm = __ballot(((wn1 & eq_own) >> kb) & 1u);
                        const uint64_t sh = ~(mm << ca);                      // bit 63 = column ca, then ca + 1, ...: clear = the run goes on
                        int run = sh ? (int)__builtin_clzll(sh) : 64;
                        run = min(run, kGaCommit - max(ca, cb));              // cells with a < 40 and b < 40
                        wn += run; ca += run; cb += run;
                        if (ca >= kGaCommit || cb >= kGaCommit) break;
                        if (d == 0) { status = AIM_PAIR_WFA_NO_LINK; break; }    // cannot happen (the recurrence guarantees one rule applies)
                        const uint32_t wl = Rb[(d - 1) * kGaW + lane];        // c_a[d-1] of my column
                        const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)wl, kGaW - 2 - ca);   // column ca + 1
                        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)wl, kGaW - 1 - ca);   // column ca
                        const int op = ((s1 >> kb) & 1u) ? 'X' : ((s0 >> ((kb - 1) & 31)) & 1u) ? 'D' : ((s1 >> ((kb + 1) & 31)) & 1u) ? 'I' : 0;
                        if (op == 0) { status = AIM_PAIR_WFA_NO_LINK; break; }   // cannot happen
                        put(op);
                        ca += op != 'D';
                        cb += op != 'I';
                        kb = kGlDiag + ca - cb;
                        --d; ++dist;
                        wd = wl;
                    }
                };
                auto walk_any = [&]() {   // any window
                    const int N = n, M = m, NM = 16 + n - m;
                    const bool LAST = last;
                    uint32_t wd = c_own;                                      // c_a[d] of my column
                    int kb = kGlDiag;                                         // wave-uniform, 0 .. 30
                    for (;;) {
                        if (ca == N) {   // the window's text is used up: pattern-only edits to the end of the pattern (to the commit bound if this is not the last window)
                            const int k = LAST ? M - cb : (ca >= kGaCommit ? 0 : max(min(M, kGaCommit) - cb, 0));
                            opsA = (lane >= wn && lane < wn + k) ? 'D' : opsA;
                            opsB = (lane + 64 >= wn && lane + 64 < wn + k) ? 'D' : opsB;
                            wn += k; cb += k; d -= k; dist += k;
                            break;
                        }
                        const uint32_t wn1 = gl_dpp<0x138, 0xf>(gl_lowmask(NM + d), wd);   // wave_shr:1: c_{a+1}[d]; lane 0's neighbour is column n
                        const uint64_t mm = __ballot(((wn1 & eq_own) >> kb) & 1u);
                        const uint64_t sh = ~(mm << (kGaW - N + ca));         // bit 63 = column ca, then ca + 1, ...: clear = the run goes on
                        int run = sh ? (int)__builtin_clzll(sh) : 64;
                        int lim = min(N - ca, M - cb);                        // cells inside the window ...
                        if (!LAST) lim = min(lim, kGaCommit - max(ca, cb));   // ... and with a < 40 and b < 40
                        run = min(run, lim);
                        wn += run; ca += run; cb += run;
                        if (cb == M) break;
                        if (!LAST && (ca >= kGaCommit || cb >= kGaCommit)) break;
                        if (ca == N) continue;
                        if (d == 0) { status = AIM_PAIR_WFA_NO_LINK; break; }    // cannot happen (the recurrence guarantees one rule applies)
                        const uint32_t wl = Rb[(d - 1) * kGaW + lane];        // c_a[d-1] of my column
                        const uint32_t s1 = ca + 1 == N ? gl_lowmask(NM + d - 1) : (uint32_t)__builtin_amdgcn_readlane((int)wl, max(N - 2 - ca, 0));   // column ca + 1
                        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)wl, N - 1 - ca);                                              // column ca
                        const int op = ((s1 >> kb) & 1u) ? 'X' : ((s0 >> ((kb - 1) & 31)) & 1u) ? 'D' : ((s1 >> ((kb + 1) & 31)) & 1u) ? 'I' : 0;
                        if (op == 0) { status = AIM_PAIR_WFA_NO_LINK; break; }   // cannot happen
                        put(op);
                        ca += op != 'D';
                        cb += op != 'I';
                        kb = kGlDiag + ca - cb;
                        --d; ++dist;
                        wd = wl;
                    }
                };
                if (regular) walk_reg(); else walk_any();
            }
            AIM_GASTAMP(4);   // traceback
            if (BT) {   // the window's ops leave as coalesced byte stores
                if (lane < wn && nops + lane < cap) ops[nops + lane] = (char)opsA;
                if (lane + 64 < wn && nops + 64 + lane < cap) ops[nops + 64 + lane] = (char)opsB;
            }
            nops += wn;
            pi += cb;
            ti += ca;
            AIM_GASTAMP(5);   // ops stores
            if (status != AIM_PAIR_OK) break;
        }
#ifdef AIM_GA_STAMPS
        if (BT && lane == 0) { unsigned long long *dbg = reinterpret_cast<unsigned long long *>(ops); for (int i = 0; i < 8; ++i) dbg[i] = gsum[i]; }
#endif
        if (status == AIM_PAIR_OK) {   // one sequence is exhausted: the rest of the other is gaps
            const int rp = plen - pi, rt = tlen - ti;
            if (BT) {
                for (int i = lane; i < rp; i += kWave) if (nops + i < cap) ops[nops + i] = 'D';
                for (int i = lane; i < rt; i += kWave) if (nops + rp + i < cap) ops[nops + rp + i] = 'I';
            }
            nops += rp + rt;
            dist += rp + rt;
        }
        if (lane == 0) {
            aim_result_t r;
            r.max_operations = plen + tlen;
            r.begin_offset = 0;
            r.end_offset = nops;
            r.score = dist;
            r.status = status;
            r.idx = rq.idx;
            store_result(a, pair, r);
        }
        __syncthreads();   // single wavefront: orders this pair's LDS traffic before the next pair's
    }
}

constexpr uint64_t kGaSlabBytes = (uint64_t)(kGaW + 1) * 64 * 8;   // one wavefront's full-width columns in HBM scratch

// Residency: 4 KB of LDS and <= 52 VGPRs allow 32 wavefronts per CU; 24 measure best (l = 10 000, 8 192 pairs: 16 / 24 / 32 per CU = 1.95 /
// 1.60 / 1.74 ms, profiles/NOTES.md R4.5).
inline void genasm_plan(const aim_params_t &p, const Knobs &kn, uint32_t n_pairs, uint32_t *grid, uint32_t *block, size_t *lds)
{
    (void)p;
    *block = kWave;
    *lds = kGlLdsBytes;
    uint32_t per_cu = (uint32_t)std::min<size_t>(24, lds_workgroups_per_cu(*lds));
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
    (void)kn;
    if (p.flags & AIM_FLAG_BACKTRACE) hipLaunchKernelGGL((genasm_wave_kernel<true>), dim3(grid), dim3(kWave), lds, s, ka);
    else hipLaunchKernelGGL((genasm_wave_kernel<false>), dim3(grid), dim3(kWave), lds, s, ka);
}
#else
void genasm_launch(const aim_params_t &p, const Knobs &kn, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s);
#endif

}  // namespace aim
