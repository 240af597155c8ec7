// genasm_wave.hpp -- GenASM (bit-vector approximate string matching, windowed traceback) for long reads: ONE PAIR PER
// WAVEFRONT, the 64 lanes are the 64 error levels of a window.  BASELINE config 5.
//
// PARITY UNPINNED: AIM's GenASM is an un-vendored, un-pinned submodule (/root/reference/.gitmodules:1-3, empty directory,
// no call sites or tests).  This kernel implements the PUBLISHED algorithm (Senol Cali et al., MICRO 2020: GenASM-DC,
// Algorithm 1; GenASM-TB and the W = 64 / O = 24 windows, Section 6) exactly as oracle/genasm_oracle.c restates it --
// every open choice is fixed there and marked [spec] -- and is tested bit for bit against that restatement.
//
// Mapping.  A pair of 100 kb reads is ~2 500 dependent windows, so what matters is the LATENCY of one window, and the
// parallelism has to come from inside it. Lane d owns error level d. GenASM-DC (Algorithm 1), columns a = n-1 .. 0:
//     R_a[d] = ((R_{a+1}[d] << 1) | PM[t_a])  &  (R_{a+1}[d-1] << 1)  &  R_{a+1}[d-1]  &  (R_a[d-1] << 1)
//              match                             substitution            text-only edit   pattern-only edit
// Level d of column a needs level d-1 of the SAME column, so a column-by-column sweep is a 16- (or 64-) deep chain per
// column (round 2's first version ran it as a prefix-AND-with-shift scan: 4-6 dependent DPP steps per column, ~440 cycles
// per column). The sweep is instead SKEWED: at step u lane d works on column n-1-(u-d). Everything it needs from lane d-1 is
// then that lane's value after step u-1 (R_a[d-1]) and after step u-2 (R_{a+1}[d-1]): ONE DPP shift by one lane per step and
// a remembered copy, n + 15 (or n + 63) steps of ~6 dependent instructions per window. The pattern masks travel the same
// way: lane j first holds PM of text column j (one ballot per DISTINCT character of the window, any byte values -- the
// reference family compares raw bytes), lane 0 picks column n-1-u with a v_readlane, and the masks shift one lane per step.
// Almost every window needs far fewer than 16 edits, so the sweep first runs levels 0..15 only (one 16-lane DPP row, columns
// kept in LDS: 13 KB per wavefront) and falls back to all 64 levels (wave_shr DPP, columns in an HBM slab) when that finds
// no alignment. Every column is kept ([a][d]) because the traceback -- a wave-uniform walk of <= ~80 steps per window --
// reads R_a[d], R_{a+1}[d] and R_{a+1}[d-1] along its path.
// Integer / bit work only; HBM sees each sequence byte once and the ops once.
#pragma once

#include <type_traits>
#include "aim_device.hpp"

namespace aim {

constexpr int kGaW = 64;        // window
constexpr int kGaCommit = 40;   // W - O
// Fast-path column store in LDS: columns -16 .. W+16 (the skewed sweep lets a level run up to 15 columns past either end, its
// mask prefetch one more; what it computes there is never read), 17 slots of 8 B per column: R_a[0..15] and the column's pattern mask.
constexpr int kGaPad = 16, kGaSlots = 17, kGaCols = kGaPad + kGaW + 1 + kGaPad;

// lane L receives the value of lane L - 1 (WIDE: of the wavefront, DPP wave_shr:1; otherwise of its own 16-lane row, DPP
// row_shr:1); lane 0 (of the wavefront / of each row) receives `fill`
template <bool WIDE>
__device__ __forceinline__ uint64_t ga_shr1(uint64_t v, uint64_t fill)
{
    constexpr int ctrl = WIDE ? 0x138 : 0x111;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)fill, (int)(uint32_t)v, ctrl, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(fill >> 32), (int)(uint32_t)(v >> 32), ctrl, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
// Lanes of the one wavefront hand values to each other through LDS (masks written by lane j are read by lane d, columns written by lanes
// 0..15 are read by all 64). The hardware executes a wavefront's LDS instructions in order, but the COMPILER reasons per thread and may
// move a load above a store to a different address of the same thread: this stops it (no instruction is emitted).
__device__ __forceinline__ void ga_lds_order() { asm volatile("" ::: "memory"); }
__device__ __forceinline__ uint64_t ga_readlane(uint64_t v, int src)   // src wave-uniform
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, src);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
// GenASM-DC of one window, skewed (see the header): levels 0 .. LV-1 in lanes 0 .. LV-1, columns n-1 .. 0; column a of
// level d is stored at cols[a * LV + d] (column n = the initial ~0 << d included). Returns R_0[lane].
template <bool WIDE>
__device__ __forceinline__ uint64_t ga_dc(int n_, int lane, uint64_t mypm, uint64_t *cols)
{
    const int n = __builtin_amdgcn_readfirstlane(n_);   // wave-uniform by construction; said so for the v_readlane index below
    constexpr int LV = WIDE ? 64 : 16;
    constexpr uint64_t ONES = ~0ull;
    const bool mine = lane < LV;
    uint64_t cur = ONES << lane;                         // R_n[d]
    if (mine) cols[n * LV + lane] = cur;
    uint64_t nb_prev = ga_shr1<WIDE>(cur, ONES);         // lane d-1 two steps ago
    uint64_t pmv = ONES;
    for (int u = 0; u < n + LV - 1; ++u) {
        const int c0 = n - 1 - u;                        // lane 0's column at this step
        pmv = ga_shr1<WIDE>(pmv, ga_readlane(mypm, c0 < 0 ? 0 : c0));
        const uint64_t nb_cur = ga_shr1<WIDE>(cur, ONES);                 // lane d-1 one step ago: R_a[d-1]
        const int col = c0 + lane;                       // my column
        uint64_t y = (cur << 1) | pmv;                   // match
        if (lane > 0) y &= (nb_prev << 1) & nb_prev & (nb_cur << 1);      // substitution, text-only edit, pattern-only edit
        nb_prev = nb_cur;
        if (mine && col >= 0 && col < n) {
            cur = y;
            cols[col * LV + lane] = y;
        }
    }
    return cur;
}
__device__ __forceinline__ uint64_t ga_uniform(uint64_t v)   // value known to be wave-uniform -> SGPRs
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// The same sweep for levels 0..15 (lanes 0..15), written for the fewest instructions per step: a single wavefront issues
// one instruction every ~4-5 cycles whatever its type, and a window is n + 15 dependent steps, so the step's instruction
// count IS the window's latency (stamps: 43 instructions, 232 cycles per step before; BASELINE config 5 is 2 500 windows
// per pair). No lane tests "is my column inside the window": a level that has not started sees pattern masks of ~0 (stored
// for columns n .. n+15), which keep it at its initial ~0 << d, and a level that is finished computes on into 15 padding
// columns below column 0 that nobody reads. The pattern mask of a column is read from the column's 17th LDS slot (no
// readlane / DPP feed), lane 0's "no level below me" is an OR with a constant, and the sweep stops at the first level whose
// column 0 reports an alignment: the traceback starts there and only ever moves to lower levels.
// Returns the ballot of levels (so far) whose R_0 has bit m-1 clear; 0 = no alignment within 15 edits.
__device__ __forceinline__ int ga_slot(int col) { return (kGaPad + kGaW - col) * kGaSlots; }   // columns are stored in DESCENDING order: the sweep's addresses ascend (immediate offsets)
__device__ __forceinline__ uint64_t ga_dc16(int n_, int m_, int lane, uint64_t mypm, uint64_t *Rs)
{
    constexpr uint64_t ONES = ~0ull;
    const int n = __builtin_amdgcn_readfirstlane(n_), m = __builtin_amdgcn_readfirstlane(m_);
    if (lane < n) Rs[ga_slot(lane) + 16] = mypm;
    uint64_t hit = 0;
    if (lane < 16) {
        Rs[ga_slot(n + lane) + 16] = ONES;
        uint64_t cur = ONES << lane;                     // R_n[d]
        Rs[ga_slot(n) + lane] = cur;
        ga_lds_order();
        const uint64_t lane0 = lane == 0 ? ONES : 0ull;
        const uint64_t endbit = 1ull << (m - 1);
        uint64_t *rp = Rs + ga_slot(n - 1 + lane) + lane;                // my R slot of my column at step 0
        const uint64_t *pp = Rs + ga_slot(n - 1 + lane) + 16;            // my column's pattern mask
        auto shr1 = [](uint64_t v) -> uint64_t {         // lane d-1's value; lane 0 receives 0 (bound_ctrl), OR-ed away below
            const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, 0x111, 0xf, 0xf, true);
            const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), 0x111, 0xf, 0xf, true);
            return ((uint64_t)hi << 32) | lo;
        };
        // one step; the roles of the two neighbour registers and of the two mask registers alternate (no copies)
        auto step = [&](const uint64_t &nb_prev, uint64_t &nb_cur, const uint64_t &pm_now, uint64_t &pm_next, int k) {
            pm_next = pp[(k + 1) * kGaSlots];            // next step's mask: off the dependent chain
            nb_cur = shr1(cur);                          // lane d-1 one step ago: R_a[d-1]
            // substitution (R_{a+1}[d-1] << 1), text-only edit (R_{a+1}[d-1]), pattern-only edit (R_a[d-1] << 1); level 0 has none
            const uint64_t t = (((nb_prev & nb_cur) << 1) & nb_prev) | lane0;
            cur = ((cur << 1) | pm_now) & t;             // match
            rp[k * kGaSlots] = cur;
        };
        uint64_t nbA = shr1(cur), nbB, pmA = pp[0], pmB;  // nbA: lane d-1 two steps ago
        int u = 0;
        for (; u + 2 <= n - 1; u += 2) {                 // no level has reached column 0 yet
            step(nbA, nbB, pmA, pmB, 0);
            step(nbB, nbA, pmB, pmA, 1);
            rp += 2 * kGaSlots;
            pp += 2 * kGaSlots;
        }
        if (u < n - 1) {
            step(nbA, nbB, pmA, pmB, 0);
            nbA = nbB; pmA = pmB;
            rp += kGaSlots; pp += kGaSlots;
        }
        for (u = n - 1; u < n + 15; ++u) {               // level u - (n-1) completes column 0 in this step
            step(nbA, nbB, pmA, pmB, 0);
            nbA = nbB; pmA = pmB;
            rp += kGaSlots; pp += kGaSlots;
            hit = __ballot(lane == u - (n - 1) && !(cur & endbit));
            if (hit) break;
        }
    }
    ga_lds_order();
    return __builtin_amdgcn_readfirstlane((uint32_t)hit) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(hit >> 32)) << 32);
}
// LONG variant (long reads: thousands of windows per pair, almost all of them REGULAR: m = n = 64 and not the pair's last). Regular windows
// take the path below; every other window -- irregular, the pair's last, or more than 15 edits -- takes the full-width 64-level path.
//
// (1) BANDED bit-vectors (round 4). The fast path stops at 15 edits, and an alignment of the window with <= 15 edits never leaves the
// diagonals |a - i| <= 15 (a = text column, i = pattern position; the window's walk starts at (0, 0) and every edit moves it one diagonal
// and costs one level). Follow one bit of the recurrence: bit q (pattern position i = 63 - q) of R_a[d] reads bit q-1 of column a+1 (match,
// substitution: the SAME diagonal), bit q-1 of R_a[d-1] and bit q of R_{a+1}[d-1] (one diagonal away, one level down). So a bit on diagonal k
// of level l depends on diagonals k-j .. k+j of level l-j only, and what the hit test (diagonal 0, level d0 <= 15) and the traceback (a cell
// on diagonal k with |k| <= d0 - d at level d, its neighbours on k +- 1 at level d-1) read is a function of diagonals -15 .. +15 -- 31 bits.
// The sweep therefore keeps, per column a, ONE 32-bit word whose bit b is bit q = b + 48 - a of the full vector (diagonal b - 15): in these
// coordinates the "<< 1" of the terms that come from column a+1 disappears (the band slides with the column), the pattern-only edit keeps
// its "<< 1" and the text-only edit becomes ">> 1". Bits shifted in at either end are garbage that moves one bit inwards per level -- at
// level l bits b < l and b > 31 - l -- and the bits read at level l are l .. 30 - l (shown above). Positions past the pattern's end (q < 0)
// are in the band for columns > 48; they hold 0 ("matches") as in the full vector, where "<< 1" shifts them in. The words are kept
// COMPLEMENTED (c = ~R, bit set = "aligns"): AND-of-ORs becomes OR-of-ANDs, which the ISA has three-operand forms for (v_and_or_b32,
// v_lshl_or_b32, v_or3_b32):
//     c_a[d] = (c_{a+1}[d] & ~PM_a)  |  c_{a+1}[d-1]  |  (c_a[d-1] << 1)  |  (c_{a+1}[d-1] >> 1)
// Results are bit-identical to the full-width sweep wherever anything reads them (tests/test_genasm.py, tools/fuzz_parity.py --focus genasm).
// First version: the round-3 skewed 16-lane sweep on these words, 5 instead of 14 VALU per step: 25.4 -> 13.9 ms per 4 096 pairs of 100 kb.
//
// (2) The sweep as a SCAN over columns (round 4: 13.9 -> 9.1 ms). In band coordinates the match term has no shift, so one level is a bitwise linear recurrence
// along the columns,   c_a = (c_{a+1} & eq_a) | g_a   with   g_a[d] = c_{a+1}[d-1] | (c_a[d-1] << 1) | (c_{a+1}[d-1] >> 1),
// i.e. a composition of the maps x -> (x & e) | g, which is associative: (e1, g1) then (e2, g2) = (e1 & e2, (g1 & e2) | g2). Lane j owns column
// 63 - j and a level is ONE inclusive scan over the wavefront -- 4 DPP row_shr steps, row_bcast:15, row_bcast:31 -- instead of a walk down the
// columns; the products of the eq words that the steps need do not depend on the level and are computed once per window (6 registers), so a
// level costs: the neighbour column's word of the level below (one DPP wave_shr), g (3), 6 x (DPP move + and-or), the initial column folded in
// (1), one store of the 64 columns, the hit test on lane 63 -- ~25 instructions with all 64 lanes busy, against 7 x (63 + d) for the skewed
// 16-lane sweep, and levels beyond the first hit are never computed. No masks in LDS, no padding columns: 16 levels x 64 columns x 4 B = 4 KB.
constexpr int kGlDiag = 15;                                           // band bit of the main diagonal
constexpr size_t kGlLdsBytes = (size_t)16 * kGaW * 4 + 64;
__device__ __forceinline__ int gl_word(int col, int lvl) { return lvl * kGaW + (kGaW - 1 - col); }   // lane j stores column 63 - j
// Band word of column `col` from the full 64-bit "equal" mask eq (bit q set <=> p[63 - q] == t[col]), complemented-PM form: bit b = eq bit
// (b + 48 - col); positions past the pattern's end (q < 0) read as set.
__device__ __forceinline__ uint32_t gl_band_eq(uint64_t eq, int col)
{
    const int s = 48 - col;                              // off(col)
    const uint64_t dn = eq >> (s & 63), up = (eq << (-s & 63)) | ((1ull << (-s & 63)) - 1ull);
    return (uint32_t)(s >= 0 ? dn : up);
}
template <int CTRL, int ROWS>
__device__ __forceinline__ uint32_t gl_dpp(uint32_t fill, uint32_t v)   // lanes without a source (or outside the row mask) receive `fill`
{
    return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, CTRL, ROWS, 0xf, false);
}
// One regular window (m = n = 64): eqcol = band "equal" word of text column `lane`. Returns the first level (0..15) whose column 0 reports an
// alignment, -1 if none does; levels 0 .. that one are in Rb[gl_word(col, level)] (lane j: column 63 - j).
__device__ __forceinline__ int ga_dc16_scan(int lane, uint32_t eqcol, uint32_t *Rb, uint32_t &eq_own, uint32_t &c_own)
{
    constexpr uint32_t ONES = ~0u;
    // lane j works on column 63 - j: the recurrence runs towards higher lanes, the direction DPP scans go
    uint32_t e = (uint32_t)__builtin_amdgcn_ds_bpermute((kGaW - 1 - lane) << 2, (int)eqcol);
    // inclusive AND-scan of eq; ek[k] = the product over this lane's segment BEFORE step k (what step k of the g-scan multiplies with)
    uint32_t ek[6];
    ek[0] = e; e &= gl_dpp<0x111, 0xf>(ONES, e);         // row_shr:1
    ek[1] = e; e &= gl_dpp<0x112, 0xf>(ONES, e);         // row_shr:2
    ek[2] = e; e &= gl_dpp<0x114, 0xf>(ONES, e);         // row_shr:4
    ek[3] = e; e &= gl_dpp<0x118, 0xf>(ONES, e);         // row_shr:8
    ek[4] = e; e &= gl_dpp<0x142, 0xa>(ONES, e);         // row_bcast:15 into rows 1 and 3
    ek[5] = e; e &= gl_dpp<0x143, 0xc>(ONES, e);         // row_bcast:31 into rows 2 and 3
    // level 0: c_a = c_64 & eq_63 & .. & eq_a, c_64[0] = ~(~0 << 0) in column 64's coordinates = the low 16 bits (positions past the pattern's end)
    uint32_t c = e & 0xffffu;
    Rb[lane] = c;
    int d = 0;
    for (;;) {
        if ((__builtin_amdgcn_readlane((int)c, kGaW - 1) >> kGlDiag) & 1) break;       // column 0, pattern position 0
        if (++d == 16) { d = -1; break; }
        const uint32_t below = ~(ONES << (15 + d));      // c_64[d-1]
        const uint32_t cp = gl_dpp<0x138, 0xf>(below, c);                               // wave_shr:1: c_{a+1}[d-1]; lane 0's neighbour is column 64
        uint32_t g = ((c << 1) | cp) | (cp >> 1);        // pattern-only edit, substitution, text-only edit
        g |= gl_dpp<0x111, 0xf>(0u, g) & ek[0];
        g |= gl_dpp<0x112, 0xf>(0u, g) & ek[1];
        g |= gl_dpp<0x114, 0xf>(0u, g) & ek[2];
        g |= gl_dpp<0x118, 0xf>(0u, g) & ek[3];
        g |= gl_dpp<0x142, 0xa>(0u, g) & ek[4];
        g |= gl_dpp<0x143, 0xc>(0u, g) & ek[5];
        c = (e & ~(ONES << (16 + d))) | g;               // the initial column c_64[d] through the whole product, then everything added on the way
        Rb[d * kGaW + lane] = c;
        // (written out as v_and_b32_dpp + v_or_b32 per step -- 18 instead of 29 instructions per level -- this loop is SLOWER: 8.4 against 7.95 ms
        // per 4 096 pairs on the same box, profiles/r04/genasm_notes.txt. The compiler's v_mov / v_mov_dpp / v_and_or triplets stay.)
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
template <bool BT, bool LONG>
__global__ __launch_bounds__(64) void genasm_wave_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    // Columns of the window for the traceback. The fast path (16 levels) keeps them in LDS, [kGaCols][17] x 8 B = 13 KB, so that
    // 8 wavefronts are resident per CU (all 64 levels in LDS were 33 KB: 4 per CU, one per SIMD, nothing to overlap the
    // dependent column chain with); the rare slow path (a window needing 16..63 edits) writes [kGaW + 1][64] to this
    // wavefront's slab of HBM scratch instead.
    uint64_t *Rs = reinterpret_cast<uint64_t *>(smem);
    uint32_t *Rb = reinterpret_cast<uint32_t *>(smem);   // LONG: banded 32-bit words (ga_dc16_scan)
    uint64_t *Rg = reinterpret_cast<uint64_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave);
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
#ifdef AIM_GA_STAMPS
        unsigned long long gsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, glast;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(glast) :: "memory");
#endif

        while (pi < plen && ti < tlen) {
            const int m = min(kGaW, plen - pi), n = min(kGaW, tlen - ti);
            const bool last = (m == plen - pi) && (n == tlen - ti);
            // lane j: p[m-1-j] (reversed, for the pattern masks), p[j] and t[j] (forward, for the traceback)
            const int prev = lane < m ? (int)gP[pi + m - 1 - lane] : 0x100;     // 0x100 never equals a byte
            int pfwd = 0x200;                                               // (LONG: only the 64-level path reads it -- loaded there)
            if (!LONG) pfwd = lane < m ? (int)gP[pi + lane] : 0x200;
            const int tfwd = lane < n ? (int)gT[ti + lane] : 0x300;
            AIM_GASTAMP(0);   // window characters from HBM
            // lane j: PM of text column j -- bit i = 0 <=> p[m-1-i] == t[j] -- from one ballot per distinct character
            uint64_t mypm = ONES;
            for (uint64_t rest = __ballot(lane < n); rest;) {
                const int c = __builtin_amdgcn_readlane(tfwd, (int)__builtin_ctzll(rest));
                const uint64_t pm = ~__ballot(prev == c);
                if (tfwd == c) mypm = pm;
                rest &= ~__ballot(tfwd == c);
            }
            AIM_GASTAMP(1);   // pattern masks
            // FAST PATH: levels 0..15 only. Level d of a column depends on levels <= d only, so these are exactly the first 16 of
            // the full computation; if the window aligns within 15 edits (at e = 10 % a 64-character window carries ~6) the
            // traceback never looks further.
            uint64_t hit = 0;
            if (!LONG) hit = ga_dc16(n, m, lane, mypm, Rs);
            int dband = -1;
            uint32_t eq_own = 0, c_own = 0;
            if (LONG && !last && m == kGaW && n == kGaW) dband = ga_dc16_scan(lane, gl_band_eq(~mypm, lane), Rb, eq_own, c_own);   // (any other window: 64-level path)
            if (dband >= 0) hit = 1ull << dband;
            const bool slow = !hit;            // wave-uniform
            AIM_GASTAMP(2);   // DC, 16 levels
            if (slow) {
                if (LONG) pfwd = lane < m ? (int)gP[pi + lane] : 0x200;
                // SLOW PATH (a window that needs 16..63 edits): all 64 levels, columns to this wavefront's HBM slab
                const uint64_t R = ga_dc<true>(n, lane, mypm, Rg);
                hit = __ballot(!((R >> (m - 1)) & 1ull));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront reads its own slab back below
            }
            AIM_GASTAMP(3);   // DC, 64 levels (rare)
            // d0 = smallest level whose bit m-1 is clear in column 0
            int d = hit ? (int)__builtin_ctzll(hit) : -1;
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
                const bool x = lane < steps && pfwd != tfwd;
                opsA = x ? 'X' : opsA;
                dist += __builtin_popcountll(__ballot(x));
                ca = cb = wn = steps;
            } else {
                // One LDS round trip per iteration. Lane i looks at the cell the sequential walk would reach after i matches,
                // (ca+i, cb+i): the characters (ds_bpermute of the window registers), R_{a+1}[d] for the match test, and
                // R_{a+1}[d-1], R_a[d-1] for the edit it would take if the run of matches ended on it. The run is the leading
                // lanes whose match test holds ('M' never changes d); the edit is then read from the first lane where it fails --
                // the same tests in the same order as the step-by-step walk (oracle/genasm_oracle.c), without a second fetch.
                auto walk = [&](auto slow_tag) {
                    constexpr bool SLOW = decltype(slow_tag)::value;
                    auto Rf = [&](int col, int lvl) -> uint64_t { return SLOW ? Rg[col * 64 + lvl] : Rs[ga_slot(col) + lvl]; };
                    const int amax = n - 1;
                    for (;;) {
                        const int ai = ca + lane, bi = cb + lane;
                        const bool inr = bi < m && ai < n && (last || (ai < kGaCommit && bi < kGaCommit));
                        const int aic = min(ai, amax), bic = min(bi, kGaW - 1);
                        const int dm1 = d > 0 ? d - 1 : 0;
                        const uint64_t rn_d = Rf(aic + 1, d), rn_dm1 = Rf(aic + 1, dm1), rc_dm1 = Rf(aic, dm1);
                        const int pch = __builtin_amdgcn_ds_bpermute(bic << 2, pfwd), tch = __builtin_amdgcn_ds_bpermute(aic << 2, tfwd);
                        // clear(r, b) := b >= m || bit (m-1-b) of r is 0
                        const int q1 = m - 2 - bi, q0 = m - 1 - bi;          // bit indices for b = bi + 1 and b = bi
                        auto clr = [&](uint64_t r, int q) -> bool { return q < 0 || !((r >> (q & 63)) & 1ull); };
                        const bool cm = inr && pch == tch && clr(rn_d, q1);
                        int code = 0;                                         // the edit this cell would take
                        if (d > 0) code = clr(rn_dm1, q1) ? 'X' : clr(rc_dm1, q1) ? 'D' : clr(rn_dm1, q0) ? 'I' : 0;
                        const uint64_t bad = ~__ballot(cm);
                        const int run = bad ? (int)__builtin_ctzll(bad) : 64;
                        wn += run; ca += run; cb += run;
                        if (cb == m) break;
                        if (!last && (ca >= kGaCommit || cb >= kGaCommit)) break;
                        if (ca == n) { put('D'); ++cb; --d; ++dist; continue; }
                        // here lane `run` is inside the window and its match test failed: its edit is the walk's next step
                        const int op = __builtin_amdgcn_readlane(code, run);
                        if (op == 0) { status = AIM_PAIR_WFA_NO_LINK; break; }   // cannot happen (the recurrence guarantees one rule applies)
                        put(op);
                        ca += op != 'D';
                        cb += op != 'I';
                        --d; ++dist;
                    }
                };
                // The same walk over the banded words of a regular window (LONG): m = n = 64, not the last window, so the walk ends on the
                // commit bound. The lanes stay with their COLUMNS (lane j: column 63 - j, as in the sweep). Every cell of a run of matches lies
                // on the walk's diagonal k = ca - cb, whose band bit is kb = 15 + k in every column, and "p[b] == t[a] and ok(a+1, b+1, d)" is
                // bit kb of (c_{a+1}[d] & eq_a) -- the recurrence's own match term: one DPP shift of the level's words, one AND, one ballot,
                // and the run is the string of set bits from column ca on (a count of leading ones). The edit after the run is decided by
                // three bits of level d-1 at the one cell where the run stopped: two v_readlane of that level's words, the rest scalar.
                // Words are complemented: bit set = clear(r, b). One LDS read per edit (the next level's words), nothing per match.
                auto walk_cols = [&]() {
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
                if (slow) walk(std::true_type{});
                else if constexpr (LONG) walk_cols();
                else walk(std::false_type{});
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

constexpr uint64_t kGaSlabBytes = (uint64_t)(kGaW + 1) * 64 * 8;   // one wavefront's slow-path columns in HBM scratch

// The LONG variant sends every window that is not regular (the pair's last one, and one or two before it) through the 64-level path and runs
// the regular ones 2-3x faster. Same box, kernel ms standard / LONG (tools/ga_sweep.py, e = 10 %, profiles/r04/ga_sweep.txt): l=100 1.86 / 4.86;
// l=200 2.11 / 1.93; l=300 3.24 / 2.53; l=500 2.76 / 1.45; l=1000 5.53 / 2.45; l=5000 7.15 / 2.43 -- LONG from READ_SIZE 224 (l = 200) up.
// Residency: 24 wavefronts per CU beat 32 (l=10000, 8 192 pairs: 1.60 / 1.74 ms; 16: 1.95).
inline bool genasm_long(const aim_params_t &p, const Knobs &kn)
{
    return kn.ga_long >= 0 ? kn.ga_long != 0 : p.read_size >= 224;
}

inline void genasm_plan(const aim_params_t &p, const Knobs &kn, uint32_t n_pairs, uint32_t *grid, uint32_t *block, size_t *lds)
{
    const bool lg = genasm_long(p, kn);
    *block = kWave;
    *lds = lg ? kGlLdsBytes : (size_t)kGaCols * kGaSlots * 8 + 64;
    uint32_t per_cu = (uint32_t)std::min<size_t>(lg ? 24 : 16, lds_workgroups_per_cu(*lds));   // (the standard variant's 13 KB: 11; capped at 8 until round 3)
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
    if (genasm_long(p, kn)) {
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
