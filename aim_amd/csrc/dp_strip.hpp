// dp_strip.hpp -- long-read NW / SWG (BASELINE config 4) as a COLUMN-STRIP PIPELINE: one pair per workgroup, every wavefront owns
// a strip of 64 * K consecutive columns, every lane K consecutive cells of it, and the previous row of those cells stays in the
// lane's REGISTERS as packed int16 pairs. Rows flow through the wavefronts like a systolic array: wavefront w works on row h
// while wavefront w + 1 is still on row h - 1; what crosses a strip boundary -- the running prefix minimum of the in-row gap
// chain and one diagonal cell -- travels through small LDS mailboxes guarded by sequence numbers. No workgroup barrier inside
// the row loop, no LDS row buffers, no int16 pack / unpack, two cells per vector instruction (v_pk_add_i16 / v_pk_min_i16).
//
// Same results as nw_compute / swg_compute (NW/DPU-WRAM/dpu/nw.c:109-153, SWG/DPU-WRAM/dpu/swg.c:121-171) by the argument of
// dp_wave.hpp: the in-row chain  D[v] = min(M[v-1] + o + e, D[v-1] + e)  (NW: R[v] = min(A[v], R[v-1] + g))  is the prefix
// minimum  D[v] = v e + min_{j < v} G[j],  G[j] = A[j] + o + e - (j + 1) e,  A = min(diag + cost, I)  -- equal to the reference's
// cell-by-cell int16 arithmetic exactly when no int16 store can wrap, which dp_strip_exact_ok() proves before this path is
// taken (it also bounds G, so that the PACKED 16-bit arithmetic cannot wrap either). The reference's flat-table aliasing for
// plen > tlen (dp_wave.hpp, header) makes cell (h, W) the boundary cell of row h + 1: the lane that owns column W - 1 computes
// it and posts it to the first wavefront, so a "tailed" pair runs its rows strictly one after the other (the dependence is
// the reference's own) while a pair without tail pipelines freely. Pairs outside the preconditions take dp_wave.hpp's literal
// single-lane path; the traceback is dp_wave.hpp's walk over the canonical table slab (unchanged layout).
//
// round 2's dp_wave_kernel (row scan, three workgroup barriers per row, 42 lane-operations per cell) stays available behind
// AIM_DPW_LEGACY=1.
#pragma once

#include "aim_device.hpp"
#include "dp_wave.hpp"
#include "wfa_lane.hpp"   // LANE_TODO_* (to-do mode)

namespace aim {

typedef short dps2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dps2 dps_from(uint32_t u) { return __builtin_bit_cast(dps2, u); }
__device__ __forceinline__ uint32_t dps_bits(dps2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ dps2 dps_splat(int x) { dps2 r; r.x = (short)x; r.y = (short)x; return r; }
__device__ __forceinline__ dps2 dps_min(dps2 a, dps2 b) { return __builtin_elementwise_min(a, b); }

// "characters differ" per 16-bit field as 0 / 1: unsigned minimum of the xor with 1. Inline assembly on purpose: written with
// __builtin_elementwise_min the compiler canonicalises umin(x, 1) into a compare + select per HALF (v_cmp_ne_u16_sdwa + v_cndmask:
// four instructions and a scalar mask pair instead of one v_pk_min_u16).
__device__ __forceinline__ uint32_t pk_ne01(uint32_t a, uint32_t b, uint32_t ones)
{
    uint32_t r;
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(a ^ b), "v"(ones));
    return r;
}
// keep a value in its vector register: the optimiser must not re-derive it from the comparison it came from inside the row loop
__device__ __forceinline__ void opaque(uint32_t &x) { asm volatile("" : "+v"(x)); }

constexpr int kStripDepth = 8;        // mailbox ring slots per strip (rows a strip may run ahead of its slowest reader)
constexpr int kStripMaxWaves = 16;
constexpr short kInf16 = 0x7fff;

// Can any int16 store of the row-scan formulation wrap (dp_wave_exact_ok), or any PACKED intermediate (G = A + o + e - (v + 1) e)?
__host__ __device__ inline bool dp_strip_exact_ok(const aim_params_t &p, bool swg_int8)
{
    if (!dp_wave_exact_ok(p, swg_int8)) return false;
    const long rs = p.read_size;
    const long ge = p.algo == AIM_ALGO_NW ? (p.gap_i > p.gap_d ? p.gap_i : p.gap_d) : p.gap_e;
    const long lo = (long)p.match * rs - (rs + 2) * ge - 4L * (p.gap_o + p.gap_e);   // smallest G (match <= 0)
    return lo > -32000;
}

__host__ __device__ inline int dp_strip_stride(int rs, int k) { return (rs + k + 16) & ~7; }

// one 8-byte mailbox word {value, sequence number}; written with one ds_write_b64, polled until the sequence number matches.
// The pointers are LDS-typed (address_space(3)) on purpose: a volatile access through a GENERIC pointer compiles to a flat
// instruction with system-scope cache policy and a full s_waitcnt vmcnt(0) -- i.e. every mailbox operation waited for all of the
// wavefront's outstanding table stores (the first version of this kernel did exactly that).
struct StripMsg { int val; int seq; };
typedef int dp_i2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) volatile dp_i2 *lds_msg_p;
typedef __attribute__((address_space(3))) volatile int *lds_int_p;
__device__ __forceinline__ void strip_post(lds_msg_p m, int val, int seq)
{
    dp_i2 v; v.x = val; v.y = seq;
    *m = v;
}
__device__ __forceinline__ int strip_wait(lds_msg_p m, int seq)
{
    for (;;) {
        const dp_i2 v = *m;
        if (v.y == seq) return v.x;
        __builtin_amdgcn_s_sleep(1);
    }
}

#ifdef AIM_STRIP_STAMPS   // diagnostic builds only (tools/strip_stamps.py): s_memtime per phase of a row, summed per wavefront, dumped into the pair's ops row
#define AIM_SSTAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); ssum[i] += t_ - slast; slast = t_; } while (0)
#else
#define AIM_SSTAMP(i) do { } while (0)
#endif

// swg_traceback (swg.c:45-119) over FOUR DIRECTION BITS per cell, decided at fill time (round 5) -- no value plane at all:
// 0.5 - 0.8 instead of 2.25 bytes per cell (config 4 wrote 61.8 GB per 256 pairs, 52 GB of it the int16 M plane). The four tests of the reference's
// walk at a cell are known when the cell is computed, because every cell the walk compares with still holds the value the fill read (the flat table's
// aliasing only replaces boundary cells B(h + 1) = tail cell (h, W), and B(h + 1) is final before row h + 1 starts; dp_strip_exact_ok: nothing wraps, so
// an operation is always found):
//     bit "M != D"        A < D                      (layer M, first test)
//     bit "M != I"        A < I                      (layer M, second test; then 'M' / 'X' is the character comparison, MISMATCH != MATCH)
//     bit "I extended"    I_up + e < M_up + o + e    (layer I: I == M_up + o + e fails)
//     bit "next D ext."   pre(v) < G(v)  <=>  D[v + 1] != M[v] + o + e  (layer D at cell v + 1; o > 0) -- kept at the cell on the LEFT of the one it belongs
//                         to, because that is the lane that knows it (the prefix minimum up to v and G(v)); the walk looks one cell to the left.
// Per row and lane NQS dwords (dword q: registers 4q .. 4q + 3; byte b of it: low nibble bit k = register 4q + k's "M != D" (b = 0, 1: low / high half) or
// "M != I" (b = 2, 3), high nibble: "next D extended" / "I extended"), row h's words at FLW[(h * FS + lane) * NQS]; boundary cells (column 0) in BF[row]:
// bits 0 - 3 the cell's own four tests (D extended = its OWN), bit 4 "D of column 1 extended".
// The lane word of direction bits: SWG four bits per cell -> a dword per FOUR registers (bit k of the low nibbles = register 4q + k's M-layer tests, high nibbles the
// two gap tests); NW has the M-layer tests only, so (round 6) a dword holds EIGHT registers -- bit k of every byte, k = 0 .. 7 -- and a lane word is half as long
// (K = 32: 8 instead of 16 bytes per lane and row; medium reads with CIGAR spent a seventh of their time storing them).
template <int K, bool SWG> struct DpBits {
    static constexpr int KP = K / 2;
    static constexpr int RSH = SWG ? 2 : 3;                       // registers per dword, as a shift ...
    static constexpr int RM = (1 << RSH) - 1;                     // ... and a mask
    static constexpr int NQ = (KP + RM) >> RSH;                   // dwords that hold a lane's K cells
    static constexpr int NQS = NQ == 3 ? 4 : NQ;                  // dwords per lane word as stored (a 12-byte word is stored as 16)
};
template <int NQS> struct DpWord { typedef uint4 type; };
template <> struct DpWord<2> { typedef uint2 type; };
template <> struct DpWord<1> { typedef uint32_t type; };

template <int K, bool SWG, bool PF = false>   // PF: the next 64-row window is prefetched (dp_group_kernel; costs 12 registers)
__device__ __forceinline__ bool dp_traceback_swg_bits(const aim_params_t &p, int plen, int tlen, int FS, const uint32_t *FLW, const unsigned char *BF,
                                                      const unsigned char *ldsP, const unsigned char *ldsT, uint32_t *tile, int tile_rows, char *ops, int lane,
                                                      int &begin_offset, bool banded, int band_lo, int band_hi)
{   // banded: direction bits exist only where C - R lies in [band_lo, band_hi] (the strips around the diagonal, dp_strip_kernel); returns true when the walk left it
    constexpr int NQS = DpBits<K, SWG>::NQS, RSH = DpBits<K, SWG>::RSH, RM = DpBits<K, SWG>::RM;
    const int kTR = tile_rows;                                    // rows of the window: 64 or 256 (what the workgroup's LDS admits)
    constexpr int kTW = 3;                                        // lane words per row of the window: a BAND around the diagonal through the cell it was filled at
    const int rs = p.read_size, W = tlen + 1;
    int sentinel = plen + tlen - 1;
    int h = tlen, v = plen;
    const int cap = 2 * rs;
    auto put = [&](char ch) { if (lane == 0 && sentinel >= 0 && sentinel < cap) ops[sentinel] = ch; --sentinel; };
    // The window: rows tR .. tR - kTR + 1; row tR - rr holds the lane words wbase(rr) .. wbase(rr) + 2, wbase(rr) = word of column (tC - rr) minus one -- the band
    // follows the diagonal through (tR, tC), which is where the walk goes (a gap moves it by one column: ~K columns of slack either side). 256 rows are 12 KB and
    // 12 loads per lane; the rectangular window of 128 rows x 8 words was 16 KB, 16 loads, and was refilled twice as often.
    int tR = -1, tC = 0;
    auto wbase_at = [&](int c0, int rr) { const int cc = c0 - rr; const int g = ((cc > 1 ? cc : 1) - 1) / K; return g > 0 ? g - 1 : 0; };
    auto wbase = [&](int rr) { return wbase_at(tC, rr); };
    // Round 6: a window of 64 rows is three loads per lane, and the NEXT window -- the 64 rows below, along the same diagonal -- is asked for as soon as the current one
    // is in LDS: by the time the walk reaches its last row the loads have long landed in registers, and the refill is a wait + three LDS stores instead of a round trip
    // to HBM every 32 - 64 steps (medium reads: 5 pairs per wavefront walked one after the other, ~10 refills of ~2 us each per pair = a seventh of NW l = 300 with
    // CIGAR). A walk that gaps its way out of the prefetched band (~K columns either side) falls back to a direct refill around its own position.
    typedef typename DpWord<NQS>::type word_t;
    const bool pf_on = PF && kTR == kWave;                        // (the 256-row window of the long-read strips is 12 loads per lane: not prefetched)
    word_t pf[PF ? kTW : 1];
    int pfR = -1, pfC = 0;
    auto prefetch = [&](int R, int C) {
        pfR = -1;
        if (!pf_on || R < 1) return;
        pfR = R; pfC = C;
#pragma unroll
        for (int i = 0; i < (PF ? kTW : 0); ++i) {
            const int q = lane + i * kWave, rr = q / kTW, w = q - rr * kTW, r = R - rr, gg = wbase_at(C, rr) + w;
            pf[i] = word_t{};
            if (r >= 0 && gg < FS) pf[i] = *reinterpret_cast<const word_t *>(&FLW[((size_t)r * FS + gg) * NQS]);
        }
    };
    auto refill = [&](int R, int C) {
        const int g = (C > 1 ? C - 1 : 0) / K;
        bool hit = false;
        if (pf_on && pfR >= 0 && R <= pfR && R > pfR - kTR) { const int w = g - wbase_at(pfC, pfR - R); hit = w >= 0 && w < kTW; }
        if (hit) {   // the prefetched window holds the cell: wait for it, move it into LDS
            tR = pfR; tC = pfC;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < (PF ? kTW : 0); ++i) *reinterpret_cast<word_t *>(&tile[(lane + i * kWave) * NQS]) = pf[i];
        } else {
            tR = R; tC = C;
            for (int q = lane; q < kTR * kTW; q += kWave) {
                const int rr = q / kTW, w = q - rr * kTW, r = R - rr, gg = wbase(rr) + w;
                if (r >= 0 && gg < FS) *reinterpret_cast<word_t *>(&tile[q * NQS]) = *reinterpret_cast<const word_t *>(&FLW[((size_t)r * FS + gg) * NQS]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        prefetch(tR - kTR, tC - kTR);
    };
    auto in_tile = [&](int R, int g) { if (!(tR >= 0 && R <= tR && R > tR - kTR)) return false; const int w = g - wbase(tR - R); return w >= 0 && w < kTW; };
    auto tilebits = [&](int R, int C, int g) -> uint32_t {        // regular cell (R, C), C >= 1, inside the window: bit 0 M != D, bit 4 next D extended, bit 16 M != I, bit 20 I extended
        const int t = (C - 1) - g * K, j = t >> 1, rr = tR - R;
        return tile[(rr * kTW + (g - wbase(rr))) * NQS + (j >> RSH)] >> (8 * (t & 1) + (j & RM));
    };
    auto cellbits = [&](int R, int C) -> uint32_t {               // wave-uniform (R, C)
        const int g = (C - 1) / K;
        if (!in_tile(R, g)) refill(R, C);
        return tilebits(R, C, g);
    };
    // (R, C): the canonical position of flat index W h + v (rows of W cells); moves: D at - 1, I at - W, diagonal at - W - 1
    int R = W ? (W * h + v) / W : 0, C = (W * h + v) - R * W;
    int layer = 0;                                                // 0: M, 1: I, 2: D
    while (h > 0 && v > 0) {
        // (the cells of a diagonal run share C - R; the D layer looks at C - 1: the band handed in is one narrower. A pair with tail cells lands, from the last
        //  row's tail, in the first plen - tlen columns of the last rows -- flat indices beyond W -- and leaves them through column 0: those cells have bits too)
        if (banded && C >= 1 && R <= tlen && (C - R < band_lo || C - R > band_hi) && !(plen > tlen && R >= tlen - (plen - tlen) - 1 && C <= plen - tlen + 1)) return true;
        if (layer == 0 && C >= 1 && R <= tlen) {
            // A RUN OF DIAGONAL MOVES, 64 cells at a time: lane i looks at cell (R - i, C - i) -- inside the table, inside the window, both "M != D" and "M != I" --
            // and the run is the leading lanes that pass; each writes its own 'M' / 'X' (the cell's own characters). At e = 1 % a run is ~100 cells:
            // one lane stepping through them (and the window's refills every 31 rows) was 4.6 of config 4's 23.8 ms.
            const int g0 = (C - 1) / K;
            if (!in_tile(R, g0) || (!pf_on && tR - kTR + 1 > 1 && R - (tR - kTR + 1) < kWave)) refill(R, C);   // (the 256-row window keeps 64 rows above the current one inside it; the 64-row window is followed by its prefetched successor: a run is cut at its last row and goes on after the switch)
            const int ri = R - lane, ci = C - lane;
            bool ok = lane < h && lane < v && ci >= 1 && ri >= 1;
            const int gi = ok ? (ci - 1) / K : 0;
            ok = ok && in_tile(ri, gi);
            uint32_t b = 0u;
            if (ok) b = tilebits(ri, ci, gi);
            ok = ok && (b & 1u) && (b & 0x10000u);
            const unsigned long long okm = __ballot(ok);
            const int run = okm == ~0ull ? kWave : __builtin_ctzll(~okm);
            if (run > 0) {
                if (lane < run) {
                    const int at = sentinel - lane;
                    if (at >= 0 && at < cap) ops[at] = ldsP[ci - 1] != ldsT[ri - 1] ? 'X' : 'M';
                }
                sentinel -= run; h -= run; v -= run; R -= run; C -= run;
                continue;
            }
        }
        uint32_t nD, nI, xD, xI;
        if (C >= 1) {
            const uint32_t b = cellbits(R, C);
            nD = b & 1u; nI = (b >> 16) & 1u; xI = (b >> 20) & 1u;
            xD = 0u;
            if (layer == 2) xD = C == 1 ? (((uint32_t)BF[R] >> 4) & 1u) : ((cellbits(R, C - 1) >> 4) & 1u);
        } else {
            const uint32_t b = BF[R];
            nD = b & 1u; nI = (b >> 1) & 1u; xD = (b >> 2) & 1u; xI = (b >> 3) & 1u;
        }
        if (!SWG && !nD) layer = 2;                           // NW (nw_traceback, nw.c:67-107: "== left + GAP_D", then "== up + GAP_I", else the diagonal): a gap move is
        else if (!SWG && !nI) layer = 1;                      // one operation, no layers -- the two gap branches below, taken at once, "extended" never consulted
        if (layer == 2) { put('D'); if (!SWG || !xD) layer = 0; --v; if (C > 0) --C; else { --R; C = W - 1; } }
        else if (layer == 1) { put('I'); if (!SWG || !xI) layer = 0; --h; --R; }
        else if (!nD) layer = 2;
        else if (!nI) layer = 1;
        else {   // diagonal move: the cell equals its diagonal + MATCH or + MISMATCH according to the characters IT was computed with -- canonical cell (R, C) was
                 // last written as cell (R, C) of the table (C >= 1, R <= tlen) or as the tail cell (R - 1, W + C) (column 0; row tlen + 1)
            //   ... or, beyond row tlen + 1 (plen > 2 tlen), as the last row's tail cell (R - tlen) W + C
            const int pi = R > tlen ? (R - tlen) * W + C - 1 : (C == 0 ? W - 1 : C - 1), ti = R > tlen ? tlen - 1 : (C == 0 ? R - 2 : R - 1);
            put(ldsP[pi] != ldsT[ti] ? 'X' : 'M');
            --h; --v; --R; if (C > 0) --C; else { --R; C = W - 1; }
        }
    }
    for (int i = lane; i < h; i += kWave) { const int at = sentinel - i; if (at >= 0 && at < cap) ops[at] = 'I'; }
    if (h > 0) sentinel -= h;
    for (int i = lane; i < v; i += kWave) { const int at = sentinel - i; if (at >= 0 && at < cap) ops[at] = 'D'; }
    if (v > 0) sentinel -= v;
    begin_offset = sentinel + 1;
    return false;
}

// K: cells per lane; NWMAX: the most wavefronts a workgroup of this instantiation is launched with (its register budget:
// 4 / 8 wavefronts 256 VGPRs, 12 -> 168, 16 -> 128)
template <int ALGO, bool BT, int K, int NWMAX>
__global__ __launch_bounds__(64 * NWMAX) void dp_strip_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    constexpr bool SWG = (ALGO == AIM_ALGO_SWG);
    constexpr int KP = K / 2;
    const int tid = threadIdx.x, NT = blockDim.x;
    const int lane = tid & (kWave - 1), wv = tid >> 6, nw = NT >> 6;
    const int rs = a.p.read_size;
    // LDS: pattern | text | last-row dump M, I (also the traceback's tile) | mailboxes
    const int seqcap = (rs + 79) & ~15;
    unsigned char *ldsP = reinterpret_cast<unsigned char *>(smem);
    unsigned char *ldsT = ldsP + seqcap;
    const int rowcap = (rs + 47) & ~7;
    int16_t *rowM = reinterpret_cast<int16_t *>(ldsT + seqcap);
    int16_t *rowI = rowM + rowcap;
    const int rowbytes = (2 * rowcap * 2 > 8 * 1024 ? 2 * rowcap * 2 : 8 * 1024);   // (the tracebacks' tiles live here afterwards: 8 KB, 16 KB where the rows leave as much)
    const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char *)smem;   // LDS byte offset of the dynamic segment
    const uint32_t mb0 = lds0 + (uint32_t)(2 * seqcap + rowbytes);
    const lds_msg_p mbC = (lds_msg_p)(uintptr_t)mb0;                                                  // [kStripMaxWaves][kStripDepth]: strip w's minimum of G over its columns, per row
    const lds_msg_p mbD = mbC + kStripMaxWaves * kStripDepth;                                          // ... diagonal cell M[h-1][c0_w - 1]
    const lds_int_p cons = (lds_int_p)(uintptr_t)(mb0 + 2 * kStripMaxWaves * kStripDepth * 8);         // [kStripMaxWaves] last row whose messages strip w has consumed
    const lds_int_p Bl = cons + kStripMaxWaves;                                                        // [kStripDepth] boundary cells {M, I, D, seq} by row (a ring: a strip
                                                                                                       // may still be waiting for B(h) when the owner posts B(h+1))
    const lds_int_p tl = Bl + 4 * kStripDepth;                                                                       // last row's tail inputs {M, D of cell W-1, diag} + score
    // The workgroup's slab holds FOUR DIRECTION BITS per cell (NW: two of them) -- FLW [row][FS lanes][NQS dwords] + the boundary cells' bytes BF [row] -- and no
    // value plane. (Round 5 kept a POOL of int16 tables behind the slabs, taken under a lock, for pairs with plen > 2 tlen, which ONE lane filled literally; round 6
    // computes those pairs' tail cells like everybody else's -- the last row's tail loop below -- and the pool, its lock and the cross-XCD release / acquire it needed are gone.)
    constexpr int NQS = DpBits<K, SWG>::NQS, RSH = DpBits<K, SWG>::RSH, RM = DpBits<K, SWG>::RM;   // dwords of direction bits per lane and row (DpBits)
    int16_t *tb = reinterpret_cast<int16_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave);
    const int FS = rs / K + 2;                    // lanes per row that can hold a column
    uint32_t *FLW = reinterpret_cast<uint32_t *>(tb);             // SWG strip path: direction bits
    unsigned char *BF = reinterpret_cast<unsigned char *>(FLW + (size_t)(rs + 3) * FS * NQS);   // ... and the boundary cells' bytes, [row]
    const int O = a.p.gap_o, E = a.p.gap_e, OE = O + E, MATCH = a.p.match, MISMATCH = a.p.mismatch;
    const int GD = a.p.gap_d, GI = a.p.gap_i, MAXS = a.p.max_score;
    const int GE = SWG ? E : GD;                 // step of the in-row chain
    const int v0 = 1 + (wv * kWave + lane) * K;  // first column of this lane

    // to-do mode (a.todo set: the pairs dp_group_kernel left, dp_group.hpp): the units are the listed pairs
    const uint32_t n_work = a.todo ? a.todo[LANE_TODO_COUNT] : a.n_pairs;
    for (uint32_t it = 0;; ++it) {
        uint32_t pair;
        if (!xcd_unit(n_work, it, &pair)) break;
        if (a.todo) pair = a.todo[LANE_TODO_LIST + pair] - a.pair_base;
        const aim_request_t rq = load_request(a, pair);
        const int plen = rq.pattern_len, tlen = rq.text_len;
        const unsigned char *gP = reinterpret_cast<const unsigned char *>(a.patterns + (uint64_t)pair * rs);
        const unsigned char *gT = reinterpret_cast<const unsigned char *>(a.texts + (uint64_t)pair * rs);
        char *ops = BT ? a.ops + (uint64_t)pair * 2 * rs : nullptr;
#ifdef AIM_STRIP_STAMPS
        unsigned long long ssum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, slast = 0;
#endif
        const int W = tlen + 1;
        int score = 0, status = AIM_PAIR_OK;
        int begin_offset = plen + tlen - 1;
        const int end_offset = plen + tlen;
        __syncthreads();                          // the previous pair's traceback is done with the LDS it used
        if (BT && SWG) {   // memset(cigar->operations, 'M', 2*READ_SIZE), swg.c:261
            uint32_t *o4 = reinterpret_cast<uint32_t *>(ops);
            for (int w = tid; w < (rs >> 1); w += NT) o4[w] = 0x4D4D4D4Du;
        }
        // ---------------------------------------------------------------------------------------- strip pipeline
        // With CIGAR the direction bits are made only by the strips AROUND THE DIAGONAL (a band of +- 32 K columns about the two diagonals through the table's
        // corners: two or three of config 4's eight strips per row) -- the four tests cost ~25 % of a row and sit on the pipeline's critical path, and the walk
        // of related reads never leaves the band. When it does (the walk checks C - R at every step), the pair is filled again with every strip's bits.
        for (int attempt = 0; attempt < 2; ++attempt) {
            constexpr bool BAND = BT && K == 20;      // (the shapes with registers to spare for a second copy of the row: K = 24 / 32 and 16 x 12 spill with it)
            const bool full_bits = !BAND || attempt == 1 || nw == 1;
            const int band_m = 32 * K, band_d0 = min(0, plen - tlen), band_d1 = max(0, plen - tlen);
            const int strip_c0 = 1 + wv * kWave * K, strip_c1 = strip_c0 + kWave * K - 1;
            for (int i = tid * 4; i < seqcap; i += NT * 4) {   // sequences into LDS, zero beyond their length (dword granularity)
                uint32_t wp = 0, wt = 0;
                for (int b = 0; b < 4; ++b) {
                    if (i + b < plen) wp |= (uint32_t)gP[i + b] << (8 * b);
                    if (i + b < tlen) wt |= (uint32_t)gT[i + b] << (8 * b);
                }
                *reinterpret_cast<uint32_t *>(ldsP + i) = wp;
                *reinterpret_cast<uint32_t *>(ldsT + i) = wt;
            }
            for (int i = tid; i < 2 * kStripMaxWaves * kStripDepth; i += NT) strip_post(mbC + i, 0, 0);   // (mbD follows mbC)
            if (tid < kStripMaxWaves) cons[tid] = 0;
            if (tid < kStripDepth) Bl[4 * tid + 3] = 0;
            const int Rr = min(plen, W - 1);          // regular columns 1..Rr
            const bool has_tail = plen >= W;
            // row 0 in registers (and its table image); boundary column of the table
            dps2 Mp[KP], Ip[KP], cD[KP], vmask[KP];
            int nvalid = Rr - v0 + 1;                  // cells of this lane inside the row
            nvalid = nvalid < 0 ? 0 : (nvalid > K ? K : nvalid);
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const int va = v0 + 2 * j, vb = va + 1;
                dps2 m, d;
                if (SWG) { m.x = (short)(O + va * E); m.y = (short)(O + vb * E); }
                else { m.x = (short)(va * GD); m.y = (short)(vb * GD); }
                d.x = (short)(va * GE); d.y = (short)(vb * GE);
                Mp[j] = m;
                Ip[j] = dps_splat(MAXS);
                cD[j] = d;
                dps2 vm; vm.x = (short)((2 * j < nvalid) ? -1 : 0); vm.y = (short)((2 * j + 1 < nvalid) ? -1 : 0);
                uint32_t vmb = dps_bits(vm);
                opaque(vmb);
                vmask[j] = dps_from(vmb);
            }
            if (BT && SWG) {   // row-init boundary cells {M = I = o + h e, D = MAX_SCORE}: the walk only ever asks whether column 1's D was extended from them (NW: nothing)
                for (int h = 1 + tid; h <= tlen + 1; h += NT) BF[h] = (unsigned char)((O + h * E) + O <= MAXS ? 0 : 16);
            }
            // (the row-init stores above vs the tail owner's store to the same boundary cell below: write after write across
            // wavefronts through HBM -- made explicit exactly as in dp_wave.hpp: every store has completed before the barrier)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            // this lane's pattern characters as 16-bit fields, two per dword (the cost of a cell is a packed compare away)
            uint32_t pc16[KP];
#pragma unroll
            for (int j = 0; j < KP; ++j) pc16[j] = (uint32_t)ldsP[v0 - 1 + 2 * j] | ((uint32_t)ldsP[v0 + 2 * j] << 16);
            const bool tail_owner = has_tail && v0 <= Rr && Rr < v0 + K;   // this lane owns column W - 1 = Rr
            const int tail_t = Rr - v0;                                     // ... as its cell tail_t
            const int pchW = has_tail ? (int)ldsP[W - 1] : 0;
            // boundary cell of the current row (flat[W*h]): analytic unless the pair has a tail
            int BM = 0, BI = 0, BD = 0, BMprev = 0;                          // B(h) and B(h-1).M (column 0 of the previous row; row 0: 0)
            const dps2 OEp = dps_splat(OE), Ep = dps_splat(E), GIp = dps_splat(GI);
            dps2 c1[KP];                                                    // G = A - c1: SWG (v+1)e - (o+e); NW v g
#pragma unroll
            for (int j = 0; j < KP; ++j) c1[j] = cD[j] + dps_splat(SWG ? (E - OE) : 0);
            const dps2 costD = dps_splat(MISMATCH - (SWG ? MATCH : 0)), costM = dps_splat(SWG ? MATCH : 0);
            uint32_t ones = 0x00010001u;
            opaque(ones);
            int tailM_M = 0, tailM_D = 0, tail_diag = 0;                    // the tail owner's {M, D} of cell (h, W-1) and M of (h-1, W-1)
            const bool owner_wave = __ballot(tail_owner) != 0ull;           // wave-uniform: this wavefront holds the tail owner
            const bool wave_full = __ballot(nvalid != K) == 0ull;           // wave-uniform: every cell of this wavefront lies inside the row
            int safe_row = 0;                                               // rows up to safe_row + kStripDepth may be posted without asking (lane 0)
            // cell t of a packed register row, t a per-lane value: a binary select tree over the dwords (a dynamic register index
            // would go through scratch; a flat chain of K compares keeps K lane masks alive in scalar registers)
            auto pick = [&](const dps2 (&arr)[KP], int t) {
                uint32_t v[KP];
#pragma unroll
                for (int j = 0; j < KP; ++j) v[j] = dps_bits(arr[j]);
                const int d = t >> 1;
#pragma unroll
                for (int step = 1; step < KP; step <<= 1) {
                    const bool odd = (d & step) != 0;
#pragma unroll
                    for (int j = 0; j + step < KP; j += 2 * step) v[j] = odd ? v[j + step] : v[j];
                    if ((KP / step) & 1) { /* an unpaired block stays where it is */ }
                }
                const uint32_t w = v[0];
                return (t & 1) ? (int)(int16_t)(w >> 16) : (int)(int16_t)(w & 0xffffu);
            };

#ifdef AIM_STRIP_STAMPS
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(slast) :: "memory");
#endif
            auto row = [&](auto gen_tag, int h) __attribute__((always_inline)) {
                constexpr bool GEN = decltype(gen_tag)::value;   // this strip makes the row's direction bits
                AIM_SSTAMP(7);   // loop back-edge
                const int slot = h & (kStripDepth - 1);
                const uint32_t tch2 = (uint32_t)ldsT[h - 1] * 0x00010001u;
                // ---- diagonal input of this lane's first cell: M[h-1][v0 - 1]
                int dfirst;
                {
                    const int mine_last = (int)Mp[KP - 1].y;
                    dfirst = __builtin_amdgcn_update_dpp(0, mine_last, 0x138, 0xf, 0xf, false);   // wave_shr:1
                    if (lane == 0) {
                        if (wv == 0) dfirst = BMprev;
                        else if (h == 1) dfirst = SWG ? O + (v0 - 1) * E : (v0 - 1) * GD;
                        else dfirst = strip_wait(mbD + wv * kStripDepth + slot, h);
                    }
                }
                AIM_SSTAMP(0);   // diagonal input (mailbox D)
                // ---- pre-carry: I, A, G of this lane's K cells, two per instruction (nothing here depends on this row's carry or
                // boundary cell: in a tailed pair it runs while the previous row's last strip is still finishing)
                dps2 A[KP], Iv[KP], G[KP];
                uint32_t fw[4] = {0u, 0u, 0u, 0u};                   // BT, SWG: the row's direction bits, assembled as the tests become known (layout: dp_traceback_swg_bits)
                dps2 gmin = dps_splat(kInf16);
                auto precarry = [&](auto MASKED) {
#pragma unroll
                    for (int j = 0; j < KP; ++j) {
                        const uint32_t up = dps_bits(Mp[j]);
                        const uint32_t prev = j ? dps_bits(Mp[j - 1]) : ((uint32_t)(uint16_t)dfirst << 16);
                        const dps2 diag = dps_from(__builtin_amdgcn_alignbit(up, prev, 16));       // {M[v-1], M[v]} of the previous row
                        // cost per cell: characters differ -> 1 (unsigned min with 1), times (MISMATCH - MATCH), plus MATCH
                        const dps2 f = dps_from(pk_ne01(pc16[j], tch2, ones));
                        const dps2 sub = f * costD + (diag + costM);
                        dps2 ins;
                        if (SWG) {
                            const dps2 insn = Mp[j] + OEp, inse = Ip[j] + Ep;
                            ins = dps_min(insn, inse);
                            if (BT && GEN) {   // "I extended": I_up + e < M_up + o + e (sign byte of the saturating difference -> bytes 2, 3, high nibble)
                                const uint32_t sI = __builtin_amdgcn_perm(0u, dps_bits(__builtin_elementwise_sub_sat(inse, insn)), 0x09080c0cu);
                                fw[j >> 2] |= sI & (0x10100000u << (j & 3));
                            }
                        }
                        else ins = Mp[j] + GIp;
                        Iv[j] = ins;
                        A[j] = dps_min(sub, ins);
                        dps2 g = A[j] - c1[j];
                        if (decltype(MASKED)::value) g = dps_from((dps_bits(g) & dps_bits(vmask[j])) | (0x7fff7fffu & ~dps_bits(vmask[j])));
                        G[j] = g;
                        gmin = dps_min(gmin, g);
                    }
                };
                if (wave_full) precarry(std::false_type{});
                else precarry(std::true_type{});
                AIM_SSTAMP(1);   // pre-carry
                int lane_min = min((int)gmin.x, (int)gmin.y);
                if (lane_min == kInf16) lane_min = kDpInf;
                int total;
                const int lane_pre = wave_excl_scan_min(lane_min, lane, &total);
                // ---- this strip's minimum to every strip on its right (a broadcast, not a chain: strip w reads the w totals on
                // its left in one LDS round trip)
                if (wv + 1 < nw && lane == 0) {
                    if (h - safe_row >= kStripDepth) {    // the ring slot of row h was last used by row h - depth: has every strip on the right read it?
                        for (;;) {
                            int m = 0x7fffffff;
                            for (int u = wv + 1; u < nw; ++u) m = min(m, (int)cons[u]);
                            if (h - m < kStripDepth) { safe_row = m; break; }
                            __builtin_amdgcn_s_sleep(1);
                        }
                    }
                    strip_post(mbC + wv * kStripDepth + slot, total, h);
                }
                AIM_SSTAMP(6);   // wave scan + post C (incl. waiting for the ring slot)
                // ---- B(h): the boundary cell of this row
                if (h == 1 || !has_tail) {
                    if (SWG) { BM = O + h * E; BI = BM; BD = MAXS; }
                    else { BM = h * GI; BI = BD = 0; }
                } else {
                    const lds_int_p bs = Bl + 4 * slot;
                    while (bs[3] != h) __builtin_amdgcn_s_sleep(1);     // posted by the lane that owns column W - 1 at the end of row h - 1
                    BM = bs[0]; BI = bs[1]; BD = bs[2];
                }
                AIM_SSTAMP(2);   // boundary cell B(h)
                // ---- the prefix minimum over everything left of this strip: G[0] and the totals of the strips on the left
                int carry_in = SWG ? min(BD, BM + O) : BM;             // G[0]
                if (wv > 0) {
                    int t_u = kDpInf;
                    if (lane < wv) t_u = strip_wait(mbC + lane * kStripDepth + slot, h);
                    carry_in = min(carry_in, wave_min_i32(t_u));
                    if (lane == 0) cons[wv] = h;                        // every message of row h is read
                }
                AIM_SSTAMP(3);   // totals of the strips on the left (mailbox C)
                // ---- post-carry: D / R and M of the K cells; the new row replaces the old one in the registers
                const int pre = min(carry_in, lane_pre);
                dps2 c = dps_splat(pre);
                dps2 Do[KP];
                const int diag_keep = owner_wave ? pick(Mp, tail_t) : 0;   // M[h-1][W-1] (before the row is replaced)
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    dps2 s_; s_.x = kInf16; s_.y = G[j].x;
                    const dps2 prej = dps_min(c, s_);                      // {pre(2j), pre(2j+1)}
                    c = dps_splat(min((int)prej.y, (int)G[j].y));
                    if (BT && GEN && SWG) {   // "the next cell's D was extended": pre(v) < G(v) (-> bytes 0, 1, high nibble)
                        const uint32_t sD = __builtin_amdgcn_perm(0u, dps_bits(__builtin_elementwise_sub_sat(prej, G[j])), 0x0c0c0908u);
                        fw[j >> 2] |= sD & (0x00001010u << (j & 3));
                    }
                    Do[j] = prej + cD[j];
                    Mp[j] = dps_min(A[j], Do[j]);
                    if (SWG) Ip[j] = Iv[j];
                }
                // ---- hand the next row's diagonal cell to the next strip
                if (wv + 1 < nw && lane == kWave - 1) strip_post(mbD + (wv + 1) * kStripDepth + ((h + 1) & (kStripDepth - 1)), (int)Mp[KP - 1].y, h + 1);
                BMprev = BM;
                AIM_SSTAMP(4);   // post-carry + post D
                // ---- first tail cell (h, W): the boundary cell of row h + 1 (rows before the last; the last row's tail is walked below)
                if (owner_wave) {
                    const int upM = pick(Mp, tail_t), upD = pick(Do, tail_t);
                    tailM_M = upM; tailM_D = upD; tail_diag = diag_keep;
                    if (tail_owner && h < tlen) {
                        const int tch = (int)(tch2 & 0xffu);
                        int cM, cI, cDd;
                        if (SWG) {
                            cDd = min(upM + OE, upD + E);
                            cI = min(BM + OE, BI + E);
                            cM = min(diag_keep + ((pchW == tch) ? MATCH : MISMATCH), min(cI, cDd));
                        } else {
                            cI = BM + GI; cDd = upM + GD;   // ("ins" and "del" of nw.c:137-143; B(h + 1)'s I / D slots are not read by NW)
                            cM = min(diag_keep + ((pchW == tch) ? 0 : MISMATCH), min(cI, cDd));
                        }
                        const lds_int_p bs = Bl + 4 * ((h + 1) & (kStripDepth - 1));
                        bs[0] = cM; bs[1] = cI; bs[2] = cDd;
                        bs[3] = h + 1;                                     // (same-wavefront LDS writes complete in order)
                        if (BT) {
                            if (SWG) BF[h + 1] = (unsigned char)((cM != cDd ? 1 : 0) | (cM != cI ? 2 : 0) | (upD + E < upM + OE ? 4 : 0) | (BI + E < BM + OE ? 8 : 0) |
                                                                 (cM + O <= cDd ? 0 : 16));
                            else BF[h + 1] = (unsigned char)((cM != cDd ? 1 : 0) | (cM != cI ? 2 : 0));   // NW: "not D", "not I"
                        }
                    }
                }
                AIM_SSTAMP(5);   // tail cell / picks
                // ---- table (BT): 16-byte stores where the lane's cells are all inside the row (off the critical path: after the posts)
                if (BT && GEN && nvalid > 0) {   // four direction bits per cell (NW: the first two): the sign bytes of four saturating differences, gathered by v_perm_b32 (selectors
                                                 // 8 .. 11 replicate a source's sign bits) and merged per four registers; the two gap tests are in fw already
#pragma unroll
                    for (int j = 0; j < KP; ++j) {
                        const uint32_t dA = dps_bits(__builtin_elementwise_sub_sat(A[j], Do[j])), dB = dps_bits(__builtin_elementwise_sub_sat(A[j], Iv[j]));
                        const uint32_t w1 = __builtin_amdgcn_perm(dB, dA, 0x0b0a0908u);
                        fw[j >> RSH] |= w1 & (0x01010101u << (j & RM));
                    }
                    uint32_t *dst = FLW + ((size_t)h * FS + wv * kWave + lane) * NQS;
                    if constexpr (NQS == 4) *reinterpret_cast<uint4 *>(dst) = make_uint4(fw[0], fw[1], fw[2], fw[3]);
                    else if constexpr (NQS == 2) *reinterpret_cast<uint2 *>(dst) = make_uint2(fw[0], fw[1]);
                    else *dst = fw[0];
                }
            };
            for (int h = 1; h <= tlen; ++h) {
                // (wave-uniform) does this strip's column range meet the band of row h?
                const bool gen = full_bits || (strip_c1 >= h + band_d0 - band_m && strip_c0 <= h + band_d1 + band_m) ||
                                 (plen > tlen && h >= tlen - (plen - tlen) - 2 && strip_c0 <= plen - tlen + 2);   // (where the walk of a pair with tail cells lands: dp_traceback_swg_bits)
                if constexpr (BAND) {
                    if (gen) row(std::true_type{}, h);
                    else row(std::false_type{}, h);
                } else row(std::true_type{}, h);
            }
            AIM_SSTAMP(7);       // (the last row's table stores)
            // ---- after the last row: its regular part into LDS (score; the tail walk reads it), then the reference's tail cells
            {
                if (nvalid > 0) {
#pragma unroll
                    for (int t = 0; t < K; ++t)
                        if (t < nvalid) {
                            rowM[v0 + t] = (t & 1) ? Mp[t >> 1].y : Mp[t >> 1].x;
                            if (SWG) rowI[v0 + t] = (t & 1) ? Ip[t >> 1].y : Ip[t >> 1].x;
                        }
                }
                if (tail_owner) { tl[0] = tailM_M; tl[1] = tailM_D; tl[2] = tail_diag; }
                if (tid == 0) { rowM[0] = (int16_t)BM; if (SWG) rowI[0] = (int16_t)BI; }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // table / ops stores of every wavefront before the traceback reads them (dp_wave.hpp)
            __syncthreads();
            if (has_tail && tlen >= 1) {
                if (wv == 0) {
                    // cells v = W .. plen of the LAST row, sequentially (wave-uniform), with the aliased inputs (dp_wave.hpp)
                    const int h = tlen;
                    const int tch = ldsT[h - 1];
                    // B(tlen): the first wavefront's copy (lane-uniform for wave 0 unless it holds the tail owner)
                    const int bM = BM, bI = BI;   // every wavefront read B(tlen) at the start of the last row
                    int upM = tl[0], upD = tl[1];
                    int lastM = 0;
                    int tw_g = -1, tw_R = -1;         // lane word (direction bits) being assembled for the tail cells, and its canonical row
                    uint32_t tw[4] = {0u, 0u, 0u, 0u};
                    auto tw_flush = [&]() {
                        if (tw_g >= 0 && lane == 0) for (int d = 0; d < NQS; ++d) FLW[((size_t)tw_R * FS + tw_g) * NQS + d] = tw[d];
                    };
                    // (Round 6: ANY plen. The cell "above" tail cell v is flat index W tlen + v - W: a regular cell of the last row while v - W < W, one of the tail cells
                    //  themselves beyond that -- plen > 2 tlen, the pairs that until now were filled by ONE lane out of a table of their own (dp_literal_fill) -- so the
                    //  tail cells join the row's LDS image as they are made, and their direction bits sit at the canonical position of their flat index, row
                    //  tlen + v / W, column v mod W, where the walk looks for them.)
                    int Rt = tlen + 1, C = 0;         // canonical position of flat index W tlen + v
                    for (int v = W; v <= plen; ++v, ++C) {
                        if (C == W) { C = 0; ++Rt; }
                        int leftM, leftI, diagM;
                        if (v == W) { leftM = bM; leftI = bI; diagM = tl[2]; }
                        else {
                            leftM = rowM[v - W];
                            leftI = SWG ? (int)rowI[v - W] : 0;
                            diagM = (v - 1 == W) ? bM : (int)rowM[v - 1 - W];
                        }
                        const int pch = ldsP[v - 1];
                        int cM, cI, cDd;
                        if (SWG) {
                            cDd = min(upM + OE, upD + E);
                            cI = min(leftM + OE, leftI + E);
                            cM = min(diagM + ((pch == tch) ? MATCH : MISMATCH), min(cI, cDd));
                        } else {
                            cI = leftM + GI; cDd = upM + GD;
                            cM = min(diagM + ((pch == tch) ? 0 : MISMATCH), min(cI, cDd));
                        }
                        if (BT) {   // the direction bits of this cell: column C = v - W of row tlen + 1 (C = 0: the boundary array); NW: "not D", "not I" only
                            const uint32_t nD = cM != cDd ? 1u : 0u, nI = cM != cI ? 1u : 0u, xD = (SWG && upD + E < upM + OE) ? 1u : 0u, xI = (SWG && leftI + E < leftM + OE) ? 1u : 0u;
                            if (C == 0) { if (lane == 0) BF[Rt] = (unsigned char)(nD | (nI << 1) | (xD << 2) | (xI << 3) | ((!SWG || cM + O <= cDd) ? 0u : 16u)); }
                            else {
                                if (C >= 2) {   // this cell's "D extended" is kept at the cell on its left (column 1's: BF bit 4, set with column 0)
                                    const int t = (C - 2) - tw_g * K, j = t >> 1;
                                    tw[j >> RSH] |= xD << (8 * (t & 1) + 4 + (j & RM));   // (SWG only: xD is 0 for NW)
                                }
                                const int g = (C - 1) / K, t = (C - 1) - g * K, j = t >> 1;
                                if (g != tw_g || Rt != tw_R) {
                                    tw_flush();
                                    tw_g = g; tw_R = Rt; tw[0] = tw[1] = tw[2] = tw[3] = 0u;
                                }
                                tw[j >> RSH] |= (nD << (8 * (t & 1) + (j & RM))) | (nI << (8 * (2 + (t & 1)) + (j & RM))) | (xI << (8 * (2 + (t & 1)) + 4 + (j & RM)));
                            }
                        }
                        if (v < plen && lane == 0) { rowM[v] = (int16_t)cM; if (SWG) rowI[v] = (int16_t)cI; }   // (read back W cells on; same-wave LDS traffic is ordered)
                        upM = cM; upD = cDd;
                        lastM = cM;
                    }
                    if (BT) tw_flush();
                    if (lane == 0) tl[3] = lastM;
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                score = tl[3];
            } else {
                score = (plen >= 1 && tlen >= 1) ? (int)rowM[plen] : 0;
            }
            if (plen == 0 || tlen == 0) score = 0;
            __syncthreads();                          // everyone has read rowM / tl before the traceback reuses the area as its tile
            if (!BT) break;
            if (wv == 0) {
                bool left = false;
                if (!(a.dbg_flags & 1u))
                    left = dp_traceback_swg_bits<K, SWG>(a.p, plen, tlen, FS, FLW, BF, ldsP, ldsT, reinterpret_cast<uint32_t *>(rowM), rowbytes >= 12 * 1024 ? 256 : 64, ops, lane,
                                                         begin_offset, !full_bits, band_d0 - band_m + 2, band_d1 + band_m - 2);
                if (lane == 0) tl[5] = left ? 1 : 0;
            }
            __syncthreads();
            if (tl[5] == 0) break;                    // (else: the walk left the band -- once more, with every strip's bits)
            begin_offset = plen + tlen - 1;
        }

#ifdef AIM_STRIP_STAMPS
        __syncthreads();         // the traceback is done with the ops row: the stamps go there (the CIGAR of a diagnostic build is void)
        if (BT && lane == 0) {
            unsigned long long *dbg = reinterpret_cast<unsigned long long *>(ops + 64) + wv * 8;
            for (int i = 0; i < 8; ++i) dbg[i] = ssum[i];
        }
#endif
        if (tid == 0) {
            aim_result_t r;
            r.max_operations = plen + tlen;
            r.begin_offset = begin_offset;
            r.end_offset = end_offset;
            r.score = score;
            r.status = status;
            r.idx = rq.idx;
            store_result(a, pair, r);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Shapes: K cells per lane (16 / 20 / 24 / 32) x nw wavefronts covering READ_SIZE columns. A row costs every wavefront a fixed
// overhead (scan, mailboxes, boundary cell) plus K cells, and the CU's four SIMDs each take ceil(nw / 4) of the wavefronts:
// the shape with the cheapest BUSIEST SIMD wins (config 4, READ_SIZE 10 112: 16 x 10 -> 3 wavefronts on two SIMDs; 24 x 7 -> 2 on three of
// them and 1 on the fourth; round 4: 20 x 8 -> 2 on each).
// AIM_STRIP_K forces K for experiments.
struct StripShape { int k, nw, nwmax; };
inline bool dp_strip_shape(const aim_params_t &p, const Knobs &kn, StripShape *sh, uint32_t n_pairs = 0)
{
    const int rs = p.read_size;
    int best_k = 0, best_nw = 0;
    double best_cost = 0;
    for (int k : {16, 20, 24, 32}) {
        if (kn.strip_k > 0 && kn.strip_k != k) continue;
        const int nw = (rs + kWave * k - 1) / (kWave * k);
        if (nw > (k == 16 ? 12 : 8)) continue;
        // The busiest SIMD's wavefronts set a row's time: pe resident pairs of nw wavefronts put ceil(pe nw / 4) on it, and pe pairs leave per such time -- pe is what the
        // registers admit (dp_strip_plan: 16 wavefronts per CU at K = 16, 12 for SWG with CIGAR, 8 from K = 20 on), or the pairs per CU the batch has (config 4: one).
        // Until round 6 the rule knew "few" (<= four pairs per CU: ceil(nw / 4)) and "many" (nw), and added a quarter to K >= 24 for SWG with CIGAR (its spills): 1 024 pairs
        // at READ_SIZE 2 049 .. 4 096 took three or four wavefronts of 16 cells per lane where two of 20 / 24 / 32 run 1.5 - 2.5x faster (profiles/r06/strip_shape_sweep.txt;
        // SWG with CIGAR at READ_SIZE 2 952: 1 304 GCUPS at K = 24 x 2 against 518 at 16 x 3).
        const bool heavy = p.algo == AIM_ALGO_SWG && (p.flags & AIM_FLAG_BACKTRACE);
        const long cus = kn.cus > 0 ? (long)kn.cus : 256L;
        const long per_cu = std::max<long>(1, (k >= 20 ? 8 : (heavy ? 12 : 16)) / nw);
        const long pe = n_pairs == 0 ? per_cu : std::min<long>(per_cu, std::max<long>(1, ((long)n_pairs + cus - 1) / cus));
        double cost = (double)((pe * nw + 3) / 4) * (110 + 10.0 * k) / (double)pe;
        // (K = 16 with at most two wavefronts per SIMD: its rows take 1.5x those of K = 20 with the SAME wavefront count -- 256 pairs at READ_SIZE 2 952: 448 against 674 GCUPS, 3 688:
        //  564 against 846, strip_shape_sweep_few.txt; with four and more pairs per CU its higher residency makes up for it)
        if (k == 16 && pe * nw <= 8) cost *= 1.5;
        if (!best_k || cost < best_cost) { best_k = k; best_nw = nw; best_cost = cost; }
    }
    if (!best_k) return false;
    sh->k = best_k; sh->nw = best_nw;
    sh->nwmax = best_k == 16 ? (best_nw <= 4 ? 4 : (best_nw <= 8 ? 8 : 12)) : 8;
    return true;
}

inline bool dp_strip_supported(const aim_params_t &p, bool cell8, const Knobs &kn)
{
    if (cell8) return false;                                   // int8 SWG cells wrap by design: literal path of dp_wave.hpp
    StripShape sh;
    return dp_strip_shape(p, kn, &sh) && dp_strip_exact_ok(p, false);
}

inline bool dp_strip_plan(const aim_params_t &p, uint32_t n_pairs, uint64_t budget, const Knobs &kn, uint32_t *grid, uint32_t *block, size_t *lds,
                          uint64_t *scratch_per_wg, size_t *scratch_total, int *k_out, uint32_t *pool_tables)
{
    const uint64_t rs = (uint64_t)p.read_size;
    const bool swg = p.algo == AIM_ALGO_SWG;
    StripShape sh;
    if (!dp_strip_shape(p, kn, &sh, n_pairs)) return false;
    *k_out = sh.k;
    // Round 5: four direction bits per cell (NW uses two; NQS dwords per lane and row) + one byte per row.
    const uint64_t nq = swg ? (uint64_t)((sh.k / 2 + 3) / 4) : (uint64_t)((sh.k / 2 + 7) / 8), nqs = nq == 3 ? 4 : nq, fs = rs / (uint64_t)sh.k + 2;   // (DpBits)
    uint64_t per = (rs + 3) * fs * nqs * 4 + (rs + 3) + 64;
    if (!(p.flags & AIM_FLAG_BACKTRACE)) per = 256;
    per = (per + 255) & ~255ull;
    if (budget < per) return false;
    const int nw = sh.nw;
    *block = (uint32_t)(kWave * nw);
    const uint64_t seqcap = (rs + 79) & ~15ull, rowcap = (rs + 47) & ~7ull;
    const uint64_t rowbytes = std::max<uint64_t>(2 * rowcap * 2, 8 * 1024);
    *lds = (size_t)(2 * seqcap + rowbytes + 2 * kStripMaxWaves * kStripDepth * sizeof(StripMsg) + kStripMaxWaves * 4 + 16 * kStripDepth + 64);
    if (*lds > 160 * 1024) return false;
    // resident workgroups per CU: LDS, and the wavefronts the instantiation's register budget admits (64 * NWMAX threads per CU-quarter...)
    // (registers: K = 16: NW 82-114, SWG 117, SWG with CIGAR 160 VGPRs; K = 32: 130-240)
    const bool heavy = swg && (p.flags & AIM_FLAG_BACKTRACE);
    const uint32_t waves_per_cu = sh.k >= 20 ? 8u : (heavy ? 12u : 16u);
    const uint32_t per_cu = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(waves_per_cu / (uint32_t)nw, (uint64_t)lds_workgroups_per_cu(*lds)));
    uint32_t g = resident_grid(kn, per_cu);
    const uint32_t need = ((n_pairs + 7u) / 8u) * 8u;
    if (g > need) g = need < 8u ? 8u : need;
    while (g > 8 && per * g > budget) g -= 8;
    if (per * g > budget) return false;
    *grid = g;
    *scratch_per_wg = per;
    *scratch_total = (size_t)(per * g);
    *pool_tables = 0;   // (round 5's pool of literal-path tables: gone, see the kernel)
    return true;
}

// k: the plan's cells per lane (dp_strip_plan); the workgroup size gives the wavefront count
// Kernels are instantiated in ONE translation unit (tu_*.hip defines AIM_TU_DP_STRIP); every other includer sees the declaration only.
#ifdef AIM_TU_DP_STRIP
void dp_strip_launch(const aim_params_t &p, int k, uint32_t grid, uint32_t block, size_t lds, const KArgs &ka, hipStream_t s)
{
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    StripShape sh;
    sh.k = k; sh.nw = (int)(block / kWave);
    sh.nwmax = k == 16 ? (sh.nw <= 4 ? 4 : (sh.nw <= 8 ? 8 : 12)) : 8;
#define AIM_STRIP(KERNEL)                                                                             \
    do {                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(block), lds, s, ka);                             \
    } while (0)
#define AIM_STRIP_K(ALGOV, BTV)                                                                      \
    do {                                                                                             \
        if (sh.k == 32) AIM_STRIP((dp_strip_kernel<ALGOV, BTV, 32, 8>));                             \
        else if (sh.k == 24) AIM_STRIP((dp_strip_kernel<ALGOV, BTV, 24, 8>));                        \
        else if (sh.k == 20) AIM_STRIP((dp_strip_kernel<ALGOV, BTV, 20, 8>));                        \
        else if (sh.nwmax == 4) AIM_STRIP((dp_strip_kernel<ALGOV, BTV, 16, 4>));                     \
        else if (sh.nwmax == 8) AIM_STRIP((dp_strip_kernel<ALGOV, BTV, 16, 8>));                     \
        else AIM_STRIP((dp_strip_kernel<ALGOV, BTV, 16, 12>));                                       \
    } while (0)
    if (p.algo == AIM_ALGO_NW) { if (bt) AIM_STRIP_K(AIM_ALGO_NW, true); else AIM_STRIP_K(AIM_ALGO_NW, false); }
    else { if (bt) AIM_STRIP_K(AIM_ALGO_SWG, true); else AIM_STRIP_K(AIM_ALGO_SWG, false); }
#undef AIM_STRIP_K
#undef AIM_STRIP
}
#else
void dp_strip_launch(const aim_params_t &p, int k, uint32_t grid, uint32_t block, size_t lds, const KArgs &ka, hipStream_t s);
#endif

}  // namespace aim
