// dp_reg.hpp -- Needleman-Wunsch on short reads with the DP ROW IN REGISTERS: one pair per lane, two int16 cells per VGPR, no LDS
// in the cell loop (round 4).
//
// Same values as nw_compute (NW/DPU-WRAM/dpu/nw.c:109-153) for the pairs it takes; nw_lane_kernel (dp_lane.hpp: rows in LDS, the
// reference's cell order literally) keeps the rest. dp_lane.hpp needs an LDS read and an LDS write per cell because the reference
// indexes its table flat with stride W = tlen + 1 while v runs to plen (quirk N1): for plen > tlen cell (h, W) IS row h + 1's
// boundary cell and cells right of it read the CURRENT row's first cells -- positions that differ per lane, which a register file
// indexed by instruction cannot serve. But what N1 does depends on plen - tlen only:
//     plen <= tlen      no cell is aliased: plain NW, boundary cells h * GAP_I;
//     plen == tlen + 1  the row's LAST cell (h, W) is row h + 1's boundary cell B(h + 1) -- and is itself an ordinary cell: the cell
//                       "above" it that the flat index makes it read is B(h), i.e. its own previous value;
//     plen >= tlen + 2  tail cells that read the current row: left to nw_lane_kernel (to-do list; about a sixth of the pairs at e = 5 %).
// Rows are RIGHT-ALIGNED in the registers: column v of a lane lives at the static index i = v + s0, s0 = RSK - 1 - plen, so that
// every lane's last column -- the final score, and for plen == tlen + 1 the next row's boundary -- is the high half of the last
// register, whatever plen is. The row's start is then per lane (index s0, the boundary cell), and it needs no handling at all:
// indices left of s0 hold a large value INF (kept large by the recurrence itself), so at index s0 the substitution and the gap chain
// deliver INF and the insertion delivers R_{h-1}[0] + GAP_I = h * GAP_I -- the boundary cell, computed like any other. Only lanes with
// plen == tlen + 1 inject their boundary (one v_bfi per register of a 16-register window in which s0 must lie; lanes left of it: to-do).
//
// Per register (two cells): diagonal via v_perm over the previous row's registers, "characters differ" as v_pk_min_u16(p ^ t, 1) on
// 16-bit-expanded pattern characters, substitution / insertion / their minimum as packed int16 (v_pk_mad / v_pk_add / v_pk_min), and the
// in-row gap chain  m[v] = min(A[v], m[v-1] + GAP_D)  as two 16-bit steps: 10 VALU instructions, no LDS, no branch. nw_reg_supported()
// (costs small enough that INF stays above every cell and below int16's end) is the plan's precondition.
//
// BACKTRACE: the table is TWO BITS per cell. nw_traceback (nw.c:67-107) asks of a cell c, in this order: c == left + GAP_D ('D')?
// c == up + GAP_I ('I')? else 'X' / 'M' by c == diag + MISMATCH. For the pairs this kernel takes every cell the walk compares with
// still holds the value the fill read (aliasing only touches boundary cells, and those are cells of the row here), so both answers
// are known when the cell is computed: "not D" = the gap chain lost strictly (sign of A - chain), "not I" = the insertion lost
// strictly (sign of sub - ins), and 'X' vs 'M' is the character comparison, which the walk redoes from the sequences. 116 cells of
// a row are 8 dwords per lane instead of 232 bytes: with int16 cells the fill was bound by its own table stream (28.6 GB per 1 M pairs
// at l = 100 = the time of nw_lane_kernel), and the walk follows ONE dependent load per step instead of three.
#pragma once

#include <type_traits>

#include "aim_device.hpp"
#include "dp_lane.hpp"
#include "dp_strip.hpp"   // dps2, pk_ne01, opaque
#include "wfa_lane.hpp"   // LANE_TODO_*

namespace aim {

typedef uint32_t aim_u32x4_u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte piece of a sequence row (rows start on 8-byte boundaries)
// dword q & 3 of a row's 16-byte unit of direction bits: bitwise selects under v_bfe_i32 masks (a `q & 1 ? a : b` tree over the elements of a vector is
// taken for a dynamic index and the vector is put into scratch)
__device__ __forceinline__ uint32_t band_pick(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3, int q)
{
    uint32_t m1 = (uint32_t)__builtin_amdgcn_sbfe(q, 0, 1), m2 = (uint32_t)__builtin_amdgcn_sbfe(q, 1, 1);
    asm volatile("" : "+v"(m1), "+v"(m2));
    const uint32_t lo = (w1 & m1) | (w0 & ~m1), hi = (w3 & m1) | (w2 & ~m1);
    return (hi & m2) | (lo & ~m2);
}

// The walk's ops are staged in LDS one contiguous row per lane: stride 2 * READ_SIZE rounded to an ODD multiple of 16 bytes (lanes spread over the banks, 16-byte pieces
// stay aligned), 16 bytes of headroom in front (the cursor may come to rest at index -1).
__host__ __device__ inline int reg_ops_stride(int read_size) { const int s = 2 * read_size; return ((s >> 4) & 1) ? s : s + 16; }

constexpr int kRegWin = 16;        // registers (32 indices) in which a row may start
constexpr int kNwTail = 8;         // tail cells of the LAST row a pair may have beyond (tlen, W): plen <= tlen + 1 + kNwTail (round 5)
constexpr int kRegInf = 16000;     // the value left of a row's start

// Shapes: NPK registers per row = 2 * NPK indices > READ_SIZE
// Registers of a row. The launchers' READ_SIZE rule (run-nw-pim-wram.py: ceil((l + l*e + 7) / 8) * 8) leaves pattern and text at READ_SIZE - 7 characters at
// most: a row of 2 * NPK >= READ_SIZE - 6 indices holds every pair they produce; a longer pattern or text (legal: up to READ_SIZE) goes to the to-do list.
// (Until late in round 4: 42 / 58 / 66 = READ_SIZE + 4 indices, 7-10 % of them never used.)
// (Round 5, ADVICE r04: classes for READ_SIZE <= 48 / 64 / 96 too -- a row may start in its first 32 indices only, so at READ_SIZE 48 or 88 no pair qualified for the
// class above it and the whole batch went through the to-do list.)
__host__ __device__ inline int nw_reg_npk(int read_size, bool bt = false)
{
    (void)bt;   // READ_SIZE 144 / 160 / 176 (l = 150): 70 / 78 / 86 registers per row with the PATTERN row in LDS (as swg_reg_kernel's large classes) and 12 dwords of direction bits
    return read_size <= 48 ? 22 : read_size <= 64 ? 30 : read_size <= 80 ? 38 : read_size <= 96 ? 46 : read_size <= 112 ? 54 : read_size <= 128 ? 62 :
           read_size <= 144 ? 70 : read_size <= 160 ? 78 : read_size <= 176 ? 86 : 0;
}

inline bool nw_reg_supported(const aim_params_t &p)
{
    if (p.algo != AIM_ALGO_NW || nw_reg_npk(p.read_size, (p.flags & AIM_FLAG_BACKTRACE) != 0) == 0 || p.read_size < 40) return false;
    if (p.gap_i <= 0 || p.gap_d <= 0 || p.mismatch <= 0) return false;
    const long g = std::max(p.mismatch, std::max(p.gap_i, p.gap_d));
    return (2L * p.read_size + 8) * g < 8000;   // every cell < 8000 < INF; INF + READ_SIZE * g < 24000: nothing wraps, nothing left of a row's start wins
}

__host__ __device__ inline size_t nw_reg_lds_bytes(const aim_params_t &p)   // text image / ops staging + the pair queue (512 B)
{
    const bool bt = (p.flags & AIM_FLAG_BACKTRACE) != 0;
    const int npk = nw_reg_npk(p.read_size, bt);
    const size_t t = (size_t)((2 * npk + 3) / 4) * kWave * 4 * (npk > 62 ? 2 : 1), o = (size_t)reg_ops_stride(p.read_size) * kWave + 32;   // (npk > 62: text image + pattern image)
    return ((bt && o > t) ? o : t) + 512;
}

// nw_traceback over the band of direction bits nw_reg_kernel left for ONE batch of 64 pairs (one pair per lane). (A function of its own since round 6's experiment with
// the walks as a KERNEL of their own behind the fill -- every batch a slab of its own, four walking wavefronts per SIMD instead of two filling ones: the walk kernel alone took
// 0.98 ms per 1 Mi pairs against the ~0.4 ms the walks add inside the fill kernel, where the other wavefront of the SIMD fills meanwhile. Refuted and removed; NOTES R6.2.)
template <int NPK>
__device__ __forceinline__ void nw_reg_walk(const KArgs &a, char *smem, const int lane, uint32_t *tbw, const bool mine, const uint32_t pair, const int plen, const int tlen,
                                            const uint32_t idx, const int s0, const int bandc, const uint32_t tailbits, const int score, uint32_t *todo)
{
    constexpr int NDQ = (NPK + 7) / 8;
    typedef uint32_t aim_u32x4 __attribute__((ext_vector_type(4)));
    const int rs = a.p.read_size, W = tlen + 1;
    const uint32_t *gP = reinterpret_cast<const uint32_t *>(a.patterns + (uint64_t)pair * rs);
    const uint32_t *gT = reinterpret_cast<const uint32_t *>(a.texts + (uint64_t)pair * rs);
    auto band_q0 = [&](int h) { return min(max((bandc + h - 24) >> 4, 0), NDQ > 4 ? NDQ - 4 : 0); };   // first dword of row h's window (as the fill)
#define TBU(h) (reinterpret_cast<aim_u32x4 *>(tbw) + ((size_t)(h) * kWave + lane))
#define TBU2(h) (reinterpret_cast<aim_u32x4 *>(tbw) + ((size_t)((h) + rs + 2) * kWave + lane))
    {
        // nw_traceback (nw.c:67-107) over the direction bits (see the header); 'X' / 'M' from the sequences. Ops staged in LDS (the text image is dead by
        // now), copied out in 16-byte pieces. Round 6: the rows' units are fetched EIGHT ROWS AT A TIME (independent loads, one round trip) and the walk runs
        // through them row by row -- the walk never returns to a row it has left, so the unrolled row index is static. (Until round 5: one dependent HBM load per
        // step, ~100 round trips of ~1 us per pair: 0.97 of the kernel's 3.6 ms at l = 100.)
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        bool lost = false;                                     // the walk left the band: the pair goes to the to-do list
        if (mine && !(a.dbg_flags & 5u)) {
            int begin_offset = plen + tlen - 1;
            const int end_offset = plen + tlen;
            char *ops_g = a.ops + (uint64_t)pair * 2 * rs;
            const unsigned char *pb = reinterpret_cast<const unsigned char *>(gP), *tbytes = reinterpret_cast<const unsigned char *>(gT);
            unsigned char *ops_l = reinterpret_cast<unsigned char *>(smem) + lane * reg_ops_stride(rs) + 16;   // this lane's row (reg_ops_stride)
#define OPS(i) ops_l[(i)]
            int sentinel = end_offset - 1;
            int h = tlen, v = plen;
            // One step of the walk on a cell's code (bit 0 "not D", bit 1 "not I"), WITHOUT branches: the walk is a chain of ~130 dependent steps per pair on all 64
            // lanes, and as three divergent branches per step it cost as many instructions as a seventh of the fill.
#define NW_STEP(code_, pch_, tch_) do { const uint32_t cd_ = (code_), pq_ = (pch_), tq_ = (tch_);   /* (the characters are READ whatever the code says: as operands of the */ \
                const uint32_t xm_ = pq_ != tq_ ? (uint32_t)'X' : (uint32_t)'M';                       /*  selects they would be fetched lazily, i.e. behind branches)          */ \
                const uint32_t op_ = !(cd_ & 1u) ? (uint32_t)'D' : (!(cd_ & 2u) ? (uint32_t)'I' : xm_);                                            \
                OPS(sentinel) = (unsigned char)op_;                                                                                               \
                --sentinel; h -= (int)(cd_ & 1u); v -= (int)(((cd_ >> 1) | ~cd_) & 1u); } while (0)
            // The walk's characters without a dependent global load per step (two per 'M' / 'X' step until round 6: what was left of the walk's time once the
            // direction bits came eight rows at a time): the PATTERN row is staged in LDS, in the ops row itself (row byte i at OPS(i)) -- the cursor
            // never reaches a pattern byte that is still to be read (cursor - (v - 1) = h >= 1 at every step) --, the TEXT characters of a batch's eight rows
            // come with the batch (row h needs t[h - 1] only).
            for (int b = 0; 16 * b < rs; ++b) {
                const aim_u32x4_u w = __builtin_nontemporal_load(reinterpret_cast<const aim_u32x4_u *>(pb) + b);
                *reinterpret_cast<uint4 *>(&OPS(16 * b)) = make_uint4(w[0], w[1], w[2], w[3]);
            }
            // A pair with tail cells starts in the last row's tail: the cell the reference reads at flat index W h + v with v > W is the tail cell v - W (h == tlen:
            // its bits are in `tailbits`, its characters its own). Above the last row such an index is cell (h + 1, v - W) of the table: one of the next row's first
            // eight columns (TBU2).
            constexpr int NB = 8;                             // rows per batch
            aim_u32x4 u[NB];                                  // the units of rows h0 .. h0 - NB + 1 ...
            uint32_t tc[NB];                                  // ... and those rows' text characters
            int h0 = -1;
            while (h > 0 && v > 0 && !lost) {
                while (h > 0 && v > W) {                      // (pairs with tail cells, until the walk is back inside the table's own columns: one dependent load per step)
                    if (h == tlen) NW_STEP((tailbits >> (2 * (v - W))) & 3u, OPS(v - 1), tbytes[h - 1]);
                    else {
                        const int vv = v - W, i = vv + s0;    // cell (h + 1, v - W); its characters are that cell's own
                        const aim_u32x4 w = __builtin_nontemporal_load(TBU2(h + 1));
                        const uint32_t word = band_pick(w[0], w[1], w[2], 0u, i >> 4) >> (8 * (i & 1) + ((i >> 1) & 7));
                        NW_STEP((word & 1u) | ((word >> 15) & 2u), OPS(vv - 1), tbytes[h]);
                    }
                }
                if (!(h > 0 && v > 0)) break;
                if (h != h0) {                                // the first batch: rows h .. h - NB + 1
                    h0 = h;
#pragma unroll
                    for (int r = 0; r < NB; ++r) { u[r] = __builtin_nontemporal_load(TBU(max(h0 - r, 1))); tc[r] = tbytes[max(h0 - r, 1) - 1]; }
                }
                aim_u32x4 un[NB];                             // the NEXT batch is asked for before this one is walked: its round trip hides behind the steps
                uint32_t tn[NB];
#pragma unroll
                for (int r = 0; r < NB; ++r) { un[r] = __builtin_nontemporal_load(TBU(max(h0 - NB - r, 1))); tn[r] = tbytes[max(h0 - NB - r, 1) - 1]; }
#pragma unroll
                for (int r = 0; r < NB; ++r) {
                    const int qw = band_q0(h0 - r);           // the row's window (per lane: the lanes' rows differ)
                    while (h == h0 - r && h > 0 && v > 0 && !lost) {     // (v <= W from here on)
                        const int i = v + s0, q = i >> 4;
                        lost = (uint32_t)(q - qw) >= 4u;                 // (outside the window: this step's result is void, the pair goes to the to-do list)
                        const uint32_t word = band_pick(u[r][0], u[r][1], u[r][2], u[r][3], q) >> (8 * (i & 1) + ((i >> 1) & 7));
                        NW_STEP((word & 1u) | ((word >> 15) & 2u), OPS(v - 1), tc[r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < NB; ++r) { u[r] = un[r]; tc[r] = tn[r]; }
                h0 -= NB;
            }
            if (!lost) {
                while (h > 0) { OPS(sentinel) = 'I'; --sentinel; --h; }
                while (v > 0) { OPS(sentinel) = 'D'; --sentinel; --v; }
                begin_offset = sentinel + 1;
                {
                    const uint4 *src = reinterpret_cast<const uint4 *>(ops_l);
                    uint4 *dst = reinterpret_cast<uint4 *>(ops_g);
                    for (int q = begin_offset >> 4; q <= (end_offset - 1) >> 4; ++q) dst[q] = src[q];
                }
                aim_result_t res;
                res.max_operations = plen + tlen;
                res.begin_offset = begin_offset;
                res.end_offset = end_offset;
                res.score = score;
                res.status = AIM_PAIR_OK;
                res.idx = idx;
                store_result(a, pair, res);
            }
#undef NW_STEP
#undef OPS
        }
        {   // pairs whose walk left the band: nw_lane_kernel aligns them again behind this kernel (one atomic per wavefront that holds one)
            const unsigned long long lm = __ballot(lost);
            if (lm) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&todo[LANE_TODO_COUNT], (uint32_t)__builtin_popcountll(lm));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (lost) todo[LANE_TODO_LIST + base + (uint32_t)__builtin_popcountll(lm & ((1ull << lane) - 1ull))] = pair;
            }
        }
    }
#undef TBU
#undef TBU2
}

template <int NPK, bool BT>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void nw_reg_kernel(KArgs a)   // (two wavefronts per SIMD: vector + accumulation registers <= 256)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    constexpr int RSK = 2 * NPK;          // indices of a row of registers
    constexpr int NWD = (RSK + 3) / 4;    // dwords of a sequence row that cover them
    constexpr bool PL = NPK > 62;             // the (shifted) pattern row lives in LDS as bytes, not in NPK more registers (READ_SIZE 144 .. 176)
    constexpr int NDQ = (NPK + 7) / 8;        // dwords of direction bits a row has per lane (8 registers per dword); four of them are kept (the band, below)
    constexpr bool kBandBits = false;         // direction bits made for the band's dwords only (see do_row)
    constexpr bool TAILS = true;              // last-row tail cells in this kernel (with CIGAR too: their direction bits stay in a register, see the walk)
    const int lane = threadIdx.x;
    const int rs = a.p.read_size, rsw = rs >> 2;
    const int GAP_D = a.p.gap_d, GAP_I = a.p.gap_i, MISMATCH = a.p.mismatch;
    uint32_t *todo = const_cast<uint32_t *>(a.todo);                     // OUT: pairs left to nw_lane_kernel ({count @0, pair ids @16..})
    int16_t *tb = BT ? reinterpret_cast<int16_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave) : nullptr;
    const uint32_t n_groups = (a.n_pairs + kWave - 1) / kWave;
    uint32_t ones = 0x00010001u;
    opaque(ones);
    // TILTED coordinates (round 4): the registers hold T(h, v) = R(h, v) - GAP_I * h - GAP_D * v. The gap moves then cost nothing -- "up" is the old
    // register itself, the chain along the row is a running minimum (two instructions per register instead of four) -- and the diagonal move
    // carries the constant: mismatch - GAP_I - GAP_D or -GAP_I - GAP_D. All three candidates of a cell are shifted by the same amount, so every
    // comparison (the minima, the direction bits and their tie order) is unchanged; the score is un-tilted once at the end. |T| <= GAP_I * tlen +
    // GAP_D * plen < 8 000 (nw_reg_supported), INF = 16 000 stays out of reach.
    const dps2 x2 = dps_splat(MISMATCH), c2 = dps_splat(-(GAP_I + GAP_D));
    // direction bits: a row has NDQ dwords per lane (dword q: registers 8q .. 8q + 7 = indices 16q .. 16q + 15; register 8q + r at bit r of every byte: byte 0 / 1 =
    // "not D" of its low / high half, byte 2 / 3 = "not I"). Round 6: only a BAND of four consecutive dwords around the diagonal is kept -- ONE 16-byte unit per row
    // and lane, lane-interleaved, dword q in slot q & 3. Rows are right-aligned, so the path's index at row h is RSK - 1 - (tlen - h) up to the pair's own drift:
    // wave-uniform up to the spread of the lanes' lengths. The window q0(h) .. q0(h) + 3 (64 indices, at least +-24 about the wavefront's centre line) is a function
    // of the row alone; a walk that leaves it sends its pair to the to-do list (nw_lane_kernel), like cfg4's band in dp_strip.hpp. (Until round 5: all 8 / 12 dwords,
    // 3.5 GB of stores per 1 Mi pairs at l = 100 and one dependent HBM round trip per step of the walk -- a third of the kernel's time, profiles/r06/reg_cigar_split.txt.)
    uint32_t *tbw = reinterpret_cast<uint32_t *>(tb);
    typedef uint32_t aim_u32x4 __attribute__((ext_vector_type(4)));
#define TBU(h) (reinterpret_cast<aim_u32x4 *>(tbw) + ((size_t)(h) * kWave + lane))
    // ... and, for pairs with tail cells (plen >= tlen + 2) only, a second unit per row: dwords 0 .. 2, the row's FIRST columns (s0 + 8 <= 39). Their walk starts in the
    // last row's tail, and a step up or diagonally from a tail cell leaves it at a flat index W h + v with v > W -- which is cell (h + 1, v - W) of the table, one of
    // the next row's first eight columns (quirk N1; half of all tail pairs at e = 5 % pass through such cells).
#define TBU2(h) (reinterpret_cast<aim_u32x4 *>(tbw) + ((size_t)((h) + rs + 2) * kWave + lane))

    // Pairs are taken through a small LDS queue: groups of 64 consecutive pairs are classified (this kernel's / to-do list) and the
    // kernel's own are queued; the row loop runs on 64 QUEUED pairs at a time. (Without it the to-do pairs' lanes idle through the whole
    // table: a sixth of the lanes at e = 5 % -- the kernel was slower than nw_lane_kernel there.)
    uint32_t *queue = reinterpret_cast<uint32_t *>(smem + nw_reg_lds_bytes(a.p) - 512);   // 128 entries behind the text / ops area
    uint32_t qn = 0, it = 0;                                 // wave-uniform
    bool more = true;
    for (;;) {
        while (qn < (uint32_t)kWave && more) {
            uint32_t grp;
            more = xcd_unit(n_groups, it, &grp);
            if (!more) break;
            ++it;
            const uint32_t cand = grp * kWave + lane;
            const bool act = cand < a.n_pairs;
            aim_request_t rc;
            rc.pattern_len = rc.text_len = 0; rc.padding = 0; rc.idx = 0;
            if (act) rc = load_request(a, cand);
            // this kernel's pairs: plen <= tlen + 1, or (round 5; with CIGAR too) at most kNwTail tail cells beyond (h, W) -- those of the rows before the last are
            // overwritten by the next row before anything else reads them, so the rows run over the columns 0 .. W like a plen == tlen + 1 pair's and the LAST
            // row's tail cells are computed once, when that row is done (tail_cells below) -- and a row start inside the window
            const int cpe = rc.pattern_len > rc.text_len ? rc.text_len + 1 : rc.pattern_len;   // columns the rows run over
            const bool take = act && rc.pattern_len >= 1 && rc.text_len >= 1 && rc.pattern_len <= rc.text_len + 1 + ((TAILS && !(a.dbg_flags & 2u)) ? kNwTail : 0) && cpe >= RSK - 2 * kRegWin &&
                              cpe <= RSK - 1 && rc.pattern_len <= rs && rc.text_len <= 4 * NWD;   // ... and both sequences inside the row / the staged text image
            const unsigned long long rest = __ballot(act && !take), mask_below = (1ull << lane) - 1ull;
            if (rest) {   // everything else: the to-do list of nw_lane_kernel (one atomic per wavefront)
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&todo[LANE_TODO_COUNT], (uint32_t)__builtin_popcountll(rest));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (act && !take) todo[LANE_TODO_LIST + base + (uint32_t)__builtin_popcountll(rest & mask_below)] = cand;
            }
            const unsigned long long tk = __ballot(take);
            if (take) queue[qn + (uint32_t)__builtin_popcountll(tk & mask_below)] = cand;
            qn += (uint32_t)__builtin_popcountll(tk);
        }
        if (qn == 0) break;
        const uint32_t ntake = qn < (uint32_t)kWave ? qn : (uint32_t)kWave;
        const bool mine = (uint32_t)lane < ntake;
        const uint32_t pair = mine ? queue[lane] : 0u;
        const uint32_t moved = (ntake + lane < qn) ? queue[ntake + lane] : 0u;   // the queue's remainder moves to its front (same-wave LDS traffic is ordered)
        if (ntake + lane < qn) queue[lane] = moved;
        qn -= ntake;
        aim_request_t rq;
        rq.pattern_len = rq.text_len = 0; rq.padding = 0; rq.idx = 0;
        if (mine) rq = load_request(a, pair);
        const int plen = rq.pattern_len, tlen = rq.text_len;
        const int W = tlen + 1;
        const bool isW = mine && plen >= W;                  // cell (h, W) is the next row's boundary cell
        const int pe = plen >= W ? W : plen;                 // the rows' last column
        const int ntail = mine ? plen - pe : 0;              // tail cells of the last row
        const int s0 = mine ? RSK - 1 - pe : 0;              // index of column 0 (0 .. 31)
        // BACKTRACE: the band's centre line (index at row 0, wave-uniform): midway between the lanes' extreme corner diagonals -- a pair's path runs between the
        // diagonal through (0, 0), index s0 + h, and the one through (tlen, pe), index RSK - 1 - tlen + h
        int bandc = 0;
        if (BT) {
            const int dlo = mine ? min(s0, RSK - 1 - tlen) : 0x7fffffff, dhi = mine ? max(s0, RSK - 1 - tlen) : -0x7fffffff;
            bandc = (wave_min_i32(dlo) - wave_min_i32(-dhi)) >> 1;
        }
        auto band_q0 = [&](int h) { return min(max((bandc + h - 24) >> 4, 0), NDQ > 4 ? NDQ - 4 : 0); };   // first dword of row h's window
        const bool wtail = BT && __ballot(ntail > 0) != 0ull;   // a pair of this wavefront has tail cells
        const uint32_t *gP = reinterpret_cast<const uint32_t *>(a.patterns + (uint64_t)pair * rs);   // (idle lanes: pair 0's rows, read and ignored)
        const uint32_t *gT = reinterpret_cast<const uint32_t *>(a.texts + (uint64_t)pair * rs);
        // the text row goes to LDS, transposed [dword][lane] (one conflict-free ds_read per ROW of the table), the pattern row into
        // registers as 16-bit fields, shifted so that character v - 1 sits at index v + s0 (column v)
        uint32_t *ldsT = reinterpret_cast<uint32_t *>(smem);
        uint32_t pc[NPK];
        __syncthreads();                                      // (single wavefront: the previous group's traceback is done with this area)
#pragma unroll
        for (int i = 0; i < NWD; ++i) {
            const uint32_t pw = i < rsw ? gP[i] : 0u;
            ldsT[i * kWave + lane] = i < rsw ? gT[i] : 0u;
            if (2 * i < NPK) pc[2 * i] = __builtin_amdgcn_perm(0u, pw, 0x0c010c00u);           // bytes 0, 1 -> 16-bit fields (indices 4i, 4i + 1)
            if (2 * i + 1 < NPK) pc[2 * i + 1] = __builtin_amdgcn_perm(0u, pw, 0x0c030c02u);   // bytes 2, 3
        }
        {   // shift right by d = s0 + 1 fields (1 .. 32): whole registers by 16 / 8 / 4 / 2 / 1, then one field
            const int d = s0 + 1;
#pragma unroll
            for (int k = 16; k >= 1; k >>= 1) {
                const bool on = ((d >> 1) & k) != 0;
#pragma unroll
                for (int j = NPK - 1; j >= 0; --j) { const uint32_t from = j - k >= 0 ? pc[j - k] : 0u; pc[j] = on ? from : pc[j]; }
            }
            if (d & 1) {
#pragma unroll
                for (int j = NPK - 1; j >= 0; --j) pc[j] = __builtin_amdgcn_alignbit(pc[j], j ? pc[j - 1] : 0u, 16);
            }
        }
        uint32_t *ldsP = ldsT + NWD * kWave;                  // PL: the shifted pattern row as bytes (field i of the row = byte i), transposed like the text's
        if (PL) {
#pragma unroll
            for (int i = 0; i < (NPK + 1) / 2; ++i)
                ldsP[i * kWave + lane] = __builtin_amdgcn_perm(2 * i + 1 < NPK ? pc[2 * i + 1] : 0u, pc[2 * i], 0x06040200u);   // the low bytes of four 16-bit fields
        }
        // row 0: column v = v * GAP_D at index v + s0, INF left of it
        uint32_t Mp[NPK];
#pragma unroll
        for (int j = 0; j < NPK; ++j) {
            const int v0 = 2 * j - s0, v1 = v0 + 1;
            Mp[j] = (uint32_t)(v0 >= 0 ? 0 : kRegInf) | ((uint32_t)(v1 >= 0 ? 0 : kRegInf) << 16);   // (tilted: R(0, v) = v * GAP_D is 0)
        }
        // plen == tlen + 1: flat cell (0, W) IS cell (1, 0), and the column initialisation wrote 1 * GAP_I there after the row's (nw.c:116-128)
        if (isW) Mp[NPK - 1] = (Mp[NPK - 1] & 0xffffu) | ((uint32_t)(uint16_t)(GAP_I - GAP_D * W) << 16);
        // isW lanes: B(h) = raw cell (h - 1, W), presented as cell (h, 0): un-tilt at (h - 1, W), tilt at (h, 0) = + GAP_D * W - GAP_I
        const uint32_t kW = (uint32_t)(uint16_t)(GAP_D * W - GAP_I);
        // isW lanes: where the boundary is injected (index s0 as an all-ones field of the window's registers)
        uint32_t inj[kRegWin];
#pragma unroll
        for (int j = 0; j < kRegWin; ++j) {
            inj[j] = isW ? ((2 * j == s0 ? 0x0000ffffu : 0u) | (2 * j + 1 == s0 ? 0xffff0000u : 0u)) : 0u;
            opaque(inj[j]);
        }
        int score = 0;
        uint32_t tailbits = 0u;                                // BACKTRACE: the direction bits of the last row's tail cells, two per cell (cell c at bits 2c, 2c + 1)
        const int hmax = -wave_min_i32(mine ? -tlen : 0);
        // one row: `src` (row h - 1) -> `dst` (row h). The row loop alternates between two register arrays: updated in place, the old value
        // of register j - 1 (the next cell's diagonal input) had to be copied aside before every overwrite -- one v_mov per register and row.
        // Registers entirely left of EVERY lane's row start hold INF and stay INF: the row starts at the first register that holds a
        // lane's column 0 (J0: a compile-time lower bound of it, chosen per wavefront below).
        uint32_t Mq[NPK];
#pragma unroll
        for (int j = 0; j < NPK; ++j) Mq[j] = j < kRegWin ? ((uint32_t)kRegInf * 0x00010001u) : 0u;
        auto do_row = [&](auto j0_tag, int h, uint32_t (&src)[NPK], uint32_t (&dst)[NPK]) __attribute__((always_inline)) {
            constexpr int J0 = decltype(j0_tag)::value;
            const uint32_t tword = ldsT[((h - 1) >> 2) * kWave + lane];
            const uint32_t tch2 = ((tword >> (((h - 1) & 3) * 8)) & 0xffu) * 0x00010001u;
            // isW lanes: B(h) = cell (h - 1, W) = the previous row's last cell (row 1: the row-init value GAP_I, set above), re-tilted for (h, 0)
            const uint32_t binj = (((src[NPK - 1] >> 16) + kW) & 0xffffu) * 0x00010001u;
            uint32_t rprev = (uint32_t)kRegInf << 16;         // the register left of this one: m[index - 1] in its HIGH half
            uint32_t dcur = 0u;                               // BACKTRACE: the direction bits of the current dword (8 registers)
            aim_u32x4 unit = {0u, 0u, 0u, 0u};                // ... and the row's window of four dwords (dword q in slot q & 3)
            const int q0 = BT ? band_q0(h) : 0;               // wave-uniform
            aim_u32x4 unit2 = {0u, 0u, 0u, 0u};               // dwords 0 .. 2 (pairs with tail cells)
            uint32_t pword = 0u;                              // PL: the pattern dword of the current two registers
            uint32_t oldprev = (uint32_t)kRegInf << 16;
            // (BACKTRACE updates the row in place: the NEXT register's diagonal is taken before this register is overwritten -- as a plain read-later the compiler kept a
            //  v_mov copy per register and row; the score-only variant alternates between two arrays and needs neither)
            uint32_t dnext = __builtin_amdgcn_alignbit(src[J0], oldprev, 16);
            // one register (two cells). BITS: with its direction bits (BACKTRACE, registers of the dwords the row keeps: one wave-uniform branch per dword of 8 registers --
            // round 6; until then every register's bits were made: 0.65 of nw l = 100's 3.6 ms)
            auto reg = [&](int j, auto bits_tag) __attribute__((always_inline)) {
                constexpr bool BITS = decltype(bits_tag)::value;
                const uint32_t oldj = src[j];
                const dps2 diag = dps_from(BT ? dnext : __builtin_amdgcn_alignbit(oldj, oldprev, 16));   // {R_{h-1}[2j - 1], R_{h-1}[2j]}
                if (BT) {
                    dnext = j + 1 < NPK ? __builtin_amdgcn_alignbit(src[j + 1], oldj, 16) : 0u;
                    opaque(dnext);
                }
                uint32_t pcj;
                if (PL) {
                    if ((j & 1) == 0 || j == J0) pword = ldsP[(j >> 1) * kWave + lane];
                    pcj = __builtin_amdgcn_perm(0u, pword, (j & 1) ? 0x0c030c02u : 0x0c010c00u);
                } else pcj = pc[j];
                const dps2 f = dps_from(pk_ne01(pcj, tch2, ones));
                const dps2 sub = (f * x2 + c2) + diag;
                const dps2 ins = dps_from(oldj);                 // (tilted: the move from the row above costs nothing)
                dps2 A = dps_min(sub, ins);
                if (j < kRegWin) A = dps_from((binj & inj[j]) | (dps_bits(A) & ~inj[j]));
                // the gap chain, two cells (tilted: a running minimum): lo = min(A.lo, m[index - 1]), hi = min(A.hi, lo), result packed {hi, lo}. SDWA by
                // hand: the second minimum writes the HIGH half of the register that holds lo (dst_unused:UNUSED_PRESERVE) and the next register's
                // first minimum reads it from there -- as C the compiler keeps lo and hi in two registers and packs them with a v_perm.
                uint32_t t1, res;   // (ONE asm statement: the compiler pads every asm statement with an s_nop)
                if (!BITS) {
                    if (BT) asm volatile("v_min_i16_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                        "v_min_i16_sdwa %0, %2, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0"
                        : "=&v"(res) : "v"(rprev), "v"(dps_bits(A)));
                    else asm("v_min_i16_sdwa %0, %2, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                        "v_min_i16_sdwa %0, %2, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0"
                        : "=&v"(res) : "v"(rprev), "v"(dps_bits(A)));
                } else {
                    // the same chain with both gap candidates kept, packed {m[2j], m[2j - 1]} = the cells' left neighbours (t1), for the direction bits
                    uint32_t sI = dps_bits(sub - ins);   // insertion lost strictly (taken BEFORE the chain: the old row's register is dead from here on and takes the new value)
                    opaque(sI);
                    asm volatile("v_min_i16_sdwa %0, %3, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_1\n\t"
                        "v_alignbit_b32 %1, %0, %2, 16\n\t"
                        "v_min_i16_sdwa %0, %3, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1 src1_sel:WORD_0"
                        : "=&v"(res), "=&v"(t1) : "v"(rprev), "v"(dps_bits(A)));
                    const uint32_t sD = dps_bits(A - dps_from(t1));   // sign bit: chain lost strictly
                    // Round 5: the four sign bits of a register as four BYTES (v_perm_b32, selectors 8 .. 11 replicate a source's sign bits), bit r of each byte kept
                    // for register 8q + r: four instructions per register where shifting, masking and three levels of v_pk_mad_u16 were eight
                    // (dword q: byte 0 / 1 = "not D" of the low / high halves, byte 2 / 3 = "not I").
                    uint32_t cj = __builtin_amdgcn_perm(sI, sD, 0x0b0a0908u);
                    opaque(cj);   // (volatile, like the chain above: the bits are made HERE -- left to itself the compiler makes all 58 registers' bits at the end of the row, with every register's sub / ins / A / chain live until then: ~400 VGPRs)
                    dcur |= cj & (0x01010101u << (j & 7));
                }
                dst[j] = res;
                rprev = res;
                oldprev = oldj;
            };
#pragma unroll
            for (int k = J0 >> 3; k < NDQ; ++k) {             // dword k: registers 8k .. 8k + 7
                // (Bits are made for EVERY dword and only the kept ones leave: making them for the kept dwords alone -- one wave-uniform branch per dword, as swg_reg_kernel
                //  does for its four bits per cell -- measured SLOWER here (2.50 against 2.38 ms per 1 Mi pairs without stores and walk): the two bits are 6 of a register's
                //  17 instructions, and 14 taken branches per row cost two wavefronts per SIMD more than 22 x 6 instructions. kBandBits = true builds that variant.)
                const bool inw = BT && (!kBandBits || (k >= q0 && k < q0 + 4) || (k < 3 && wtail));
                if (inw) {
#pragma unroll
                    for (int j = (8 * k > J0 ? 8 * k : J0); j < 8 * k + 8 && j < NPK; ++j) reg(j, std::true_type{});
                    if (k >= q0 && k < q0 + 4) unit[k & 3] = dcur;
                    if (k < 3) unit2[k] = dcur;
                    dcur = 0u;
                } else {
#pragma unroll
                    for (int j = (8 * k > J0 ? 8 * k : J0); j < 8 * k + 8 && j < NPK; ++j) reg(j, std::false_type{});
                }
            }
            if (h == tlen) score = (int)(int16_t)(dst[NPK - 1] >> 16) + GAP_I * tlen + GAP_D * pe;   // R_tlen[pe], un-tilted
            if (BT && mine && h <= tlen && !(a.dbg_flags & 4u)) {
                __builtin_nontemporal_store(unit, TBU(h));   // the row's window: one 16-byte store per lane
                if (ntail > 0) __builtin_nontemporal_store(unit2, TBU2(h));
            }
        };
        // plen >= tlen + 2: the last row's tail cells v = W + c, c = 1 .. ntail, right after that row (nw.c:137-145 with the flat indices resolved: the
        // cell on the left is the previous tail cell, the cell "above" is cell (tlen, c) of this same row, the diagonal one cell (tlen, c - 1)); `row` is row tlen
        // of the lanes concerned. Fields of the row at per-lane indices: binary select trees over VALUES (a chain of `idx == j ?` selects would be taken for a
        // dynamic index and put the row into scratch).
        auto tail_cells = [&](int h, uint32_t (&row)[NPK]) __attribute__((always_inline)) {
            const bool me = TAILS && mine && ntail > 0 && tlen == h;
            if (!TAILS || __ballot(me) == 0ull) return;        // wave-uniform
            // the ten fields s0 .. s0 + 9 of the row (columns 0 .. 9), at STATIC positions: the row's registers shifted down by s0 >> 1 registers in four
            // mask-select stages (v_bfi under v_bfe_i32 masks: compare + v_cndmask pairs cost a scalar hazard each, and a chain of `idx == j ?` selects would be
            // taken for a dynamic index), then by one field where s0 is odd
            const int q = s0 >> 1;
            auto sel = [](uint32_t m, uint32_t x, uint32_t y) __attribute__((always_inline)) { return (x & m) | (y & ~m); };   // m ? x : y, bitwise
            uint32_t m8 = (uint32_t)__builtin_amdgcn_sbfe(q, 3, 1), m4 = (uint32_t)__builtin_amdgcn_sbfe(q, 2, 1), m2 = (uint32_t)__builtin_amdgcn_sbfe(q, 1, 1),
                     m1 = (uint32_t)__builtin_amdgcn_sbfe(q, 0, 1), mo = (uint32_t)__builtin_amdgcn_sbfe(s0, 0, 1);
            opaque(m8); opaque(m4); opaque(m2); opaque(m1); opaque(mo);
            uint32_t ta[14], tb[10], tc[8], tw[7], w[6];
#pragma unroll
            for (int k = 0; k < 14; ++k) ta[k] = sel(m8, (k + 8 < NPK) ? row[k + 8] : 0u, (k < NPK) ? row[k] : 0u);
#pragma unroll
            for (int k = 0; k < 10; ++k) tb[k] = sel(m4, ta[k + 4], ta[k]);
#pragma unroll
            for (int k = 0; k < 8; ++k) tc[k] = sel(m2, tb[k + 2], tb[k]);
#pragma unroll
            for (int k = 0; k < 7; ++k) tw[k] = sel(m1, tc[k + 1], tc[k]);
#pragma unroll
            for (int k = 0; k < 6; ++k) w[k] = sel(mo, __builtin_amdgcn_alignbit(tw[k + 1], tw[k], 16), tw[k]);   // field c of the row's columns: half c & 1 of w[c >> 1]
            if (me) {
                const unsigned char *pb8 = reinterpret_cast<const unsigned char *>(gP), *tb8 = reinterpret_cast<const unsigned char *>(gT);
                const int tch = (int)tb8[tlen - 1];
                int left = score;                                  // R(tlen, W), un-tilted above
                int prev_up = (int)(int16_t)(w[0] & 0xffffu) + GAP_I * tlen;   // R(tlen, 0) = B(tlen): the diagonal of the first tail cell
#pragma unroll
                for (int c = 1; c <= kNwTail; ++c) {
                    const int tf = (int)(int16_t)((c & 1) ? (w[c >> 1] >> 16) : (w[c >> 1] & 0xffffu));
                    const int up = tf + GAP_I * tlen + GAP_D * c;  // R(tlen, c)
                    const int pch = c <= ntail ? (int)pb8[W + c - 1] : 0;
                    const int mm = prev_up + ((pch == tch) ? 0 : MISMATCH), ins = up + GAP_I, del = left + GAP_D;
                    const int cell = min(mm, min(ins, del));
                    if (BT && c <= ntail) tailbits |= ((min(mm, ins) < del ? 1u : 0u) | (mm < ins ? 2u : 0u)) << (2 * c);   // "not D", "not I": the tests of nw_traceback, in its order
                    prev_up = up;
                    left = c <= ntail ? cell : left;
                }
                score = left;
            }
        };
        auto rows = [&](auto j0_tag) __attribute__((always_inline)) {
            if (BT) {   // (with the direction bits the two-array form needs more than 256 VGPRs = one wavefront per SIMD: in place, one copy per register and row)
                for (int h = 1; h <= hmax; ++h) { do_row(j0_tag, h, Mp, Mp); tail_cells(h, Mp); }
            } else {
                for (int h = 1; h <= hmax; h += 2) {
                    do_row(j0_tag, h, Mp, Mq);
                    tail_cells(h, Mq);
                    do_row(j0_tag, h + 1, Mq, Mp);             // (a row past hmax computes on and is read by nobody)
                    tail_cells(h + 1, Mp);
                }
            }
        };
        // (Tried: compile-time variants of the row that skip the registers left of every lane's row start -- 4 .. 10 of 58, chosen per
        // wavefront. Five copies of the row loop pushed the kernel from 178 to 256 VGPRs (one wavefront per SIMD): dropped.)
        rows(std::integral_constant<int, 0>{});
        if (!BT) {
            if (mine) {
                aim_result_t res;
                res.max_operations = plen + tlen;
                res.begin_offset = plen + tlen - 1;
                res.end_offset = plen + tlen;
                res.score = score;
                res.status = AIM_PAIR_OK;
                res.idx = rq.idx;
                store_result(a, pair, res);
            }
            continue;
        }
        nw_reg_walk<NPK>(a, smem, lane, tbw, mine, pair, plen, tlen, rq.idx, s0, bandc, tailbits, score, todo);
    }
#undef TBU
#undef TBU2
}

// bytes of one wavefront's table slab (BACKTRACE) and of the workgroup's LDS
inline size_t nw_reg_slab_bytes(int npk, int read_size) { (void)npk; return (size_t)(read_size + 2) * 2 * 16 * kWave; }   // per row and lane one 16-byte unit of direction bits (the band) + one for the first columns (pairs with tail cells)


// =====================================================================================================================================
// swg_reg_kernel -- Smith-Waterman-Gotoh (global, affine gaps) on short reads with the M and I rows IN REGISTERS (round 5).
//
// Same values as swg_compute / swg_traceback (SWG/DPU-WRAM/dpu/swg.c:121-171, 45-119) for the pairs it takes; swg_lane_kernel (dp_lane.hpp: rows in
// LDS, the reference's cell order literally) keeps the rest through the to-do list. One pair per lane, two cells per VGPR, rows LEFT-aligned: column
// v (pattern character v - 1) at index v - 1, register j = indices 2j (low half), 2j + 1 (high half); the boundary cell (h, 0) lives in per-lane
// scalars and enters a row as the "register left of register 0".
//   * Cells are int8 (MAX_SCORE < 127, SWG/DPU-WRAM/common/common.h:71-86) or int16, and the reference wraps on every store (quirk S3). The
//     registers hold value * SC in 16-bit fields, SC = 256 for int8 cells: v_pk_add_u16 / v_pk_mad_u16 then wrap exactly like the (int8_t) casts,
//     v_pk_min_i16 orders like a signed int8 compare -- no sign-extension instruction anywhere. MAX_SCORE is the reference's +infinity (S2), literally.
//   * The D layer is the in-row chain D[v] = min(M[v-1] + o + e, D[v-1] + e), M[v] = min(A[v], D[v]) with A = min(M_diag + cost, I). It runs in the
//     shortened form D[v] = min(A[v-1] + o + e, D[v-1] + e) (four 16-bit SDWA steps per register), which equals the reference's whenever NO
//     intermediate wraps (o >= 0: D[v-1] + o + e never beats D[v-1] + e). All costs are >= 0 (swg_reg_supported), so without a wrap every stored value is
//     >= 0, and the first candidate the reference wraps also wraps here (A[v-1] >= M[v-1]) to a NEGATIVE value that wins its minimum and reaches the
//     cell's M: the OR of all stored M fields has a sign bit set if and only if something may have wrapped. Such a pair's results are discarded and the
//     pair goes to the to-do list (none of the synthetic l = 100 sets up to e = 5 % has one).
//   * Quirk N1 / S1 (flat table, stride W = tlen + 1): plen <= tlen -- nothing aliased; plen > tlen -- cell (h, W) IS row h + 1's boundary cell {M, D}
//     (and reads, as its cell "above", its own previous value: the stored row), row 1's boundary is the row initialisation's (it is written after the
//     column's). The cells right of it (plen >= tlen + 2) read the CURRENT row's first cells and are overwritten by the next row before anything
//     else reads them: only the LAST row's matter (they hold the score), so the rows run over the columns 1 .. W like those of a plen == tlen + 1 pair and
//     the last row's tail cells W + 1 .. plen are computed once, after the loop, from the final row (at most kSwgTail of them; with CIGAR their four
//     direction bits stay in a register, `tailbits`). Cell (h, W) sits at a per-lane index: pairs with plen > tlen are queued separately and only THEIR wavefronts pick
//     {M, D} out of the last kSwgWin registers after every row (two v_cndmask per register); the queues' remainders share a last, mixed batch.
//   * BACKTRACE: FOUR BITS per cell, decided at fill time (every cell the walk compares with still holds the value the fill read, as in nw_reg_kernel):
//     "M != D" (A < D), "M != I" (M_diag + cost < I), "D != M_left + o + e" (D < A_left + o + e: the gap was extended; needs o > 0) and
//     "I != M_up + o + e" (I_up + e < M_up + o + e) -- the four tests of swg_traceback; 'M' / 'X' is the character comparison. The sign bytes of the four
//     packed differences are gathered by two v_perm_b32 (selectors 8 .. 11 replicate a source's sign bits) and merged into a dword per FOUR registers with
//     v_bfi: byte b of the dword = {low nibble: bit k = register k's "M != D" (b = 0, 1: low / high half) or "M != I" (b = 2, 3); high nibble: the two gap
//     tests}. Without a wrap the walk always finds an operation (AIM_PAIR_SWG_NO_OP needs wrapped cells: those pairs are on the to-do list).
constexpr int kSwgWin = 16;        // registers (32 indices) in which a row may END

// Registers of a row: the launchers' READ_SIZE rule (run-swg-pim-wram.py: ceil((l + l*e + 7) / 8) * 8) leaves pattern and text at READ_SIZE - 7 characters at most, and
// column v sits at index v - 1: 2 * NPK >= READ_SIZE - 7 indices hold every pair they produce (a longer pattern, legal up to READ_SIZE, goes to the to-do list).
// READ_SIZE 144 / 160 / 176 (l = 150: the common short-read length): 69 / 77 / 85 registers per row -- M and I rows alone are up to 170 registers, so these classes keep the
// PATTERN in LDS (bytes, one ds_read per two registers and row, expanded to 16-bit fields with one v_perm per register) instead of in 85 more registers.
__host__ __device__ inline int swg_reg_npk(int read_size)
{
    return read_size <= 48 ? 21 : read_size <= 64 ? 29 : read_size <= 80 ? 37 : read_size <= 96 ? 45 : read_size <= 112 ? 53 : read_size <= 128 ? 61 :
           read_size <= 144 ? 69 : read_size <= 160 ? 77 : read_size <= 176 ? 85 : 0;
}
constexpr int kSwgTail = 8;        // tail cells of the last row a pair may have beyond (tlen, W): plen <= tlen + 1 + kSwgTail
inline int swg_reg_cell_bytes(const aim_params_t &p) { return (p.flags & AIM_FLAG_SWG_W16) ? 2 : (p.max_score < 127 ? 1 : 2); }   // = swg_cell_bytes (dp_lane.hpp)

inline bool swg_reg_supported(const aim_params_t &p)
{
    if (p.algo != AIM_ALGO_SWG || swg_reg_npk(p.read_size) == 0 || p.read_size < 40) return false;
    if (p.match != 0 || p.mismatch < 1 || p.gap_o < 1 || p.gap_e < 1 || p.max_score < 0) return false;
    const long oe = (long)p.gap_o + p.gap_e, cmax = std::max<long>(oe, p.mismatch);
    if (swg_reg_cell_bytes(p) == 1)   // (two values <= 127: a wrapped sum is negative; and the column initialisation o + v e must not wrap for the lengths the launchers'
                                      //  READ_SIZE rule leaves -- at l = 150 EVERY int8 pair wraps (quirk S3) and would only pass through here to be flagged)
        return cmax <= 127 && p.max_score <= 127 && (long)p.gap_o + (p.read_size - 8L) * p.gap_e <= 127;
    return std::max<long>(p.max_score, p.gap_o) + (p.read_size + 2L) * p.gap_e + 2 * cmax < 32000;   // int16 cells: nothing can wrap
}

__host__ __device__ inline int swg_reg_units(int npk) { return ((npk + 3) / 4 + 3) / 4; }   // 16-byte units of direction bits per row and lane
__host__ __device__ inline size_t swg_reg_lds_bytes(const aim_params_t &p, int npk)   // text image / ops staging + the two pair queues (1 KB)
{
    const bool bt = (p.flags & AIM_FLAG_BACKTRACE) != 0;
    const size_t t = (size_t)((2 * npk + 3) / 4) * kWave * 4 * (npk > 61 ? 2 : 1), o = (size_t)reg_ops_stride(p.read_size) * kWave + 32;   // (npk > 61: text image + pattern image)
    return ((bt && o > t) ? o : t) + 1024;
}
inline size_t swg_reg_slab_bytes(int npk, int read_size) { (void)npk; return (size_t)(read_size + 2) * (16 + 4) * kWave; }   // per row and lane one 16-byte unit of direction bits (the band) + dword 0 (pairs with tail cells)

// the D chain of one register: Dprev / Aprev carry the left neighbour's D and A + o + e in their HIGH halves, aoe = {A.lo + oe, A.hi + oe}
__device__ __forceinline__ uint32_t swg_chain(uint32_t Dprev, uint32_t Aprev, uint32_t aoe, uint32_t e2)
{
    uint32_t d, t;   // (ONE asm statement: the compiler pads every asm statement with an s_nop)
    asm("v_add_u16_sdwa %0, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n\t"
        "v_min_i16_sdwa %0, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n\t"
        "v_add_u16_sdwa %1, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:WORD_0\n\t"
        "v_min_i16_sdwa %0, %5, %1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0 src1_sel:WORD_0"
        : "=&v"(d), "=&v"(t) : "v"(Dprev), "v"(Aprev), "v"(e2), "v"(aoe));
    return d;
}

// swg_traceback over the band of direction bits swg_reg_kernel left for ONE batch of 64 pairs (one pair per lane; see nw_reg_walk).
template <int NPK>
__device__ __forceinline__ void swg_reg_walk(const KArgs &a, char *smem, const int lane, uint32_t *tbw, const bool mine, const uint32_t pair, const int plen, const int tlen,
                                             const uint32_t idx, const int bandc, const uint32_t tailbits, const int score, uint32_t *todo)
{
    constexpr int NQ = (NPK + 3) / 4;
    typedef uint32_t aim_u32x4 __attribute__((ext_vector_type(4)));
    const int rs = a.p.read_size;
    const uint32_t *gP = reinterpret_cast<const uint32_t *>(a.patterns + (uint64_t)pair * rs);
    const uint32_t *gT = reinterpret_cast<const uint32_t *>(a.texts + (uint64_t)pair * rs);
    const unsigned long long mask_below = (1ull << lane) - 1ull;
    auto band_q0 = [&](int h) { return min(max((bandc + h - 13) >> 3, 0), NQ > 4 ? NQ - 4 : 0); };   // first dword of row h's window (as the fill)
#define TBU(h) (reinterpret_cast<aim_u32x4 *>(tbw) + ((size_t)(h) * kWave + lane))
#define TBD0(h) (tbw + ((size_t)(rs + 2) * kWave * 4 + (size_t)(h) * kWave + lane))
    {
        int begin_offset = plen + tlen - 1;
        const int end_offset = plen + tlen;
        bool lost = false;                                    // the walk left the band: the pair goes to the to-do list
        {
            // swg_traceback (swg.c:45-119) over the direction bits; 'X' / 'M' from the sequences. Ops staged in LDS (the text image is dead by now), copied out in
            // 16-byte pieces. Round 6: the rows' units are fetched eight rows at a time and the walk runs through them row by row (see nw_reg_kernel).
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (mine && !(a.dbg_flags & 5u)) {
                char *ops_g = a.ops + (uint64_t)pair * 2 * rs;
                const unsigned char *pb = reinterpret_cast<const unsigned char *>(gP), *tbytes = reinterpret_cast<const unsigned char *>(gT);
                unsigned char *ops_l = reinterpret_cast<unsigned char *>(smem) + lane * reg_ops_stride(rs) + 16;   // this lane's row (reg_ops_stride)
#define OPS(i) ops_l[(i)]
                int sentinel = end_offset - 1;
                int h = tlen, v = plen;
                int layer = 0;                                // 0: M, 1: I, 2: D
                const int Wc = tlen + 1;
                // One step of swg_traceback on a cell's bits (bit 0: M != D, bit 4: D extended, bit 16: M != I, bit 20: I extended), without branches (see nw_reg_kernel):
                // in layer D / I the step emits the gap and stays or returns to M; in layer M it changes layer (nothing emitted: the byte written at the cursor is
                // overwritten by the next step) or emits 'M' / 'X'.
#define SWG_STEP(bits_, pch_, tch_) do { const uint32_t bq_ = (bits_), pq_ = (pch_), tq_ = (tch_);   /* (the characters are read whatever the layer: see NW_STEP) */ \
                    const bool inD_ = layer == 2, inI_ = layer == 1, inM_ = layer == 0;                                                       \
                    const bool toD_ = inM_ && !(bq_ & 1u), toI_ = inM_ && (bq_ & 1u) && !(bq_ & 0x10000u), dg_ = inM_ && (bq_ & 1u) && (bq_ & 0x10000u); \
                    const uint32_t xm_ = pq_ != tq_ ? (uint32_t)'X' : (uint32_t)'M';                                                          \
                    OPS(sentinel) = (unsigned char)(inD_ ? (uint32_t)'D' : (inI_ ? (uint32_t)'I' : xm_));                                     \
                    sentinel -= (inD_ || inI_ || dg_) ? 1 : 0; v -= (inD_ || dg_) ? 1 : 0; h -= (inI_ || dg_) ? 1 : 0;                        \
                    layer = inD_ ? ((bq_ & 0x10u) ? 2 : 0) : inI_ ? ((bq_ & 0x100000u) ? 1 : 0) : toD_ ? 2 : toI_ ? 1 : 0; } while (0)
                // the pattern row staged in LDS in the ops area's own layout, the text characters of a batch's rows with the batch (see nw_reg_kernel)
                for (int b = 0; 16 * b < rs; ++b) {
                    const aim_u32x4_u w = __builtin_nontemporal_load(reinterpret_cast<const aim_u32x4_u *>(pb) + b);
                    *reinterpret_cast<uint4 *>(&OPS(16 * b)) = make_uint4(w[0], w[1], w[2], w[3]);
                }
                // A pair with tail cells starts in the last row's tail: flat index W h + v with v > W is the tail cell v - W (h == tlen: its bits are in `tailbits`,
                // its characters its own); above the last row it is cell (h + 1, v - W) of the table -- the row's first columns, far outside the band: to-do list.
                constexpr int NB = 8;                         // rows per batch
                aim_u32x4 u[NB];                              // the units of rows h0 .. h0 - NB + 1 ...
                uint32_t tc[NB];                              // ... and those rows' text characters
                int h0 = -1;
                while (h > 0 && v > 0 && !lost) {
                    while (h > 0 && v > Wc) {                 // (pairs with tail cells, until the walk is back inside the table's own columns: one dependent load per step)
                        if (h == tlen) {
                            const uint32_t nb = tailbits >> (4 * ((v - Wc) & 7));
                            SWG_STEP((nb & 1u) | ((nb & 2u) << 15) | ((nb & 4u) << 2) | ((nb & 8u) << 17), OPS(v - 1), tbytes[h - 1]);
                        } else {
                            const int vv = v - Wc, i = vv - 1;  // cell (h + 1, v - W), columns 1 .. 8: dword 0; its characters are that cell's own
                            SWG_STEP(__builtin_nontemporal_load(TBD0(h + 1)) >> (8 * (i & 1) + ((i >> 1) & 3)), OPS(vv - 1), tbytes[h]);
                        }
                    }
                    if (!(h > 0 && v > 0)) break;
                    if (h != h0) {                            // the first batch: rows h .. h - NB + 1
                        h0 = h;
#pragma unroll
                        for (int r = 0; r < NB; ++r) { u[r] = __builtin_nontemporal_load(TBU(max(h0 - r, 1))); tc[r] = tbytes[max(h0 - r, 1) - 1]; }
                    }
                    aim_u32x4 un[NB];                         // the NEXT batch is asked for before this one is walked
                    uint32_t tn[NB];
#pragma unroll
                    for (int r = 0; r < NB; ++r) { un[r] = __builtin_nontemporal_load(TBU(max(h0 - NB - r, 1))); tn[r] = tbytes[max(h0 - NB - r, 1) - 1]; }
#pragma unroll
                    for (int r = 0; r < NB; ++r) {
                        const int qw = band_q0(h0 - r);       // the row's window (per lane: the lanes' rows differ)
                        while (h == h0 - r && h > 0 && v > 0 && !lost) {     // (v <= Wc from here on)
                            const int i = v - 1, q = i >> 3;
                            lost = (uint32_t)(q - qw) >= 4u;         // (outside the window: this step's result is void, the pair goes to the to-do list)
                            SWG_STEP(band_pick(u[r][0], u[r][1], u[r][2], u[r][3], q) >> (8 * (i & 1) + ((i >> 1) & 3)), OPS(v - 1), tc[r]);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < NB; ++r) { u[r] = un[r]; tc[r] = tn[r]; }
                    h0 -= NB;
                }
                if (!lost) {
                    while (h > 0) { OPS(sentinel) = 'I'; --sentinel; --h; }
                    while (v > 0) { OPS(sentinel) = 'D'; --sentinel; --v; }
                    begin_offset = sentinel + 1;
                    const uint4 *src = reinterpret_cast<const uint4 *>(ops_l);
                    uint4 *dst = reinterpret_cast<uint4 *>(ops_g);
                    for (int qq = begin_offset >> 4; qq <= (end_offset - 1) >> 4; ++qq) dst[qq] = src[qq];
                }
#undef SWG_STEP
#undef OPS
            }
            const unsigned long long lm = __ballot(lost);
            if (lm) {   // pairs whose walk left the band: swg_lane_kernel aligns them again behind this kernel
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&todo[LANE_TODO_COUNT], (uint32_t)__builtin_popcountll(lm));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (lost) todo[LANE_TODO_LIST + base + (uint32_t)__builtin_popcountll(lm & mask_below)] = pair;
            }
        }
        if (mine && !lost) {
            aim_result_t res;
            res.max_operations = plen + tlen;
            res.begin_offset = begin_offset;
            res.end_offset = end_offset;
            res.score = score;
            res.status = AIM_PAIR_OK;
            res.idx = idx;
            store_result(a, pair, res);
        }
    }
#undef TBU
#undef TBD0
}

template <int NPK, bool BT, bool C8>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void swg_reg_kernel(KArgs a)   // (two wavefronts per SIMD: vector + accumulation registers <= 256)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    constexpr int RSK = 2 * NPK;          // indices (columns 1 .. RSK) of a row of registers
    constexpr int NWD = (RSK + 3) / 4;    // dwords of a sequence row that cover them
    constexpr int SC = C8 ? 256 : 1;
    constexpr int NQ = (NPK + 3) / 4;     // dwords of direction bits per row and lane
    constexpr int NU = (NQ + 3) / 4;      // ... in 16-byte units
    constexpr int JW = NPK - kSwgWin;     // first register in which a row may end
    constexpr bool PL = NPK > 61;         // the pattern row lives in LDS (READ_SIZE 144 .. 176), not in NPK more registers
    const int lane = threadIdx.x;
    const int rs = a.p.read_size, rsw = rs >> 2;
    const int OE = a.p.gap_o + a.p.gap_e, GAP_E = a.p.gap_e, MISMATCH = a.p.mismatch, MAXS = a.p.max_score;
    uint32_t *todo = const_cast<uint32_t *>(a.todo);                     // OUT: pairs left to swg_lane_kernel ({count @0, pair ids @16..})
    uint32_t *tbw = BT ? reinterpret_cast<uint32_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave) : nullptr;
    // (round 6: ONE 16-byte unit per row and lane -- the band of four consecutive dwords = 32 columns around the diagonal, dword q in slot q & 3; see nw_reg_kernel)
    typedef uint32_t aim_u32x4 __attribute__((ext_vector_type(4)));
#define TBU(h) (reinterpret_cast<aim_u32x4 *>(tbw) + ((size_t)(h) * kWave + lane))
    // ... and, for pairs with tail cells (plen >= tlen + 2), dword 0 of every row (columns 1 .. 8) in a plane of its own: a step up or diagonally from a tail
    // cell continues at cell (h + 1, v - W), one of the next row's first eight columns (see nw_reg_kernel)
#define TBD0(h) (tbw + ((size_t)(rs + 2) * kWave * 4 + (size_t)(h) * kWave + lane))
    const uint32_t n_groups = (a.n_pairs + kWave - 1) / kWave;
    uint32_t ones = 0x00010001u;
    opaque(ones);
    const uint32_t x2 = (uint32_t)(uint16_t)(MISMATCH * SC) * 0x00010001u, oe2 = (uint32_t)(uint16_t)(OE * SC) * 0x00010001u;
    uint32_t e2 = (uint32_t)(uint16_t)(GAP_E * SC) * 0x00010001u;
    opaque(e2);                                                          // (the chain's SDWA operands are vector registers)
    const uint32_t eH = (uint32_t)(uint16_t)(GAP_E * SC) << 16, oeH = (uint32_t)(uint16_t)(OE * SC) << 16, msH = (uint32_t)(uint16_t)(MAXS * SC) << 16;
    auto add2 = [](uint32_t x, uint32_t y) __attribute__((always_inline)) { return dps_bits(dps_from(x) + dps_from(y)); };
    auto sub2 = [](uint32_t x, uint32_t y) __attribute__((always_inline)) { return dps_bits(dps_from(x) - dps_from(y)); };
    auto min2 = [](uint32_t x, uint32_t y) __attribute__((always_inline)) { return dps_bits(dps_min(dps_from(x), dps_from(y))); };

    // Two LDS queues (pairs with plen <= tlen; pairs with plen == tlen + 1), filled from groups of 64 consecutive pairs; the row loop runs on 64
    // queued pairs of ONE class at a time.
    uint32_t *queue = reinterpret_cast<uint32_t *>(smem + swg_reg_lds_bytes(a.p, NPK) - 1024);   // 2 x 128 entries behind the text / ops area
    uint32_t qn[2] = {0u, 0u}, it = 0;                       // wave-uniform
    bool more = true;
    const unsigned long long mask_below = (1ull << lane) - 1ull;

    auto run = [&](auto isw_tag, uint32_t *q, uint32_t &qcount) __attribute__((always_inline)) {
        constexpr bool ISW = decltype(isw_tag)::value;
        const uint32_t ntake = qcount < (uint32_t)kWave ? qcount : (uint32_t)kWave;
        const bool mine = (uint32_t)lane < ntake;
        const uint32_t pair = mine ? q[lane] : 0u;
        const uint32_t moved = (ntake + lane < qcount) ? q[ntake + lane] : 0u;   // the queue's remainder moves to its front (same-wave LDS traffic is ordered)
        if (ntake + lane < qcount) q[lane] = moved;
        qcount -= ntake;
        aim_request_t rq;
        rq.pattern_len = rq.text_len = 0; rq.padding = 0; rq.idx = 0;
        if (mine) rq = load_request(a, pair);
        const int plen = rq.pattern_len, tlen = rq.text_len;
        const bool lw = ISW && plen > tlen;                   // this lane's pair has cell (h, W) = B(h + 1) (a mixed batch also holds plen <= tlen pairs)
        const int pe = lw ? tlen + 1 : plen;                  // columns the rows run over: 1 .. pe
        const uint32_t *gP = reinterpret_cast<const uint32_t *>(a.patterns + (uint64_t)pair * rs);   // (idle lanes: pair 0's rows, read and ignored)
        const uint32_t *gT = reinterpret_cast<const uint32_t *>(a.texts + (uint64_t)pair * rs);
        // the text row goes to LDS, transposed [dword][lane] (one conflict-free ds_read per ROW of the table), the pattern row into registers as
        // 16-bit fields (character v - 1 at index v - 1: no shift)
        uint32_t *ldsT = reinterpret_cast<uint32_t *>(smem);
        uint32_t *ldsP = ldsT + NWD * kWave;                  // PL: the pattern rows, transposed like the text's
        uint32_t pc[PL ? 1 : NPK];
        __syncthreads();                                      // (single wavefront: the previous batch's traceback is done with this area)
#pragma unroll
        for (int i = 0; i < NWD; ++i) {
            const uint32_t pw = i < rsw ? gP[i] : 0u;
            ldsT[i * kWave + lane] = i < rsw ? gT[i] : 0u;
            if (PL) ldsP[i * kWave + lane] = pw;
            else {
                if (2 * i < NPK) pc[2 * i] = __builtin_amdgcn_perm(0u, pw, 0x0c010c00u);           // bytes 0, 1 -> 16-bit fields (indices 4i, 4i + 1)
                if (2 * i + 1 < NPK) pc[2 * i + 1] = __builtin_amdgcn_perm(0u, pw, 0x0c030c02u);   // bytes 2, 3
            }
        }
        // row 0 (the "first column" loop of swg_compute, swg.c:131-136): M = D = o + v e (wrapped like the reference's store), I = MAX_SCORE; columns
        // beyond plen hold 0 (never read by a cell of the pair). plen == tlen + 1: flat cell (0, W) IS the boundary cell (1, 0), and the row
        // initialisation wrote {M = I = o + e} there AFTER the column's (swg.c:137-142).
        uint32_t M[NPK], I[NPK];
        uint32_t acc = 0u;                                    // OR of every stored M: a sign bit = something may have wrapped
        {
            uint32_t val = (uint32_t)(uint16_t)(a.p.gap_o * SC);
#pragma unroll
            for (int j = 0; j < NPK; ++j) {
                uint32_t lo, hi;
                val = (val + (uint32_t)(GAP_E * SC)) & 0xffffu;
                lo = (2 * j + 1 <= pe) ? val : 0u;
                val = (val + (uint32_t)(GAP_E * SC)) & 0xffffu;
                hi = (2 * j + 2 <= pe) ? val : 0u;
                uint32_t ilo = msH >> 16, ihi = msH >> 16;
                if (ISW) {
                    if (lw && 2 * j + 1 == pe) lo = ilo = oeH >> 16;
                    if (lw && 2 * j + 2 == pe) hi = ihi = oeH >> 16;
                }
                M[j] = lo | (hi << 16);
                I[j] = ilo | (ihi << 16);
                acc |= M[j];
            }
        }
        // the boundary cell of the current row, in HIGH halves: M (= I for the cells the row initialisation wrote), D
        uint32_t MbH = oeH, DbH = msH, MbOldH = 0u;           // row 1: {o + e, MAX_SCORE}; M[0][0] = 0
        const int jl = (pe - 1) >> 1;                         // the register of the row's last cell
        const bool lhi = ((pe - 1) & 1) != 0;
        // ISW: cell (h, W) is picked out of the last kSwgWin registers with v_bfi under per-register lane masks (a chain of `jl == j ? m : Ml` selects is
        // recognised as a dynamic index into an array of the m values, which then lives in scratch)
        // (the masks are made from ONE one-hot register, v_bfe_i32 per use: sixteen mask registers were sixteen registers too many)
        uint32_t hot = (ISW && jl >= JW) ? 1u << (jl - JW) : 0u;
        opaque(hot);
        const int hmax = -wave_min_i32(mine ? -tlen : 0);
        // BACKTRACE: the band's centre line relative to index h - 1 (wave-uniform). Rows are LEFT-aligned (column v at index v - 1): a pair's path runs between the
        // diagonal through (0, 0), index h - 1, and the one through (tlen, pe), index h - 1 + (pe - tlen). The window of row h is the four dwords from q0(h) on:
        // 32 indices, at least +-12 about the centre; direction bits are MADE for the window's registers only (4 bits per cell cost 9 of a register's 24
        // instructions), and a walk that leaves the window sends its pair to the to-do list (swg_lane_kernel).
        int bandc = 0;
        if (BT) {
            const int dd = pe - tlen;
            bandc = (wave_min_i32(mine ? min(dd, 0) : 0) - wave_min_i32(mine ? -max(dd, 0) : 0)) >> 1;
        }
        auto band_q0 = [&](int h) { return min(max((bandc + h - 13) >> 3, 0), NQ > 4 ? NQ - 4 : 0); };
        const bool wtail = BT && ISW && __ballot(mine && lw && plen > pe) != 0ull;   // a pair of this wavefront has tail cells
        // registers the rows run over: the last four are skipped when no lane's columns reach them (score-only; a wave-uniform test per register: a column
        // never reads a higher index, so stale values there harm nobody). l = 100 at READ_SIZE 112: 50 of 53.
        const int njrun = BT ? NPK : -wave_min_i32(mine ? -((pe + 1) >> 1) : 0);
        uint32_t tword = 0u;
        for (int h = 1; h <= hmax; ++h) {
            if (((h - 1) & 3) == 0) tword = ldsT[((h - 1) >> 2) * kWave + lane];   // one text dword per four rows
            if (h <= tlen) {                                  // (lanes whose table is complete keep their rows: the score is picked up after the loop)
                const uint32_t tch2 = ((tword >> (((h - 1) & 3) * 8)) & 0xffu) * 0x00010001u;
                acc |= MbH;
                uint32_t Dprev = DbH, Aprev = MbH + oeH;      // left of register 0: the boundary cell (its M + o + e is the reference's own del_new)
                uint32_t oldprev = MbOldH;                    // M[h-1][0] in the high half
                uint32_t Ml = 0u, Dl = 0u;                    // ISW: {M, D} of the row's last cell
                uint32_t dword = 0u;
                aim_u32x4 unit = {0u, 0u, 0u, 0u};
                const int q0 = BT ? band_q0(h) : 0;           // wave-uniform
                uint32_t diag = __builtin_amdgcn_alignbit(M[0], oldprev, 16);           // {M[h-1][index - 1]} of register 0's cells
                uint32_t pword = 0u;                                                    // PL: the pattern dword of the current two registers
                // one register (two cells). BITS: with its four direction bits per cell (BACKTRACE, registers of the dwords the row keeps: ONE wave-uniform branch per dword
                // of four registers; a branch per register left two wavefronts per SIMD waiting on 37 taken branches per row)
                auto reg = [&](int j, auto bits_tag) __attribute__((always_inline)) {
                    constexpr bool BITS = decltype(bits_tag)::value;
                    // everything that reads the OLD row's register j is taken first -- the next register's diagonal too -- so that the new value
                    // can be written over it (in place: as a plain read-later the compiler kept a v_mov copy per register and row)
                    uint32_t insn = add2(M[j], oe2);
                    uint32_t dnext = j + 1 < NPK ? __builtin_amdgcn_alignbit(M[j + 1], M[j], 16) : 0u;
                    opaque(insn);
                    opaque(dnext);
                    uint32_t pcj;
                    if (PL) {
                        if ((j & 1) == 0) pword = ldsP[(j >> 1) * kWave + lane];
                        pcj = __builtin_amdgcn_perm(0u, pword, (j & 1) ? 0x0c030c02u : 0x0c010c00u);
                    } else pcj = pc[j];
                    const uint32_t f = pk_ne01(pcj, tch2, ones);
                    const uint32_t mm = dps_bits(dps_from(f) * dps_from(x2) + dps_from(diag));   // m_match (MATCH = 0)
                    const uint32_t inse = add2(I[j], e2);
                    const uint32_t ins = min2(insn, inse);
                    const uint32_t A = min2(mm, ins);
                    const uint32_t aoe = add2(A, oe2);
                    const uint32_t d = swg_chain(Dprev, Aprev, aoe, e2);
                    uint32_t m = min2(A, d);
                    opaque(m);   // (made HERE, inside the `h <= tlen` region: otherwise the compiler sinks the 53 minima behind the region's end -- legal, min(min(A, d), d) is what a
                                 //  lane outside it needs -- and keeps every register's d alive until then: 256 VGPRs + 33 AGPRs = one wavefront per SIMD)
                    if (BITS) {
                        const uint32_t aleft = __builtin_amdgcn_alignbit(aoe, Aprev, 16);       // {A + o + e} of both cells' left neighbours
                        const uint32_t dA = sub2(A, d), dB = sub2(mm, ins), dC = sub2(d, aleft), dD = sub2(inse, insn);
                        uint32_t w1 = __builtin_amdgcn_perm(dB, dA, 0x0b0a0908u);               // bytes {A.lo, A.hi, B.lo, B.hi} sign -> 0x00 / 0xff
                        uint32_t w2 = __builtin_amdgcn_perm(dD, dC, 0x0b0a0908u);
                        opaque(w1);   // (the bits are made HERE: left to itself the compiler makes all registers' bits at the end of the row, with every
                        opaque(w2);   //  register's candidates live until then)
                        const uint32_t k1 = 0x01010101u << (j & 3), k2 = 0x10101010u << (j & 3);
                        dword = (w1 & k1) | (dword & ~k1);
                        dword = (w2 & k2) | (dword & ~k2);
                    }
                    if (ISW && j >= JW) {
                        const uint32_t wmk = (uint32_t)__builtin_amdgcn_sbfe((int)hot, j - JW, 1);   // all ones in the lanes whose cell (h, W) lies in register j
                        Ml = (m & wmk) | (Ml & ~wmk);
                        Dl = (d & wmk) | (Dl & ~wmk);
                    }
                    M[j] = m;
                    I[j] = ins;
                    acc |= m;
                    Dprev = d;
                    Aprev = aoe;
                    diag = dnext;
                };
#pragma unroll
                for (int k = 0; k < NQ; ++k) {                // dword k: registers 4k .. 4k + 3
                    const bool inw = BT && ((k >= q0 && k < q0 + 4) || (ISW && k == 0 && wtail));   // kept by the row (the band), or dword 0 and a pair of the wavefront has tail cells: wave-uniform
                    if (inw) {
#pragma unroll
                        for (int j = 4 * k; j < 4 * k + 4 && j < NPK; ++j) reg(j, std::true_type{});
                        if (k >= q0 && k < q0 + 4) unit[k & 3] = dword;
                        if (ISW && k == 0 && lw && plen > pe && !(a.dbg_flags & 4u)) __builtin_nontemporal_store(dword, TBD0(h));
                        dword = 0u;
                    } else {
#pragma unroll
                        for (int j = 4 * k; j < 4 * k + 4 && j < NPK; ++j) {
                            if (!BT && j >= NPK - 4 && j >= njrun) continue;   // (score-only: the last four registers are skipped when no lane's columns reach them)
                            reg(j, std::false_type{});
                        }
                    }
                }
                if (BT && mine && !(a.dbg_flags & 4u)) __builtin_nontemporal_store(unit, TBU(h));   // the row's window: one 16-byte store per lane
                MbOldH = MbH;
                if (ISW) {                                    // plen > tlen: the next row's boundary cell is this row's cell (h, W); else o + (h + 1) e, wrapped, D stays MAX_SCORE
                    MbH = lw ? (lhi ? (Ml & 0xffff0000u) : (Ml << 16)) : MbH + eH;
                    DbH = lw ? (lhi ? (Dl & 0xffff0000u) : (Dl << 16)) : DbH;
                } else {
                    MbH += eH;
                }
            }
        }
        // M[tlen][pe]: a binary select tree over the last kSwgWin registers
        uint32_t sreg;
        {
            uint32_t t[kSwgWin];
#pragma unroll
            for (int k = 0; k < kSwgWin; ++k) t[k] = M[JW + k];
            const int d = jl - JW;
#pragma unroll
            for (int step = 1; step < kSwgWin; step <<= 1) {
                const bool odd = (d & step) != 0;
#pragma unroll
                for (int k = 0; k + step < kSwgWin; k += 2 * step) t[k] = odd ? t[k + step] : t[k];
            }
            sreg = t[0];
        }
        int sc16 = (int)(int16_t)(lhi ? (sreg >> 16) : (sreg & 0xffffu));     // M[tlen][pe] * SC
        uint32_t tailbits = 0u;                               // BACKTRACE: the four direction bits of the last row's tail cells (cell c at bits 4c .. 4c + 3)
        if (ISW) {
            // plen >= tlen + 2: the LAST row's tail cells v = W + c, c = 1 .. plen - W, one after the other (swg.c:151-162 with the flat indices resolved: the cell
            // on the left is the previous tail cell, the cell "above" is cell (tlen, c) of this same row, the diagonal one cell (tlen, c - 1))
            const int ntail = lw ? plen - pe : 0;
            const int tmax = -wave_min_i32(mine ? -ntail : 0);
            if (tmax > 0) {
                // field idx (0 .. 7) of a row's first four registers: a binary select tree over VALUES (a chain of `idx == j ? arr[j] : r` is recognised as a
                // dynamic index, and an array passed by reference has its address taken: either way the row would live in scratch)
                static_assert(kSwgTail <= 8, "pick16 looks at four registers");
                auto pick16 = [](uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, int idx) __attribute__((always_inline)) {
                    const int d = idx >> 1;
                    const uint32_t r0 = (d & 1) ? a1 : a0, r1 = (d & 1) ? a3 : a2;
                    const uint32_t r = (d & 2) ? r1 : r0;
                    return (int)(int16_t)((idx & 1) ? (r >> 16) : (r & 0xffffu));
                };
                const uint32_t m0 = M[0], m1 = M[1], m2 = M[2], m3 = M[3], i0 = I[0], i1 = I[1], i2 = I[2], i3 = I[3];
                const unsigned char *pb8 = reinterpret_cast<const unsigned char *>(gP), *tb8 = reinterpret_cast<const unsigned char *>(gT);
                const int tch = mine && tlen >= 1 ? (int)tb8[tlen - 1] : 0;
                int upM = (int)(int16_t)(MbH >> 16), upD = (int)(int16_t)(DbH >> 16);      // cell (tlen, W): picked up after the last row
                const int OEs = OE * SC, Es = GAP_E * SC, Xs = MISMATCH * SC;
                for (int c = 1; c <= tmax; ++c) {
                    if (c <= ntail) {
                        const int mU = pick16(m0, m1, m2, m3, c - 1), iU = pick16(i0, i1, i2, i3, c - 1);
                        const int dg = c == 1 ? (int)(int16_t)(MbOldH >> 16) : pick16(m0, m1, m2, m3, c - 2);   // (c == 1: the last row's boundary cell)
                        const int dn = (int)(int16_t)(upM + OEs), de = (int)(int16_t)(upD + Es), in_ = (int)(int16_t)(mU + OEs), ie = (int)(int16_t)(iU + Es);
                        const int cD = min(dn, de), cI = min(in_, ie);
                        const int mm = (int)(int16_t)(dg + (((int)pb8[pe + c - 1] == tch) ? 0 : Xs));
                        const int cM = min(mm, min(cI, cD));
                        if (BT) tailbits |= ((min(mm, cI) < cD ? 1u : 0u) | (mm < cI ? 2u : 0u) | (de < dn ? 4u : 0u) | (ie < in_ ? 8u : 0u)) << (4 * (c & 7));   // "M != D", "M != I", "D extended", "I extended" (c = 8 -> nibble 0: cell 0 is no tail cell)
                        acc |= (uint32_t)cM & 0x8000u;
                        upM = cM; upD = cD;
                        sc16 = cM;
                    }
                }
            }
        }
        const int score = C8 ? (sc16 >> 8) : sc16;
        // pairs in which something may have wrapped: to the to-do list, results discarded
        const bool bad = mine && (acc & 0x80008000u) != 0u;
        const unsigned long long badm = __ballot(bad);
        if (badm) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&todo[LANE_TODO_COUNT], (uint32_t)__builtin_popcountll(badm));
            base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (bad) todo[LANE_TODO_LIST + base + (uint32_t)__builtin_popcountll(badm & mask_below)] = pair;
        }
        if (!BT) {
            if (mine && !bad) {
                aim_result_t res;
                res.max_operations = plen + tlen;
                res.begin_offset = plen + tlen - 1;
                res.end_offset = plen + tlen;
                res.score = score;
                res.status = AIM_PAIR_OK;
                res.idx = rq.idx;
                store_result(a, pair, res);
            }
        } else
            swg_reg_walk<NPK>(a, smem, lane, tbw, mine && !bad, pair, plen, tlen, rq.idx, bandc, tailbits, score, todo);
    };

    for (;;) {
        while (qn[0] < (uint32_t)kWave && qn[1] < (uint32_t)kWave && more) {
            uint32_t grp;
            more = xcd_unit(n_groups, it, &grp);
            if (!more) break;
            ++it;
            const uint32_t cand = grp * kWave + lane;
            const bool act = cand < a.n_pairs;
            aim_request_t rc;
            rc.pattern_len = rc.text_len = 0; rc.padding = 0; rc.idx = 0;
            if (act) rc = load_request(a, cand);
            // this kernel's pairs: at most kSwgTail tail cells in the last row (score-only and with CIGAR), the rows' end inside the window, both sequences inside the
            // row / the staged text image
            const int cpe = rc.pattern_len > rc.text_len ? rc.text_len + 1 : rc.pattern_len;
            const bool take = act && rc.pattern_len >= 1 && rc.text_len >= 1 && rc.pattern_len <= rc.text_len + 1 + kSwgTail && cpe >= RSK - 2 * kSwgWin + 1 &&
                              rc.pattern_len <= RSK && rc.text_len <= 4 * NWD && rc.text_len <= rs;
            const bool w = take && rc.pattern_len > rc.text_len;
            const unsigned long long rest = __ballot(act && !take);
            if (rest) {   // everything else: the to-do list of swg_lane_kernel (one atomic per wavefront)
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&todo[LANE_TODO_COUNT], (uint32_t)__builtin_popcountll(rest));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (act && !take) todo[LANE_TODO_LIST + base + (uint32_t)__builtin_popcountll(rest & mask_below)] = cand;
            }
            const unsigned long long t0 = __ballot(take && !w), t1 = __ballot(w);
            if (take && !w) queue[qn[0] + (uint32_t)__builtin_popcountll(t0 & mask_below)] = cand;
            if (w) queue[128 + qn[1] + (uint32_t)__builtin_popcountll(t1 & mask_below)] = cand;
            qn[0] += (uint32_t)__builtin_popcountll(t0);
            qn[1] += (uint32_t)__builtin_popcountll(t1);
        }
        bool second = qn[1] >= (uint32_t)kWave;
        if (qn[0] < (uint32_t)kWave && !second) {   // no groups left, both queues below 64: their remainders share the last batches (the plen > tlen variant takes every lane's own class)
            if (qn[0] + qn[1] == 0u) break;
            if ((uint32_t)lane < qn[0]) queue[128 + qn[1] + lane] = queue[lane];
            qn[1] += qn[0];
            qn[0] = 0u;
            second = true;
        }
        if (qn[0] >= (uint32_t)kWave) run(std::false_type{}, queue, qn[0]);    // (each variant is inlined once)
        else if (second) run(std::true_type{}, queue + 128, qn[1]);
    }
#undef TBU
}

// Kernels are instantiated in ONE translation unit (tu_dp_reg.hip defines AIM_TU_DP_REG); every other includer sees the declaration only.
#ifdef AIM_TU_DP_REG
void nw_reg_launch(const aim_params_t &p, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s)
{
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    const int npk = nw_reg_npk(p.read_size, bt);
#define AIM_NWREG(N)                                                                                           \
    do {                                                                                                       \
        if (bt) hipLaunchKernelGGL((nw_reg_kernel<N, true>), dim3(grid), dim3(kWave), lds, s, ka);             \
        else hipLaunchKernelGGL((nw_reg_kernel<N, false>), dim3(grid), dim3(kWave), lds, s, ka);               \
    } while (0)
    if (npk == 22) AIM_NWREG(22);
    else if (npk == 30) AIM_NWREG(30);
    else if (npk == 38) AIM_NWREG(38);
    else if (npk == 46) AIM_NWREG(46);
    else if (npk == 54) AIM_NWREG(54);
    else if (npk == 62) AIM_NWREG(62);
    else if (npk == 70) AIM_NWREG(70);
    else if (npk == 78) AIM_NWREG(78);
    else if (npk == 86) AIM_NWREG(86);
#undef AIM_NWREG
}
#else
void nw_reg_launch(const aim_params_t &p, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s);
#endif
#ifdef AIM_TU_SWG_REG
void swg_reg_launch(const aim_params_t &p, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s)
{
    const bool bt = p.flags & AIM_FLAG_BACKTRACE, c8 = swg_reg_cell_bytes(p) == 1;
    const int npk = swg_reg_npk(p.read_size);
#define AIM_SWGREG2(N, B, C)                                                                                                              \
    do {                                                                                                                                  \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&swg_reg_kernel<N, B, C>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL((swg_reg_kernel<N, B, C>), dim3(grid), dim3(kWave), lds, s, ka);                                               \
    } while (0)
#define AIM_SWGREG(N)                                                                                          \
    do {                                                                                                       \
        if (bt) { if (c8) AIM_SWGREG2(N, true, true); else AIM_SWGREG2(N, true, false); }                      \
        else { if (c8) AIM_SWGREG2(N, false, true); else AIM_SWGREG2(N, false, false); }                       \
    } while (0)
    if (npk == 21) AIM_SWGREG(21);
    else if (npk == 29) AIM_SWGREG(29);
    else if (npk == 37) AIM_SWGREG(37);
    else if (npk == 45) AIM_SWGREG(45);
    else if (npk == 53) AIM_SWGREG(53);
    else if (npk == 61) AIM_SWGREG(61);
    else if (npk == 69) AIM_SWGREG(69);
    else if (npk == 77) AIM_SWGREG(77);
    else if (npk == 85) AIM_SWGREG(85);
#undef AIM_SWGREG
#undef AIM_SWGREG2
}
#else
void swg_reg_launch(const aim_params_t &p, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s);
#endif

}  // namespace aim
