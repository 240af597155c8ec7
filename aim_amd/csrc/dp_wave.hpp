// dp_wave.hpp -- long-read NW / SWG: ONE PAIR PER WORKGROUP of NW wavefronts, row by row, each wavefront takes
// one 512-cell block (64 lanes x 8 cells) per step and the in-row carry is combined across wavefronts through LDS.
//
// Same results as nw_compute/nw_traceback (NW/DPU-WRAM/dpu/nw.c:67-153) and swg_compute/swg_traceback
// (SWG/DPU-WRAM/dpu/swg.c:45-171) for reads whose table does not fit the one-pair-per-lane kernel
// (dp_lane.hpp), e.g. BASELINE config 4: SWG, l = 10 000 (10^8 cells per pair).
//
// The reference's table is flat, indexed num_cols*h + v with num_cols = tlen+1 while v runs to plen.
// Write W = tlen+1.  In row h the cells v < W ("regular") depend only on row h-1 and on their left
// neighbour chain; the cells v >= W ("tail", only when plen > tlen) alias row h+1: cell (h, W) IS the
// boundary cell of row h+1 and (h, v > W) occupy cells that row h+1 recomputes.  Hence:
//   * regular cells of a row are computed by all lanes, K consecutive cells per lane per step; the
//     vertical/diagonal inputs come from the previous row kept in LDS; the in-row gap chain
//         D[v] = min(M[v-1]+o+e, D[v-1]+e)   (SWG)      R[v] = min(A[v], R[v-1]+g)   (NW)
//     is a prefix-min: D[v] = v*e + min_{j<v} G[j] with G[j] = A[j]+o+e-(j+1)e (A = min(diag+cost, I)),
//     evaluated with a per-lane sequential pass, one wave-level exclusive scan and a running carry.
//     This equals the reference's cell-by-cell arithmetic exactly as long as no int16 store can wrap,
//     which dp_wave_exact_ok() proves from (READ_SIZE, penalties, MAX_SCORE) before this path is taken.
//   * tail cells are evaluated sequentially after the regular part of their row with the aliased inputs
//     (left = boundary / current row, diag likewise), cell (h, W) becomes row h+1's boundary, and only
//     the last row's tail survives in the final table -- exactly what the flat table ends up holding.
//   * the final table (all three layers, int16) is kept in a per-wave HBM slab in canonical form
//     (flat index f -> row f / W, column f % W, padded rows), and the traceback is the reference's
//     value-comparison walk over flat indices.
// Pairs outside these preconditions (plen > 2*tlen, possible int16 wrap, int8 cells) take a literal
// single-lane path over a flat table in the same slab: correct for everything, slow, and rare.
#pragma once

#include <cstdlib>
#include "aim_device.hpp"

namespace aim {

constexpr int kDpK = 8;                      // cells per lane per step
constexpr int kDpBlock = kWave * kDpK;       // 512 cells per step
constexpr int kDpInf = 0x3fffffff;

// Exclusive prefix-min over the 64 lanes (and the wave total), all in DPP: row_shr 1/2/4/8 scan each 16-lane row,
// row_bcast:15 / row_bcast:31 carry the row totals across rows, wave_shr:1 makes it exclusive.
__device__ __forceinline__ int wave_excl_scan_min(int x, int lane, int *total)
{
    (void)lane;
    int v = x;
#define AIM_SCAN_STEP(ctrl, rmask) v = min(v, __builtin_amdgcn_update_dpp(kDpInf, v, ctrl, rmask, 0xf, false))
    AIM_SCAN_STEP(0x111, 0xf);
    AIM_SCAN_STEP(0x112, 0xf);
    AIM_SCAN_STEP(0x114, 0xf);
    AIM_SCAN_STEP(0x118, 0xf);
    AIM_SCAN_STEP(0x142, 0xa);
    AIM_SCAN_STEP(0x143, 0xc);
#undef AIM_SCAN_STEP
    *total = __builtin_amdgcn_readlane(v, kWave - 1);
    return __builtin_amdgcn_update_dpp(kDpInf, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
}

struct DpCell { int M, I, D; };

// Host + device: can any int16 store of the row-scan path wrap?  
__host__ __device__ inline bool dp_wave_exact_ok(const aim_params_t &p, bool swg_int8)
{
    if (swg_int8) return false;
    const long rs = p.read_size;
    if (p.algo == AIM_ALGO_NW) {
        const long g = p.gap_i > p.gap_d ? p.gap_i : p.gap_d;
        if (p.gap_i < 0 || p.gap_d < 0 || p.mismatch < 0) return (2 * rs + 4) * g + 2 * p.mismatch < 32000 && g >= 0 && p.mismatch >= 0;
        // Round 6. With costs >= 0 every stored cell is a minimum of (neighbour + cost) and so at most the cost of ANY path of the recurrence's own moves that ends in it:
        //   * a regular cell (h, v), v >= 1: diagonal steps from (0, 0) = 0 to (m, m), m = min(h, v), then |h - v| gap steps along row h / column v -- all of them cells of
        //     columns >= 1 that hold their own row's values when they are read (the flat table's aliasing only ever replaces column 0 and cells of LATER rows):
        //     <= m min(x, gi + gd) + |h - v| g <= READ_SIZE * M, M = max(min(x, gi + gd), gi, gd);
        //   * the aliased boundary cell B(h + 1) = tail cell (h, W) <= cell (h, W - 1) + gd; a tail cell (h, v >= W) <= its "up" (h, v - W) + gi, so by induction
        //     <= cell (h, v mod W) + floor(v / W) gi <= tlen M + (READ_SIZE / (tlen + 1)) g <= READ_SIZE * M + g;
        // and the reference's int16 casts of the three candidates see at most that + max(x, g). (Until round 6 the bound was the gap-only path, (2 READ_SIZE + 4) g: at the
        // launchers' costs every NW pair from READ_SIZE 3 998 on took the literal one-lane path -- 4 GCUPS, tools/strip_shape_sweep.py -- where nothing can wrap before 7 990.)
        const long x2 = p.mismatch < p.gap_i + p.gap_d ? p.mismatch : p.gap_i + p.gap_d;
        const long m = x2 > g ? x2 : g;
        return (rs + 4) * m + 2L * p.mismatch + 2 * g < 32000;
    }
    const long hi = 3L * p.gap_o + (2 * rs + 4) * p.gap_e + 2L * p.mismatch + (p.max_score > 0 ? p.max_score : 0);
    const long lo = (long)p.match * rs;   // match <= 0
    return hi < 32000 && lo > -32000 && p.max_score < 32000;
}

// The reference's loops verbatim over a flat table (plane-separated) in the slab: one lane, correct for everything (aliased rows,
// int8 / wrapping cells), slow, rare. Returns the score (nw.c:109-153, swg.c:121-171).
template <bool SWG, bool CELL8>
__device__ __forceinline__ int dp_literal_fill(const aim_params_t &p, int plen, int tlen, const unsigned char *gP, const unsigned char *gT,
                                               int16_t *TM, int16_t *TI, int16_t *TD)
{
    typedef typename std::conditional<CELL8, int8_t, int16_t>::type cell_t;
    const int O = p.gap_o, E = p.gap_e, OE = O + E, MATCH = p.match, MISMATCH = p.mismatch, GD = p.gap_d, GI = p.gap_i, MAXS = p.max_score;
    const int W = tlen + 1;
    int score = 0;

    if (!SWG) {
        int cell = 0;
        TM[0] = 0;
        for (int v = 1; v <= plen; ++v) { cell += GD; TM[v] = (int16_t)cell; }
        cell = 0;
        for (int h = 1; h <= tlen; ++h) { cell += GI; TM[(size_t)W * h] = (int16_t)cell; }
        int16_t sc = 0;
        for (int h = 1; h <= tlen; ++h) {
            const int tch = gT[h - 1];
            const size_t row = (size_t)W * h, prow = row - W;
            for (int v = 1; v <= plen; ++v) {
                const int16_t del = (int16_t)(TM[row + v - 1] + GD);
                const int16_t ins = (int16_t)(TM[prow + v] + GI);
                const int16_t mm = (int16_t)(TM[prow + v - 1] + ((gP[v - 1] == tch) ? 0 : MISMATCH));
                sc = TM[row + v] = min(mm, min(ins, del));
            }
        }
        score = sc;
    } else {
        TD[0] = (cell_t)MAXS; TI[0] = (cell_t)MAXS; TM[0] = 0;
        for (int v = 1; v <= plen; ++v) { const cell_t d = (cell_t)(O + v * E); TD[v] = d; TI[v] = (cell_t)MAXS; TM[v] = d; }
        for (int h = 1; h <= tlen; ++h) {
            const cell_t i = (cell_t)(O + h * E);
            TD[(size_t)W * h] = (cell_t)MAXS; TI[(size_t)W * h] = i; TM[(size_t)W * h] = i;
        }
        for (int h = 1; h <= tlen; ++h) {
            const int tch = gT[h - 1];
            const size_t row = (size_t)W * h, prow = row - W;
            for (int v = 1; v <= plen; ++v) {
                const cell_t del = min((cell_t)((cell_t)TM[row + v - 1] + OE), (cell_t)((cell_t)TD[row + v - 1] + E));
                const cell_t ins = min((cell_t)((cell_t)TM[prow + v] + OE), (cell_t)((cell_t)TI[prow + v] + E));
                const cell_t mm = (cell_t)((cell_t)TM[prow + v - 1] + ((gP[v - 1] == tch) ? MATCH : MISMATCH));
                const cell_t m = min(mm, min(ins, del));
                TD[row + v] = del; TI[row + v] = ins; TM[row + v] = m;
                score = m;
            }
        }
    }
    return score;
}

// nw_traceback / swg_traceback (nw.c:67-107, swg.c:45-119) over flat indices, by ONE wavefront (wave-uniform walk); the table is
// the canonical slab (flat index f lives at row f / W, column f % W of stride S, + 7) unless `literal` (flat planes). `tile`:
// 3 x 512 int16 of LDS for the 8-row x 64-column window the walk reads from when use_tile.
template <bool SWG>
__device__ __forceinline__ void dp_traceback(const aim_params_t &p, bool literal, int plen, int tlen, int S, const int16_t *TM, const int16_t *TI,
                                             const int16_t *TD, int16_t *tile, bool use_tile_in, char *ops, int lane, int &begin_offset, int &status)
{
    const int rs = p.read_size, W = tlen + 1, end_offset = plen + tlen;
    const int OE = p.gap_o + p.gap_e, MATCH = p.match, MISMATCH = p.mismatch, GD = p.gap_d, GI = p.gap_i;
#ifdef AIM_DPW_NO_TILE
    const bool use_tile = false;     // diagnostic builds: per-step reads only
#else
    const bool use_tile = use_tile_in;
#endif

    // nw_traceback / swg_traceback over flat indices (first wavefront); canonical slab unless literal
    auto addr = [&](int f) -> size_t {
        if (literal) return (size_t)f;
        const int r = f / W;
        return (size_t)r * S + 7 + (f - r * W);
    };
    int sentinel = end_offset - 1;
    int h = tlen, v = plen;
#ifdef AIM_DPW_DIAG_NO_TRACEBACK
    h = 0; v = 0;      // diagnostic builds only (results are wrong)
#endif
    const int cap = 2 * rs;
    auto put = [&](char ch) { if (lane == 0 && sentinel >= 0 && sentinel < cap) ops[sentinel] = ch; --sentinel; };
    // Tiled walk. The walk is wave-uniform and the slab is canonical -- flat index f lives at (R, C) = (f / W,
    // f % W) -- so whenever C >= 1 the cells a step compares are (R, C), (R, C-1), (R-1, C), (R-1, C-1). The 64
    // lanes fetch an 8-row x 64-column window ending at (R, C) with ONE 16-B load per lane and plane into LDS
    // (the row buffers are dead by now) and the walk reads from it until it leaves: a refill every >= 7 steps
    // instead of an HBM round trip per step (decomposition: the walk was 27 % / 20 % of NW / SWG at l = 1000).
    // Steps on the boundary column (C == 0) and the literal path keep the per-step reads.
    int16_t *tileM = tile, *tileI = tile + 512, *tileD = tile + 1024;   // [8 rows][8 units][8 cells]
    int tR = -1, tC0 = 0;                                                       // rows tR-7..tR, columns tC0..tC0+63
    auto refill = [&](int R, int C) {                                           // C >= 1
        const int u0 = ((C - 1) >> 3) - 7;                                      // units of 8 columns: unit u = 8u+1 .. 8u+8
        tR = R; tC0 = 8 * u0 + 1;
        const int r = R - (lane >> 3), u = u0 + (lane & 7);
        if (r >= 0 && u >= -1) {                                                // unit -1: column 0 (and row padding)
            const size_t e = (size_t)r * S + 8 * (size_t)(u + 1);               // == r*S + 7 + (8u+1), 16-B aligned
            *reinterpret_cast<uint4 *>(&tileM[lane * 8]) = *reinterpret_cast<const uint4 *>(&TM[e]);
            if (SWG) {
                *reinterpret_cast<uint4 *>(&tileI[lane * 8]) = *reinterpret_cast<const uint4 *>(&TI[e]);
                *reinterpret_cast<uint4 *>(&tileD[lane * 8]) = *reinterpret_cast<const uint4 *>(&TD[e]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");              // one wavefront: in-order LDS, no barrier
    };
    auto in_tile = [&](int R, int C) { return tR >= 0 && R <= tR && R - 1 >= tR - 7 && C - 1 >= tC0 && C <= tC0 + 63; };
    auto tget = [&](const int16_t *t, int r, int c) {
        const int cc = c - tC0;
        return (int)t[(((tR - r) * 8 + (cc >> 3)) << 3) + (cc & 7)];
    };
    if (!SWG) {
        // the three neighbours are fetched together (one HBM round trip per step instead of up to three dependent
        // ones) and the cell moved to becomes the next step's table[at]: same cells, values and comparison order
        int c = (h > 0 && v > 0) ? (int)TM[addr(W * h + v)] : 0;
        while (h > 0 && v > 0) {
            const int at = W * h + v;
            const int R = at / W, C = at - R * W;
            int cl, cu, cg;
            if (use_tile && C >= 1) {
                if (!in_tile(R, C)) refill(R, C);
                cl = tget(tileM, R, C - 1); cu = tget(tileM, R - 1, C); cg = tget(tileM, R - 1, C - 1);
            } else {
                cl = TM[addr(at - 1)]; cu = TM[addr(at - W)]; cg = TM[addr(at - W - 1)];
            }
            if (c == cl + GD) { put('D'); --v; c = cl; }
            else if (c == cu + GI) { put('I'); --h; c = cu; }
            else { put((c == cg + MISMATCH) ? 'X' : 'M'); --h; --v; c = cg; }
        }
    } else {
        enum { L_M, L_I, L_D };
        int layer = L_M;
        while (h > 0 && v > 0) {
            const int at = W * h + v;
            // everything any branch of this step compares, fetched together (one round trip, not a chain)
            const int R = at / W, C = at - R * W;
            int m, cdd, cii, mu, ml, mg;
            if (use_tile && C >= 1) {
                if (!in_tile(R, C)) refill(R, C);
                m = tget(tileM, R, C); cdd = tget(tileD, R, C); cii = tget(tileI, R, C);
                mu = tget(tileM, R, C - 1); ml = tget(tileM, R - 1, C); mg = tget(tileM, R - 1, C - 1);
            } else {
                const size_t a0 = addr(at);
                m = TM[a0]; cdd = TD[a0]; cii = TI[a0];
                mu = TM[addr(at - 1)]; ml = TM[addr(at - W)]; mg = TM[addr(at - W - 1)];
            }
            if (layer == L_D) {
                put('D');
                if (cdd == mu + OE) layer = L_M;
                --v;
            } else if (layer == L_I) {
                put('I');
                if (cii == ml + OE) layer = L_M;
                --h;
            } else {
                if (m == cdd) layer = L_D;
                else if (m == cii) layer = L_I;
                else if (m == mg + MATCH) { put('M'); --h; --v; }
                else if (m == mg + MISMATCH) { put('X'); --h; --v; }
                else { status = AIM_PAIR_SWG_NO_OP; break; }
            }
        }
    }
    if (status == AIM_PAIR_OK) {
        for (int i = lane; i < h; i += kWave) { const int at = sentinel - i; if (at >= 0 && at < cap) ops[at] = 'I'; }
        if (h > 0) sentinel -= h;
        for (int i = lane; i < v; i += kWave) { const int at = sentinel - i; if (at >= 0 && at < cap) ops[at] = 'D'; }
        if (v > 0) sentinel -= v;
    }
    begin_offset = sentinel + 1;
}

#ifdef AIM_DPW_STAMPS   // diagnostic builds only: s_memtime per phase, summed by thread 0, dumped into the pair's ops row
#define AIM_DPW_STAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); dpw_sum[i] += t_ - dpw_last; dpw_last = t_; } while (0)
#else
#define AIM_DPW_STAMP(i) do { } while (0)
#endif
// ALGO: AIM_ALGO_NW or AIM_ALGO_SWG.  CELL8: SWG with int8 cells (literal path only).
template <int ALGO, bool BT, bool CELL8, int NW>
__global__ __launch_bounds__(64 * NW) void dp_wave_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    constexpr bool SWG = (ALGO == AIM_ALGO_SWG);
    constexpr int NT = kWave * NW;
    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1), wv = tid >> 6;
    const int rs = a.p.read_size;
    const int rowcap = (rs + 31) & ~7;                // int16 entries per LDS row buffer; cell v lives at index v + 7 so
                                                      // that the 8-cell groups starting at v = 1 + 8j are 16-B aligned
    unsigned char *ldsP = reinterpret_cast<unsigned char *>(smem);
    int16_t *rowbuf = reinterpret_cast<int16_t *>(smem + ((rs + 31) & ~15));
    int16_t *Mrow[2] = {rowbuf + 7, rowbuf + rowcap + 7};
    int16_t *Irow[2] = {rowbuf + 2 * rowcap + 7, rowbuf + 3 * rowcap + 7};   // SWG only
    int16_t *tailM = rowbuf + (SWG ? 4 : 2) * rowcap;                 // {M, D} of cell (h, W-1): up-neighbour of the first tail cell
    int *wt = reinterpret_cast<int *>(tailM + 8);                     // [2][NW] per-wave block minima (double buffered by step)
    int *Bl = wt + 2 * NW;                                            // boundary cell published by the tail phase {M, I, D}
    int *sc_sh = Bl + 4;                                              // score broadcast
    // table slab: canonical rows of stride S, layers as planes
    const int S = (rs + 16) & ~7;
    const size_t plane = (size_t)S * (size_t)(rs + 3);
    int16_t *tb = reinterpret_cast<int16_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave);
    int16_t *TM = tb, *TI = tb + plane, *TD = tb + 2 * plane;
    const int O = a.p.gap_o, E = a.p.gap_e, OE = O + E, MATCH = a.p.match, MISMATCH = a.p.mismatch;
    const int GD = a.p.gap_d, GI = a.p.gap_i, MAXS = a.p.max_score;
    const bool exact_ok = dp_wave_exact_ok(a.p, CELL8);

    for (uint32_t it = 0;; ++it) {
        uint32_t pair;
        if (!xcd_unit(a.n_pairs, it, &pair)) break;
        const aim_request_t rq = load_request(a, pair);
        const int plen = rq.pattern_len, tlen = rq.text_len;
        const unsigned char *gP = reinterpret_cast<const unsigned char *>(a.patterns + (uint64_t)pair * rs);
        const unsigned char *gT = reinterpret_cast<const unsigned char *>(a.texts + (uint64_t)pair * rs);
        char *ops = BT ? a.ops + (uint64_t)pair * 2 * rs : nullptr;
        const int W = tlen + 1;
        int score = 0, status = AIM_PAIR_OK;
        int begin_offset = plen + tlen - 1;
        const int end_offset = plen + tlen;
        __syncthreads();
        if (BT && SWG) {   // memset(cigar->operations, 'M', 2*READ_SIZE), swg.c:261
            uint32_t *o4 = reinterpret_cast<uint32_t *>(ops);
            for (int w = tid; w < (rs >> 1); w += NT) o4[w] = 0x4D4D4D4Du;
        }
        const bool literal = !exact_ok || plen > 2 * tlen;

        if (literal) {
            // ------------------------------------------------------------------ literal single-lane path
            // The reference's loops verbatim over a flat table (plane-separated) in the slab.
            if (tid == 0) score = dp_literal_fill<SWG, CELL8>(a.p, plen, tlen, gP, gT, TM, TI, TD);
            if (tid == 0) sc_sh[0] = score;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // SWG's ops prefill by all threads completes before the traceback patches it
            __syncthreads();
            score = sc_sh[0];
        } else {
            // ------------------------------------------------------------------ row-scan path
            for (int i = tid; i < plen; i += NT) ldsP[i] = gP[i];
            const int Rr = min(plen, W - 1);          // regular columns 1..Rr
            const bool has_tail = plen >= W;
            // row 0 (and its table image) ; boundary column of the table
            int cur = 0;
            for (int v = tid; v <= Rr; v += NT) {
                int m0, i0;
                if (SWG) { m0 = v ? O + v * E : 0; i0 = MAXS; }
                else { m0 = v * GD; i0 = 0; }
                Mrow[cur][v] = (int16_t)m0;
                if (SWG) Irow[cur][v] = (int16_t)i0;
                if (BT) {   // the table feeds the traceback only: score-only launches of the row-scan path never touch the slab
                    TM[7 + v] = (int16_t)m0;
                    if (SWG) { TI[7 + v] = (int16_t)i0; TD[7 + v] = (int16_t)(v ? m0 : MAXS); }
                }
            }
            for (int h = 1 + tid; BT && h <= tlen; h += NT) {   // row-init boundary cells flat[W*h]
                const size_t at = (size_t)h * S + 7;
                if (SWG) { TM[at] = (int16_t)(O + h * E); TI[at] = (int16_t)(O + h * E); TD[at] = (int16_t)MAXS; }
                else TM[at] = (int16_t)(h * GI);
            }
#ifdef AIM_DPW_STAMPS
            unsigned long long dpw_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, dpw_last;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(dpw_last) :: "memory");
#endif
            DpCell B;                                  // boundary cell of the current row (flat[W*h])
            const int nblocks = (Rr + kDpBlock - 1) / kDpBlock;
            const int nsteps = (nblocks + NW - 1) / NW;
            // Cross-wavefront GLOBAL-memory dependency 1 of 2 (write after write): the row-init stores just above put h*GI at
            // TM[h*S+7] from whatever thread owns h; first_tail_cell (below) later overwrites TM[(h+1)*S+7] from the lane that
            // computes cell W-1, usually a different wavefront. __syncthreads() compiles to "s_waitcnt lgkmcnt(0); s_barrier"
            // here (non-tgsplit mode: the compiler relies on the CU keeping the vector-memory stream of one workgroup in
            // order), so the order of the two stores was implicit. Made explicit: every row-init store has COMPLETED (vmcnt
            // counts stores on gfx9) before any wavefront passes the first row-start barrier. Once per pair.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            for (int h = 1; h <= tlen; ++h) {
                AIM_DPW_STAMP(7);                      // tail phase / loop overhead of the previous row
                __syncthreads();                       // previous row (and its tail / boundary cell) is complete in LDS
                AIM_DPW_STAMP(0);                      // row-start barrier
                const int nxt = cur ^ 1;
                const int tch = gT[h - 1];
                if (h == 1 || !has_tail) {
                    if (SWG) { B.M = O + h * E; B.I = B.M; B.D = MAXS; }
                    else { B.M = h * GI; B.I = B.D = 0; }
                } else {
                    B.M = Bl[0]; B.I = Bl[1]; B.D = Bl[2];   // cell (h-1, W) landed on flat[W*h]
                }
                if (tid == 0) { Mrow[nxt][0] = (int16_t)B.M; if (SWG) Irow[nxt][0] = (int16_t)B.I; }
                int carry = SWG ? min(B.D, B.M + O) : B.M;     // G[0]
                const size_t trow = (size_t)h * S + 7;
                // First tail cell (h, W) of a row before the last -- the only tail cell of such a row that is ever used
                // (it is row h+1's boundary). Its inputs are this row's boundary (uniform), cell (h, W-1) and the previous
                // row's cell (W-1): all in the registers of the lane that computes cell W-1, so that lane produces it and
                // publishes the boundary; no extra barrier, no serial phase on the first wavefront (which the stamps put at
                // 21 % of a tailed pair's row). The next row-start barrier publishes Bl; every thread read this row's Bl
                // before its first step barrier.
                auto first_tail_cell = [&](int upM, int upD, int diagM) {
                    const int pch = ldsP[W - 1];
                    DpCell c;
                    if (SWG) {
                        c.D = min(upM + OE, upD + E);
                        c.I = min(B.M + OE, B.I + E);
                        c.M = min(diagM + ((pch == tch) ? MATCH : MISMATCH), min(c.I, c.D));
                    } else {
                        c.I = c.D = 0;
                        c.M = min(diagM + ((pch == tch) ? 0 : MISMATCH), min(B.M + GI, upM + GD));
                    }
                    Bl[0] = c.M; Bl[1] = c.I; Bl[2] = c.D;
                    const size_t tdst = (size_t)(h + 1) * S + 7;   // canonical home of flat[W*h + W]
                    if (BT) {
                        TM[tdst] = (int16_t)c.M;
                        if (SWG) { TI[tdst] = (int16_t)c.I; TD[tdst] = (int16_t)c.D; }
                    }
                };
                for (int step = 0; step < nsteps; ++step) {
                    const int base = 1 + (step * NW + wv) * kDpBlock;   // may lie past Rr: the wave still joins the barrier
                    const int v0 = base + lane * kDpK;
                    const bool full = v0 + kDpK - 1 <= Rr;   // whole 8-cell group inside the row: 16-B vector traffic
                    // v * gap for the 8 cells of this lane: ONE full-rate 24-bit multiply per step (v < 2^24, gap small) and
                    // additions; the per-cell 32-bit multiplies this replaces are quarter-rate instructions
                    const int GE = SWG ? E : GD;
                    const int v0E = __mul24(v0, GE);
                    int A[kDpK], Iv[kDpK], G[kDpK];
                    int lane_min = kDpInf;
                    int16_t lm[kDpK + 1], li[kDpK];
                    unsigned char pc[kDpK];
                    if (full) {
                        const uint4 qm = *reinterpret_cast<const uint4 *>(&Mrow[cur][v0]);
                        const uint32_t wm[4] = {qm.x, qm.y, qm.z, qm.w};
                        lm[0] = Mrow[cur][v0 - 1];
#pragma unroll
                        for (int t = 0; t < kDpK; ++t) lm[t + 1] = (int16_t)(wm[t >> 1] >> ((t & 1) * 16));
                        if (SWG) {
                            const uint4 qi = *reinterpret_cast<const uint4 *>(&Irow[cur][v0]);
                            const uint32_t wi[4] = {qi.x, qi.y, qi.z, qi.w};
#pragma unroll
                            for (int t = 0; t < kDpK; ++t) li[t] = (int16_t)(wi[t >> 1] >> ((t & 1) * 16));
                        }
                        const uint2 qp = *reinterpret_cast<const uint2 *>(&ldsP[v0 - 1]);
#pragma unroll
                        for (int t = 0; t < kDpK; ++t) pc[t] = (unsigned char)((t < 4 ? qp.x : qp.y) >> ((t & 3) * 8));
                    } else {
#pragma unroll
                        for (int t = 0; t < kDpK; ++t) {
                            const int v = v0 + t;
                            const bool in = v <= Rr;
                            lm[t + 1] = in ? Mrow[cur][v] : (int16_t)0;
                            li[t] = (SWG && in) ? Irow[cur][v] : (int16_t)0;
                            pc[t] = in ? ldsP[v - 1] : (unsigned char)0;
                        }
                        lm[0] = (v0 <= Rr) ? Mrow[cur][v0 - 1] : (int16_t)0;
                    }
#pragma unroll
                    for (int t = 0; t < kDpK; ++t) {
                        const int v = v0 + t;
                        if (v <= Rr) {
                            const int leftM = lm[t + 1], diagM = lm[t];
                            const int pch = pc[t];
                            if (SWG) {
                                const int ins = min(leftM + OE, (int)li[t] + E);
                                Iv[t] = ins;
                                A[t] = min(diagM + ((pch == tch) ? MATCH : MISMATCH), ins);
                                G[t] = A[t] + OE - (v0E + (t + 1) * GE);
                            } else {
                                Iv[t] = 0;
                                A[t] = min(diagM + ((pch == tch) ? 0 : MISMATCH), leftM + GI);
                                G[t] = A[t] - (v0E + t * GE);
                            }
                        } else {
                            A[t] = Iv[t] = 0;
                            G[t] = kDpInf;
                        }
                        lane_min = min(lane_min, G[t]);
                    }
                    AIM_DPW_STAMP(1);                  // LDS row reads + A/I/G of 8 cells
                    int total;
                    const int lane_pre = wave_excl_scan_min(lane_min, lane, &total);
                    AIM_DPW_STAMP(2);                  // wave scan
                    int *wts = wt + (step & 1) * NW;
                    if (NW > 1) {
                        if (lane == 0) wts[wv] = total;
                        __syncthreads();
                    }
                    AIM_DPW_STAMP(3);                  // carry barrier
                    int before = carry;                // minimum over everything left of this wave's block
                    if (NW > 1) {
#pragma unroll
                        for (int u = 0; u < NW; ++u) {
                            const int tu = wts[u];
                            if (u < wv) before = min(before, tu);
                            carry = min(carry, tu);
                        }
                    } else {
                        carry = min(carry, total);
                    }
                    int pre = min(before, lane_pre);
                    int Mo[kDpK], Do[kDpK];
#pragma unroll
                    for (int t = 0; t < kDpK; ++t) {
                        const int v = v0 + t;
                        Do[t] = pre + (v0E + t * GE);
                        Mo[t] = min(A[t], Do[t]);
                        pre = min(pre, G[t]);
                    }
                    AIM_DPW_STAMP(4);                  // carry reads + M/D of 8 cells
                    if (full) {
                        auto pack8 = [](const int (&x)[kDpK]) {
                            uint4 r;
                            r.x = (uint32_t)(x[0] & 0xffff) | ((uint32_t)x[1] << 16);
                            r.y = (uint32_t)(x[2] & 0xffff) | ((uint32_t)x[3] << 16);
                            r.z = (uint32_t)(x[4] & 0xffff) | ((uint32_t)x[5] << 16);
                            r.w = (uint32_t)(x[6] & 0xffff) | ((uint32_t)x[7] << 16);
                            return r;
                        };
                        const uint4 pm = pack8(Mo);
                        *reinterpret_cast<uint4 *>(&Mrow[nxt][v0]) = pm;
#ifndef AIM_DPW_DIAG_NO_TABLE
                        if (BT) *reinterpret_cast<uint4 *>(&TM[trow + v0]) = pm;
#endif
                        if (SWG) {
                            const uint4 pi = pack8(Iv), pd = pack8(Do);
                            *reinterpret_cast<uint4 *>(&Irow[nxt][v0]) = pi;
#ifndef AIM_DPW_DIAG_NO_TABLE
                            if (BT) {
                                *reinterpret_cast<uint4 *>(&TI[trow + v0]) = pi;
                                *reinterpret_cast<uint4 *>(&TD[trow + v0]) = pd;
                            }
#endif
                        }
                        if (v0 + kDpK - 1 == Rr) {
                            tailM[0] = (int16_t)Mo[kDpK - 1]; tailM[1] = (int16_t)Do[kDpK - 1];
                            if (has_tail && h < tlen) first_tail_cell(Mo[kDpK - 1], Do[kDpK - 1], lm[kDpK]);
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < kDpK; ++t) {
                            const int v = v0 + t;
                            if (v <= Rr) {
                                Mrow[nxt][v] = (int16_t)Mo[t];
                                if (BT) TM[trow + v] = (int16_t)Mo[t];
                                if (SWG) {
                                    Irow[nxt][v] = (int16_t)Iv[t];
                                    if (BT) { TI[trow + v] = (int16_t)Iv[t]; TD[trow + v] = (int16_t)Do[t]; }
                                }
                                if (v == Rr) {
                                    tailM[0] = (int16_t)Mo[t]; tailM[1] = (int16_t)Do[t];
                                    if (has_tail && h < tlen) first_tail_cell(Mo[t], Do[t], lm[t + 1]);
                                }
                            }
                        }
                    }
                }
                AIM_DPW_STAMP(5);                      // pack + LDS/HBM stores (last step)
                if (has_tail && h == tlen) {           // the last row's tail survives in the table: walk it (rows before: above)
                    __syncthreads();                   // the whole regular part of row h is in LDS
                    AIM_DPW_STAMP(6);                  // tail barrier
                  if (wv == 0) {
                    // cells v = W .. plen, sequentially (wave-uniform, first wavefront only), with the aliased inputs
                    DpCell up = {(int)tailM[0], 0, (int)tailM[1]};
                    DpCell lastTail = {0, 0, 0};
                    const size_t tdst = (size_t)(h + 1) * S + 7;   // canonical row of flat[W*h + v], v >= W
                    // Only cell (h, W) of a row before the last is ever used (it lands on flat[W*(h+1)] = row h+1's boundary);
                    // cells (h, v > W) alias flat positions that row h+1 overwrites before anything reads them, and within
                    // the tail they only feed each other. Walking the whole chain W..plen on every row was dead work that
                    // set config 4's time: 87.6 ms vs 49.0 ms for the same pairs without tails (round-2 probe that swapped pattern and text; since removed).
                    const int vend = (h == tlen) ? plen : W;
                    for (int v = W; v <= vend; ++v) {
                        int leftM, leftI, diagM;
                        if (v == W) { leftM = B.M; leftI = B.I; diagM = Mrow[cur][W - 1]; }
                        else {
                            leftM = Mrow[nxt][v - W];
                            leftI = SWG ? (int)Irow[nxt][v - W] : 0;
                            diagM = (v - 1 == W) ? B.M : (int)Mrow[nxt][v - 1 - W];
                        }
                        const int pch = ldsP[v - 1];
                        DpCell c;
                        if (SWG) {
                            c.D = min(up.M + OE, up.D + E);
                            c.I = min(leftM + OE, leftI + E);
                            c.M = min(diagM + ((pch == tch) ? MATCH : MISMATCH), min(c.I, c.D));
                        } else {
                            c.I = c.D = 0;
                            c.M = min(diagM + ((pch == tch) ? 0 : MISMATCH), min(leftM + GI, up.M + GD));
                        }
                        if (v == W) lastTail = c;
                        if (BT && (h == tlen || v == W) && lane == 0) {
                            TM[tdst + (v - W)] = (int16_t)c.M;
                            if (SWG) { TI[tdst + (v - W)] = (int16_t)c.I; TD[tdst + (v - W)] = (int16_t)c.D; }
                        }
                        up = c;
                    }
                    if (lane == 0) {
                        Bl[0] = lastTail.M; Bl[1] = lastTail.I; Bl[2] = lastTail.D;
                        if (h == tlen) sc_sh[0] = up.M;
                    }
                  }
                }
                cur = nxt;
            }
            // Cross-wavefront GLOBAL-memory dependency 2 of 2 (read after write, and write after write on ops): the table
            // planes are stored by every wavefront (and SWG's 'M' prefill of the ops row by every thread), the traceback
            // below reads the table and patches ops from the FIRST wavefront only. As above the barrier alone orders LDS,
            // not outstanding global stores, so each wavefront drains its stores before it arrives. Once per pair.
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#ifdef AIM_DPW_STAMPS
            if (tid == 0) { unsigned long long *dbg = reinterpret_cast<unsigned long long *>(ops); for (int i = 0; i < 8; ++i) dbg[i] = dpw_sum[i]; }
#endif
            if (has_tail) score = sc_sh[0];
            else score = (plen >= 1 && tlen >= 1) ? (int)Mrow[cur][plen] : 0;
            if (plen == 0 || tlen == 0) score = 0;
        }

        if (BT && wv == 0)
            dp_traceback<SWG>(a.p, literal, plen, tlen, S, TM, TI, TD, rowbuf, !literal && (size_t)(SWG ? 4 : 2) * rowcap * 2 >= (size_t)(SWG ? 3 : 1) * 1024,
                              ops, lane, begin_offset, status);
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

inline int dp_wave_nw(const aim_params_t &p, uint32_t n_pairs, const Knobs &kn)
{
    // A row is ceil(nblocks / NW) sequential steps, each ending in a workgroup barrier, and block b is taken by wavefront
    // b % NW. ASSUMING wavefronts are placed round-robin on the CU's 4 SIMDs (w % 4; not verified, but the measured
    // ordering below is what this model predicts), the barrier waits for the busiest SIMD. So above 8 blocks the wavefront
    // count is chosen for the fewest blocks per row on the busiest SIMD, then the fewest steps -- not as a power of two.
    // READ_SIZE 10112 = 20 blocks: 8 wavefronts 3 steps, 10 wavefronts 2 steps but 6/4 blocks per SIMD, 12 wavefronts
    // 2 steps and 5 per SIMD. Measured on config 4 (three interleaved rounds): 54.3 / 52.3 / 49.1 ms.
    const int nblocks = (p.read_size + kDpBlock - 1) / kDpBlock;
    if (kn.dpw_nw >= 0) {   // A/B runs
        const int f = kn.dpw_nw;
        if (f == 1 || f == 2 || f == 4 || f == 8 || f == 10 || f == 12) return f;
    }
    // Up to 8 blocks per row: just enough wavefronts per pair to put ~4096 wavefronts (4 per SIMD) on the chip, never more
    // than the row has blocks. Fewer wavefronts per pair mean less (at 1: no) cross-wavefront synchronisation and more pairs
    // resident; with few pairs the wavefronts of a pair are the only parallelism there is. Measured, kernel ms at 1 / 2 / 4
    // wavefronts (round-2 probe, since removed: AIM_DPW_NW forces the count): READ_SIZE 1064 NW 4096 pairs 4.85 / 5.85 / 10.4, 1024 pairs 2.60 / 2.37 / -, 512 pairs
    // 2.49 / 2.09 / 2.38; READ_SIZE 2048, 2048 pairs NW 9.66 / 9.46 / 11.2, SWG 17.2 / 15.9 / 17.1; READ_SIZE 3072, 2048 pairs
    // NW 19.1 / 18.8 / 24.0, SWG 38.9 / 34.3 / 37.6 (8 wavefronts 32.8 / 45.6).
    if (nblocks <= 8) {
        const uint32_t target = n_pairs ? (16u * kn.cus + n_pairs - 1) / n_pairs : 4u;   // (4 096 wavefronts on 256 CUs)
        const int cap = (int)std::min<uint32_t>(target, (uint32_t)nblocks);
        return cap >= 4 ? 4 : (cap >= 2 ? 2 : 1);
    }
    int best = 8, best_busy = 1 << 30, best_steps = 1 << 30;
    for (int nw : {8, 10, 12}) {
        int per_simd[4] = {0, 0, 0, 0};
        for (int b = 0; b < nblocks; ++b) ++per_simd[(b % nw) % 4];
        const int busy = std::max(std::max(per_simd[0], per_simd[1]), std::max(per_simd[2], per_simd[3]));
        const int steps = (nblocks + nw - 1) / nw;
        if (busy < best_busy || (busy == best_busy && steps < best_steps)) { best = nw; best_busy = busy; best_steps = steps; }
    }
    return best;
}

inline bool dp_wave_plan(const aim_params_t &p, uint32_t n_pairs, uint64_t budget, const Knobs &kn, bool cell8, uint32_t *grid, uint32_t *block,
                         size_t *lds, uint64_t *scratch_per_wg, size_t *scratch_total)
{
    const uint64_t rs = (uint64_t)p.read_size;
    const uint64_t S = (rs + 16) & ~7ull;
    const bool swg = p.algo == AIM_ALGO_SWG;
    uint64_t per = (swg ? 3 : 1) * S * (rs + 3) * 2;   // int16 planes: M, I, D for SWG; NW has the one table
    per = (per + 255) & ~255ull;
    (void)cell8;
    const int nw = dp_wave_nw(p, n_pairs, kn);
    *block = (uint32_t)(kWave * nw);
    const uint64_t rowcap = (rs + 31) & ~7ull;
    *lds = ((rs + 31) & ~15ull) + (size_t)((swg ? 4 : 2) * rowcap + 64) * 2 + 256;
    if (*lds > 160 * 1024) return false;
    const uint32_t per_cu = (uint32_t)std::min<uint64_t>(32 / nw, (uint64_t)lds_workgroups_per_cu(*lds));
    uint32_t g = resident_grid(kn, per_cu);
    const uint32_t need = ((n_pairs + 7u) / 8u) * 8u;
    if (g > need) g = need < 8u ? 8u : need;
    while (g > 8 && per * g > budget) g -= 8;
    if (per * g > budget) return false;
    *grid = g;
    *scratch_per_wg = per;
    *scratch_total = (size_t)(per * g);
    return true;
}

// Kernels are instantiated in ONE translation unit (tu_*.hip defines AIM_TU_DP_WAVE); every other includer sees the declaration only.
#ifdef AIM_TU_DP_WAVE
void dp_wave_launch(const aim_params_t &p, bool cell8, uint32_t grid, uint32_t block, size_t lds, const KArgs &ka, hipStream_t s)
{
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    const int nw = (int)(block / kWave);   // the plan's choice (it depends on the number of pairs)
#define AIM_DPW(KERNEL, NWV)                                                                          \
    do {                                                                                             \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(kWave * NWV), lds, s, ka);                       \
    } while (0)
#define AIM_DPW_NW(ALGOV, BTV, C8V)                                                                   \
    do {                                                                                             \
        if (nw == 1) AIM_DPW((dp_wave_kernel<ALGOV, BTV, C8V, 1>), 1);                               \
        else if (nw == 2) AIM_DPW((dp_wave_kernel<ALGOV, BTV, C8V, 2>), 2);                          \
        else if (nw == 4) AIM_DPW((dp_wave_kernel<ALGOV, BTV, C8V, 4>), 4);                          \
        else if (nw == 8) AIM_DPW((dp_wave_kernel<ALGOV, BTV, C8V, 8>), 8);                          \
        else if (nw == 10) AIM_DPW((dp_wave_kernel<ALGOV, BTV, C8V, 10>), 10);                       \
        else AIM_DPW((dp_wave_kernel<ALGOV, BTV, C8V, 12>), 12);                                     \
    } while (0)
    if (p.algo == AIM_ALGO_NW) {
        if (bt) AIM_DPW_NW(AIM_ALGO_NW, true, false); else AIM_DPW_NW(AIM_ALGO_NW, false, false);
    } else if (cell8) {
        if (bt) AIM_DPW_NW(AIM_ALGO_SWG, true, true); else AIM_DPW_NW(AIM_ALGO_SWG, false, true);
    } else {
        if (bt) AIM_DPW_NW(AIM_ALGO_SWG, true, false); else AIM_DPW_NW(AIM_ALGO_SWG, false, false);
    }
#undef AIM_DPW_NW
#undef AIM_DPW
}
#else
void dp_wave_launch(const aim_params_t &p, bool cell8, uint32_t grid, uint32_t block, size_t lds, const KArgs &ka, hipStream_t s);
#endif

}  // namespace aim
