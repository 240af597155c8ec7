// dp_lane.hpp -- NW and SWG (full DP) kernels: ONE PAIR PER LANE, 64 pairs per wavefront.
//
// Replaces nw_compute/nw_traceback (NW/DPU-WRAM/dpu/nw.c:67-153) and
// swg_compute/swg_traceback (SWG/DPU-WRAM/dpu/swg.c:45-171).
//
// Why one pair per lane: the reference indexes its table flat with stride
// tlen+1 while v runs to plen (quirk N1/S1).  For plen > tlen row h's cell W
// lands on row h+1's boundary cell, so row h+1 cannot start before row h has
// finished -- an anti-diagonal wavefront inside one pair degenerates to a
// serial walk.  Bit-exactness (including int8/int16 wrap on store, S3) needs
// the reference's elementary operations in the reference's order; pairs are
// independent, so the parallel axis is the pair: each lane evaluates its own
// table with the reference's cell order.  Rows live in LDS (in-place, see
// below); with BACKTRACE the flat table is additionally streamed, write-only,
// to a per-wave HBM slab that is lane-interleaved (cell idx of lane l at
// [idx*64 + l]) and read back by the traceback; sequences are staged per wave
// in LDS, transposed to [dword][lane] so every per-lane read is conflict free.
#pragma once

#include <type_traits>

#include <cstdlib>
#include "aim_device.hpp"
#include "wfa_lane.hpp"   // LANE_TODO_* (the to-do list layout)

namespace aim {

// byte b of lane's sequence from the transposed LDS image
__device__ __forceinline__ uint32_t seq_byte(const uint32_t *img, int b, int lane)
{
    return (img[(b >> 2) * kWave + lane] >> ((b & 3) * 8)) & 0xffu;
}

// Cooperative, coalesced copy of 64 consecutive sequence rows into the
// transposed image.  Rows beyond n_rows are skipped.
__device__ __forceinline__ void stage_rows_transposed(uint32_t *img, const char *rows, int rsw, int n_rows, int lane)
{
    const uint32_t *g = reinterpret_cast<const uint32_t *>(rows);
    const int total = rsw * n_rows;
    for (int gi = lane; gi < total; gi += kWave) {
        const int r = gi / rsw, w = gi - r * rsw;
        img[w * kWave + r] = g[gi];
    }
}

// In-place row arrays in LDS, transposed [cell][lane] (int16): lane l's cell v at R[v*64 + l]; a wave-instruction
// touches one dword per lane pair whatever v each lane uses, so every access is bank-conflict free.
//
// One array per layer holds BOTH rows: while row h is computed, positions < v already hold row h and positions >= v
// still hold row h-1.  That is exactly the overlap the reference's flat indexing creates (flat[W*(h-1) + v] for
// v > W IS flat[W*h + (v-W)]), so its reads resolve as:
//     left (flat[W*(h-1)+v])   : v < W -> R[v] (old),  v == W -> B_h,  v > W -> R[v-W] (new)
//     diag (flat[W*(h-1)+v-1]) : v-1 < W -> old R[v-1] (carried in a register),  v-1 == W -> B_h,  v-1 > W -> R[v-1-W]
//     up   (flat[W*h+v-1])     : the cell just computed (register); for v == 1 the boundary cell B_h
//     B_h  (flat[W*h])         : h == 1 or plen < W -> the row-init value, else cell (h-1, W) = R[W] before row h
// for every plen/tlen relation, with the reference's elementary operations in the reference's order (int8/int16
// wrap on store included).  With BACKTRACE every computed cell (h, v) is also stored -- write-only -- at the
// UNIFORM index (rs+1)*h + v of a lane-interleaved HBM slab, so all 64 lanes store to one contiguous segment per
// step whatever their tlen.  The traceback reads flat index f = W*h + v exactly like the reference; flat_to_slab()
// maps f to the cell that wrote flat[f] LAST in the reference's order:
//     f = W*h' + v', v' >= 1, h' <= tlen  -> (h', v')          (an earlier tail cell (h'-1, v'+W) was overwritten)
//     v' == 0                             -> (h'-1, W) if plen >= W and h' >= 2, else the row-init boundary (h', 0)
//     h' > tlen                           -> (tlen, f - W*tlen)  (tail of the last row: nothing overwrote it)

// flat index of the reference's table -> index in the uniform-stride slab (see the comment above)
__device__ __forceinline__ int flat_to_slab(int f, int W, int S, int plen, int tlen)
{
    int hq = f / W, vq = f - hq * W;
    if (hq > tlen) { vq = f - W * tlen; hq = tlen; }
    else if (vq == 0 && hq >= 2 && plen >= W) { hq -= 1; vq = W; }
    return S * hq + vq;
}

// Can any int16 store of the NW recurrence wrap? All costs non-negative and every cell bounded by a gap-only path:
// (2*READ_SIZE + 4) * max(gap_i, gap_d) + 2 * mismatch < 32000 (the bound dp_wave_exact_ok uses). If not, the reference's
// (int16) casts on every intermediate are identities and nw_lane_kernel<.., NOWRAP = true> computes in int without them
// (3-4 sign-extension instructions per cell); storage stays int16 either way.
__host__ __device__ inline bool nw_lane_nowrap(const aim_params_t &p)
{
    if (p.gap_i < 0 || p.gap_d < 0 || p.mismatch < 0) return false;
    const long g = p.gap_i > p.gap_d ? p.gap_i : p.gap_d;
    return (2L * p.read_size + 4) * g + 2L * p.mismatch < 32000;
}

// SEQ: where the inner loop takes the pattern bytes from -- 1: the transposed LDS image; 0: global memory; 2: REGISTERS (READ_SIZE <= 128:
// the row is 32 dwords, indexed by the chunk counter, which is wave-uniform -- the compiler addresses the vector with s_set_gpr_idx, no
// scratch). These kernels scale with residency (NW l = 100: 5 / 6 / 7 workgroups per CU = 5.59 / 4.65 / 4.30 ms) and the image is a third of
// their LDS; global reads in the cell loop cost more than the residency they free (5.24 ms at 10 per CU), registers do not.
typedef uint32_t dpl_u32x32 __attribute__((ext_vector_type(32)));
template <bool BT, int SEQ, bool NOWRAP>
__global__ __launch_bounds__(64) void nw_lane_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    // NW_W16 (NW/DPU-WRAM/common/common.h:87-97): cells are int16. cell_t is the type the arithmetic runs in -- int16_t
    // with the reference's casts, or int when nw_lane_nowrap() proved them identities; LDS rows and the table are int16.
    typedef typename std::conditional<NOWRAP, int, int16_t>::type cell_t;
    const int lane = threadIdx.x;
    const int rs = a.p.read_size, rsw = rs >> 2;
    uint32_t *imgP = reinterpret_cast<uint32_t *>(smem);                             // pattern image [dword][lane]
    constexpr bool SEQ_LDS = SEQ == 1;
    int16_t *R = reinterpret_cast<int16_t *>(imgP + (SEQ_LDS ? rsw * kWave : 0));    // [(rs+1)][64]
    int16_t *tb = BT ? reinterpret_cast<int16_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave) : nullptr;
    const int GAP_D = a.p.gap_d, GAP_I = a.p.gap_i, MISMATCH = a.p.mismatch;
    // to-do mode (a.todo set; SEQ != 1: every lane reads its own pair's rows from global memory -- the listed pairs are not consecutive):
    // the pairs nw_reg_kernel (dp_reg.hpp) left over
    const bool todo_mode = SEQ != 1 && a.todo != nullptr;
    const uint32_t n_work = todo_mode ? a.todo[LANE_TODO_COUNT] : a.n_pairs;
    const uint32_t n_groups = (n_work + kWave - 1) / kWave;
    // HBM table slab: 8 consecutive slab indices of a lane form one 16-B unit, units lane-interleaved. With S a multiple
    // of 8 and the +7 offset every chunk of the row loop (v0 = 1, 9, 17, ...) starts a unit, so a guard-free chunk is ONE
    // 16-B store per lane (1 KB per wavefront) instead of eight 2-B stores: the table stream was bound by store
    // instructions (one per cell, ~16 cycles of address processing each), not by bytes.
#define TBI(idx) ((((size_t)((idx) + 7) >> 3) * kWave + lane) * 8 + (size_t)(((idx) + 7) & 7))
#define TB(idx) tb[TBI(idx)]
#define RW(v) R[(v) * kWave + lane]

    for (uint32_t it = 0;; ++it) {
        uint32_t grp;
        if (!xcd_unit(n_groups, it, &grp)) break;
        const uint32_t pair0 = grp * kWave;
        const bool active = pair0 + lane < n_work;
        const uint32_t pair = todo_mode ? (active ? a.todo[LANE_TODO_LIST + pair0 + lane] : 0u) : pair0 + lane;
        const int n_rows = min((uint32_t)kWave, n_work - pair0);
        __syncthreads();
        if ((SEQ_LDS || SEQ == 2) && !todo_mode) stage_rows_transposed(imgP, a.patterns + (uint64_t)pair0 * rs, rsw, n_rows, lane);   // SEQ == 2: into the row area, which is not live yet
        __syncthreads();
        dpl_u32x32 preg = {};
        if (SEQ == 2) {
            const uint32_t *own = reinterpret_cast<const uint32_t *>(a.patterns + (uint64_t)pair * rs);
#pragma unroll
            for (int j = 0; j < 32; ++j) preg[j] = j < rsw ? (todo_mode ? own[j] : imgP[j * kWave + lane]) : 0u;
            __syncthreads();                          // every lane holds its row before the row area is initialised
        }
        if (!active) continue;
        const aim_request_t rq = load_request(a, pair);
        const int plen = rq.pattern_len, tlen = rq.text_len;
        const uint32_t *gP32 = reinterpret_cast<const uint32_t *>(a.patterns + (uint64_t)pair * rs);
        const uint32_t *gT32 = reinterpret_cast<const uint32_t *>(a.texts + (uint64_t)pair * rs);
        const int W = tlen + 1;   // num_cols
        const int S = rs + 8;     // uniform slab stride: a multiple of 8 (READ_SIZE is), >= plen + 1
        uint32_t tw = 0;

        // nw_compute, nw.c:109-153
        {
            int cell = 0;
            RW(0) = 0;
            if (BT) TB(0) = 0;
            for (int v = 1; v <= plen; ++v) { cell += GAP_D; RW(v) = (cell_t)cell; if (BT) TB(v) = (cell_t)cell; }
            if (BT) {
                cell = 0;
                for (int h = 1; h <= tlen; ++h) { cell += GAP_I; TB(S * h) = (cell_t)cell; }
            }
        }
        cell_t score = 0;
        for (int h = 1; h <= tlen; ++h) {
            if (((h - 1) & 3) == 0) tw = gT32[(h - 1) >> 2];          // one text dword per 4 rows
            const uint32_t tch = (tw >> (((h - 1) & 3) * 8)) & 0xffu;
            const cell_t B = (h == 1 || plen < W) ? (cell_t)(h * GAP_I) : (cell_t)RW(W);
            cell_t dgc = RW(0);
            RW(0) = B;
            cell_t up = B;
            const int row = S * h;
            // chunks of 8 cells: the old-row values and pattern bytes of a chunk are fetched up front (the
            // writes of the chunk only touch lower indices), then the 8 cells run back to back in registers
            for (int v0 = 1; v0 <= plen; v0 += 8) {
                const int w0 = (v0 - 1) >> 2;
                const uint32_t pa = SEQ == 2 ? preg[w0] : SEQ_LDS ? imgP[w0 * kWave + lane] : gP32[w0];
                const uint32_t pb = SEQ == 2 ? preg[w0 + 1] : (w0 + 1 < rsw) ? (SEQ_LDS ? imgP[(w0 + 1) * kWave + lane] : gP32[w0 + 1]) : 0u;
                cell_t olds[8];
                if (__all((v0 + 7 <= plen) && (v0 + 7 < W))) {
                    // whole chunk inside the row and left of the aliasing column for EVERY lane: no guards, no branches
#pragma unroll
                    for (int j = 0; j < 8; ++j) olds[j] = (cell_t)RW(v0 + j);
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t pch = ((j < 4 ? pa : pb) >> ((j & 3) * 8)) & 0xffu;
                        const cell_t del = (cell_t)(up + GAP_D);
                        const cell_t od = olds[j];
                        const cell_t ins = (cell_t)(od + GAP_I);
                        const cell_t mm = (cell_t)(dgc + ((pch == tch) ? 0 : MISMATCH));
                        const cell_t m = min(mm, min(ins, del));
                        RW(v0 + j) = (int16_t)m;
                        olds[j] = m;          // olds[j] is consumed (dgc below reads the old value first): reuse as the output
                        up = m;
                        dgc = od;
                    }
                    if (BT) {
                        uint4 pk;
                        pk.x = (uint32_t)(uint16_t)olds[0] | ((uint32_t)(uint16_t)olds[1] << 16);
                        pk.y = (uint32_t)(uint16_t)olds[2] | ((uint32_t)(uint16_t)olds[3] << 16);
                        pk.z = (uint32_t)(uint16_t)olds[4] | ((uint32_t)(uint16_t)olds[5] << 16);
                        pk.w = (uint32_t)(uint16_t)olds[6] | ((uint32_t)(uint16_t)olds[7] << 16);
                        {   // write-once stream, read sparsely by the traceback: nontemporal (same-box A/B -4.7 %)
                            typedef uint32_t aim_u32x4 __attribute__((ext_vector_type(4)));
                            aim_u32x4 nv = {pk.x, pk.y, pk.z, pk.w};
                            __builtin_nontemporal_store(nv, reinterpret_cast<aim_u32x4 *>(tb + TBI(row + v0)));
                        }
                    }
                    score = up;
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) olds[j] = (v0 + j <= plen) ? (cell_t)RW(v0 + j) : (cell_t)0;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int v = v0 + j;
                    if (v <= plen) {
                        const cell_t old = olds[j];
                        cell_t left, dg;
                        if (v < W) { left = old; dg = dgc; }
                        else if (v == W) { left = B; dg = dgc; }
                        else { left = RW(v - W); dg = (v - 1 == W) ? B : (cell_t)RW(v - 1 - W); }
                        const uint32_t pch = ((j < 4 ? pa : pb) >> ((j & 3) * 8)) & 0xffu;
                        const cell_t del = (cell_t)(up + GAP_D);
                        const cell_t ins = (cell_t)(left + GAP_I);
                        const cell_t mm = (cell_t)(dg + ((pch == tch) ? 0 : MISMATCH));
                        const cell_t m = min(mm, min(ins, del));
                        RW(v) = m;
                        if (BT) TB(row + v) = m;
                        score = m;
                        up = m;
                        dgc = old;
                    }
                }
            }
        }
        int begin_offset = plen + tlen - 1;
        const int end_offset = plen + tlen;
        if (BT) {   // nw_traceback, nw.c:67-107
            // Edit operations are produced one byte per step, backwards. Stored straight to HBM that is one byte-store
            // instruction per step scattering to 64 different rows (measured: 1.06 of the traceback's 1.53 ms at l=100).
            // They are staged in LDS instead -- the row buffer R is dead by now; 2*READ_SIZE bytes per lane, 16-B pieces
            // lane-interleaved -- and the 16-B pieces covering [begin_offset, end_offset) are copied out afterwards.
            char *ops_g = a.ops + (uint64_t)pair * 2 * rs;
            unsigned char *ops_l = reinterpret_cast<unsigned char *>(R);
#define OPS(i) ops_l[((((i) >> 4) * kWave + lane) << 4) + ((i) & 15)]
            int sentinel = end_offset - 1;
            int h = tlen, v = plen;
            // The reference's walk reads table[at], then table[at-1], table[at-W], table[at-W-1] as its if/else chain
            // needs them. The three neighbours are fetched together (one HBM round trip per step instead of up to three
            // dependent ones) and the cell moved to is carried as the next step's table[at]: same cells, same values,
            // same comparisons in the same order.
            int c = (h > 0 && v > 0) ? (int)TB(flat_to_slab(W * h + v, W, S, plen, tlen)) : 0;
            while (h > 0 && v > 0) {
                const int at = W * h + v;
                const int cl = TB(flat_to_slab(at - 1, W, S, plen, tlen));
                const int cu = TB(flat_to_slab(at - W, W, S, plen, tlen));
                const int cd = TB(flat_to_slab(at - W - 1, W, S, plen, tlen));
                if (c == cl + GAP_D) { OPS(sentinel) = 'D'; --sentinel; --v; c = cl; }
                else if (c == cu + GAP_I) { OPS(sentinel) = 'I'; --sentinel; --h; c = cu; }
                else { OPS(sentinel) = (c == cd + MISMATCH) ? 'X' : 'M'; --sentinel; --h; --v; c = cd; }
            }
            while (h > 0) { OPS(sentinel) = 'I'; --sentinel; --h; }
            while (v > 0) { OPS(sentinel) = 'D'; --sentinel; --v; }
            begin_offset = sentinel + 1;
            {   // rows are 2*READ_SIZE bytes (a multiple of 16) at 16-B aligned addresses: whole pieces stay inside the row
                const uint4 *src = reinterpret_cast<const uint4 *>(R);
                uint4 *dst = reinterpret_cast<uint4 *>(ops_g);
                for (int q = begin_offset >> 4; q <= (end_offset - 1) >> 4; ++q) dst[q] = src[q * kWave + lane];
            }
#undef OPS
        }
        aim_result_t r;
        r.max_operations = plen + tlen;
        r.begin_offset = begin_offset;
        r.end_offset = end_offset;
        r.score = (int)score;
        r.status = AIM_PAIR_OK;
        r.idx = rq.idx;
        store_result(a, pair, r);
    }
#undef TB
#undef TBI
#undef RW
}

template <typename CELL, bool BT, int SEQ>
__global__ __launch_bounds__(64) void swg_lane_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    const int lane = threadIdx.x;
    const int rs = a.p.read_size, rsw = rs >> 2;
    uint32_t *imgP = reinterpret_cast<uint32_t *>(smem);                             // pattern image [dword][lane]
    // row buffers hold CELL-typed values (int8 cells are stored wrapped, exactly as the reference's dp_cell_t does), so
    // they are CELL-typed: the int8 configuration (MAX_SCORE < 127, e.g. l=100 e=1 %) needs half the LDS, 7 instead of 4
    // workgroups per CU -- this kernel scales with residency (NW: 4 / 6 / 7 per CU = 6.93 / 5.00 / 4.60 ms)
    constexpr bool SEQ_LDS = SEQ == 1;
    CELL *RMa = reinterpret_cast<CELL *>(imgP + (SEQ_LDS ? rsw * kWave : 0));        // M row, [(rs+1)][64]
    CELL *RIa = RMa + (rs + 1) * kWave;                                              // I row
    // dp_cell_t {M, I, D} (SWG/DPU-WRAM/common/common.h:112-118): one packed word per cell and lane, [idx][lane]:
    // int8 cells -> uint32 (M | I<<8 | D<<16), int16 cells -> uint2 ({M | I<<16}, D): a single store per cell
    typedef typename std::conditional<sizeof(CELL) == 1, uint32_t, uint2>::type cellpack_t;
    cellpack_t *tb = BT ? reinterpret_cast<cellpack_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave) : nullptr;
    auto tb_store = [&](int idx, CELL m, CELL i, CELL d) {
        if constexpr (sizeof(CELL) == 1) {
            __builtin_nontemporal_store((uint32_t)(uint8_t)m | ((uint32_t)(uint8_t)i << 8) | ((uint32_t)(uint8_t)d << 16),
                                        reinterpret_cast<uint32_t *>(tb) + (size_t)idx * kWave + lane);   // write-once stream
        } else {
            typedef uint32_t aim_u32x2 __attribute__((ext_vector_type(2)));
            aim_u32x2 nv = {(uint32_t)(uint16_t)m | ((uint32_t)(uint16_t)i << 16), (uint32_t)(uint16_t)d};
            __builtin_nontemporal_store(nv, reinterpret_cast<aim_u32x2 *>(tb) + (size_t)idx * kWave + lane);
        }
    };
    auto tb_load = [&](int idx, int &m, int &i, int &d) {
        if constexpr (sizeof(CELL) == 1) {
            const uint32_t w = tb[(size_t)idx * kWave + lane];
            m = (int8_t)(w & 0xff); i = (int8_t)((w >> 8) & 0xff); d = (int8_t)((w >> 16) & 0xff);
        } else {
            const uint2 w = tb[(size_t)idx * kWave + lane];
            m = (int16_t)(w.x & 0xffff); i = (int16_t)(w.x >> 16); d = (int16_t)(w.y & 0xffff);
        }
    };
    const int GAP_O = a.p.gap_o, GAP_E = a.p.gap_e, MATCH = a.p.match, MISMATCH = a.p.mismatch;
    const int OE = GAP_O + GAP_E;
    const int MAX_SCORE = a.p.max_score;
    // to-do mode (a.todo set; SEQ != 1: every lane reads its own pair's rows from global memory -- the listed pairs are not consecutive):
    // the pairs swg_reg_kernel (dp_reg.hpp) left over (tail cells, outliers, pairs whose int8 cells wrap)
    const bool todo_mode = SEQ != 1 && a.todo != nullptr;
    const uint32_t n_work = todo_mode ? a.todo[LANE_TODO_COUNT] : a.n_pairs;
    const uint32_t n_groups = (n_work + kWave - 1) / kWave;
#define RM(v) RMa[(v) * kWave + lane]
#define RI(v) RIa[(v) * kWave + lane]

    for (uint32_t it = 0;; ++it) {
        uint32_t grp;
        if (!xcd_unit(n_groups, it, &grp)) break;
        const uint32_t pair0 = grp * kWave;
        const bool active = pair0 + lane < n_work;
        const uint32_t pair = todo_mode ? (active ? a.todo[LANE_TODO_LIST + pair0 + lane] : 0u) : pair0 + lane;
        const int n_rows = min((uint32_t)kWave, n_work - pair0);
        __syncthreads();
        if ((SEQ_LDS || SEQ == 2) && !todo_mode) stage_rows_transposed(imgP, a.patterns + (uint64_t)pair0 * rs, rsw, n_rows, lane);   // SEQ == 2: into the row area, which is not live yet
        __syncthreads();
        dpl_u32x32 preg = {};
        if (SEQ == 2) {
            const uint32_t *own = reinterpret_cast<const uint32_t *>(a.patterns + (uint64_t)pair * rs);
#pragma unroll
            for (int j = 0; j < 32; ++j) preg[j] = j < rsw ? (todo_mode ? own[j] : imgP[j * kWave + lane]) : 0u;
            __syncthreads();                          // every lane holds its row before the row area is initialised
        }
        if (!active) continue;
        const aim_request_t rq = load_request(a, pair);
        const int plen = rq.pattern_len, tlen = rq.text_len;
        const uint32_t *gP32 = reinterpret_cast<const uint32_t *>(a.patterns + (uint64_t)pair * rs);
        const uint32_t *gT32 = reinterpret_cast<const uint32_t *>(a.texts + (uint64_t)pair * rs);
        const int W = tlen + 1;
        const int S = rs + 1;     // uniform slab stride
        uint32_t tw = 0;

        // swg_compute, swg.c:121-171
        RM(0) = 0;
        RI(0) = (CELL)MAX_SCORE;
        if (BT) tb_store(0, 0, (CELL)MAX_SCORE, (CELL)MAX_SCORE);
        for (int v = 1; v <= plen; ++v) {
            const CELL d = (CELL)(GAP_O + v * GAP_E);
            RM(v) = d;
            RI(v) = (CELL)MAX_SCORE;
            if (BT) tb_store(v, d, (CELL)MAX_SCORE, d);
        }
        if (BT) {
            for (int h = 1; h <= tlen; ++h) {
                const CELL i = (CELL)(GAP_O + h * GAP_E);
                tb_store(S * h, i, i, (CELL)MAX_SCORE);
            }
        }
        int score = 0;
        CELL nextBD = (CELL)MAX_SCORE;   // D of cell (h-1, W): the D layer of the next row's boundary cell
        for (int h = 1; h <= tlen; ++h) {
            if (((h - 1) & 3) == 0) tw = gT32[(h - 1) >> 2];
            const uint32_t tch = (tw >> (((h - 1) & 3) * 8)) & 0xffu;
            CELL BM, BI, BD;
            if (h == 1 || plen < W) { BM = BI = (CELL)(GAP_O + h * GAP_E); BD = (CELL)MAX_SCORE; }
            else { BM = (CELL)RM(W); BI = (CELL)RI(W); BD = nextBD; }
            CELL dgc = (CELL)RM(0);
            RM(0) = BM;
            RI(0) = BI;
            CELL upM = BM, upD = BD;
            const int row = S * h;
            for (int v0 = 1; v0 <= plen; v0 += 8) {   // chunked as in nw_lane_kernel
                const int w0 = (v0 - 1) >> 2;
                const uint32_t pa = SEQ == 2 ? preg[w0] : SEQ_LDS ? imgP[w0 * kWave + lane] : gP32[w0];
                const uint32_t pb = SEQ == 2 ? preg[w0 + 1] : (w0 + 1 < rsw) ? (SEQ_LDS ? imgP[(w0 + 1) * kWave + lane] : gP32[w0 + 1]) : 0u;
                CELL oldMs[8], oldIs[8];
                if (__all((v0 + 7 <= plen) && (v0 + 7 < W))) {   // guard-free chunk (see nw_lane_kernel)
#pragma unroll
                    for (int j = 0; j < 8; ++j) { oldMs[j] = (CELL)RM(v0 + j); oldIs[j] = (CELL)RI(v0 + j); }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const uint32_t pch = ((j < 4 ? pa : pb) >> ((j & 3) * 8)) & 0xffu;
                        const CELL del = min((CELL)(upM + OE), (CELL)(upD + GAP_E));
                        const CELL ins = min((CELL)(oldMs[j] + OE), (CELL)(oldIs[j] + GAP_E));
                        const CELL mm = (CELL)(dgc + ((pch == tch) ? MATCH : MISMATCH));
                        const CELL m = (CELL)min(mm, min(ins, del));
                        RM(v0 + j) = m;
                        RI(v0 + j) = ins;
                        if (BT) tb_store(row + v0 + j, m, ins, del);
                        upM = m;
                        upD = del;
                        dgc = oldMs[j];
                    }
                    score = upM;
                    continue;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool in = v0 + j <= plen;
                    oldMs[j] = in ? (CELL)RM(v0 + j) : (CELL)0;
                    oldIs[j] = in ? (CELL)RI(v0 + j) : (CELL)0;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int v = v0 + j;
                    if (v <= plen) {
                        const CELL oldM = oldMs[j], oldI = oldIs[j];
                        CELL leftM, leftI, dg;
                        if (v < W) { leftM = oldM; leftI = oldI; dg = dgc; }
                        else if (v == W) { leftM = BM; leftI = BI; dg = dgc; }
                        else { leftM = (CELL)RM(v - W); leftI = (CELL)RI(v - W); dg = (v - 1 == W) ? BM : (CELL)RM(v - 1 - W); }
                        const uint32_t pch = ((j < 4 ? pa : pb) >> ((j & 3) * 8)) & 0xffu;
                        const CELL del = min((CELL)(upM + OE), (CELL)(upD + GAP_E));
                        const CELL ins = min((CELL)(leftM + OE), (CELL)(leftI + GAP_E));
                        const CELL mm = (CELL)(dg + ((pch == tch) ? MATCH : MISMATCH));
                        const CELL m = (CELL)min(mm, min(ins, del));
                        RM(v) = m;
                        RI(v) = ins;
                        if (v == W) nextBD = del;
                        if (BT) tb_store(row + v, m, ins, del);
                        score = m;
                        upM = m;
                        upD = del;
                        dgc = oldM;
                    }
                }
            }
        }
        int begin_offset = plen + tlen - 1;
        const int end_offset = plen + tlen;
        int status = AIM_PAIR_OK;
        if (BT) {   // swg_traceback, swg.c:45-119 (ops row was memset to 'M', swg.c:261)
            // The ops row (memset to 'M' by the reference, swg.c:261, then patched one byte per step) is built in LDS --
            // the row buffers are dead by now; 2*READ_SIZE bytes per lane in lane-interleaved 16-B pieces -- and copied
            // out whole: 2*READ_SIZE/16 16-B stores per pair instead of READ_SIZE/2 dword stores plus a byte store per step.
            char *ops_g = a.ops + (uint64_t)pair * 2 * rs;
            unsigned char *ops_l = reinterpret_cast<unsigned char *>(RMa);
#define OPS(i) ops_l[((((i) >> 4) * kWave + lane) << 4) + ((i) & 15)]
            {
                uint4 *l4 = reinterpret_cast<uint4 *>(RMa);
                const uint4 mm = make_uint4(0x4D4D4D4Du, 0x4D4D4D4Du, 0x4D4D4D4Du, 0x4D4D4D4Du);
                for (int q = 0; q < (rs >> 3); ++q) l4[q * kWave + lane] = mm;
            }
            enum { L_M, L_I, L_D };
            int sentinel = end_offset - 1;
            int h = tlen, v = plen;
            int layer = L_M;
            while (h > 0 && v > 0) {
                const int at = W * h + v;
                int cm, ci, cd, um, ui, ud, lm, li, ld, gm, gi, gd;
                tb_load(flat_to_slab(at, W, S, plen, tlen), cm, ci, cd);
                tb_load(flat_to_slab(at - 1, W, S, plen, tlen), um, ui, ud);
                tb_load(flat_to_slab(at - W, W, S, plen, tlen), lm, li, ld);
                tb_load(flat_to_slab(at - W - 1, W, S, plen, tlen), gm, gi, gd);
                if (layer == L_D) {
                    OPS(sentinel) = 'D', --sentinel;
                    if (cd == um + OE) layer = L_M;
                    --v;
                } else if (layer == L_I) {
                    OPS(sentinel) = 'I', --sentinel;
                    if (ci == lm + OE) layer = L_M;
                    --h;
                } else {
                    if (cm == cd) layer = L_D;
                    else if (cm == ci) layer = L_I;
                    else if (cm == gm + MATCH) { OPS(sentinel) = 'M', --sentinel; --h; --v; }
                    else if (cm == gm + MISMATCH) { OPS(sentinel) = 'X', --sentinel; --h; --v; }
                    else { status = AIM_PAIR_SWG_NO_OP; break; }
                }
            }
            if (status == AIM_PAIR_OK) {
                while (h > 0) { OPS(sentinel) = 'I', --sentinel; --h; }
                while (v > 0) { OPS(sentinel) = 'D', --sentinel; --v; }
            }
            begin_offset = sentinel + 1;
            {   // (only the pieces that hold printed operations, ops[begin_offset, end_offset): host.c:347-349)
                const uint4 *src = reinterpret_cast<const uint4 *>(RMa);
                uint4 *dst = reinterpret_cast<uint4 *>(ops_g);
                for (int q = max(0, begin_offset) >> 4; q <= (end_offset - 1) >> 4 && q < (rs >> 3); ++q) dst[q] = src[q * kWave + lane];
            }
#undef OPS
        }
        aim_result_t r;
        r.max_operations = plen + tlen;
        r.begin_offset = begin_offset;
        r.end_offset = end_offset;
        r.score = score;
        r.status = status;
        r.idx = rq.idx;
        store_result(a, pair, r);
    }
#undef RM
#undef RI
}

inline int swg_cell_bytes(const aim_params_t &p)
{
    if (p.flags & AIM_FLAG_SWG_W16) return 2;
    return p.max_score < 127 ? 1 : 2;   // SWG/DPU-WRAM/common/common.h:71-75
}

// Returns false when even the smallest grid does not fit the scratch budget.
inline bool dp_lane_plan(const aim_params_t &p, uint32_t n_pairs, uint64_t budget, const Knobs &kn, uint32_t *grid, uint32_t *block,
                         size_t *lds, uint64_t *scratch_per_wg, size_t *scratch_total, bool *seq_lds)
{
    const uint64_t rs = (uint64_t)p.read_size;
    // uniform-stride slab: NW (rs+8) columns (16-B units of 8 cells, +7 offset), SWG (rs+1); (rs+1) rows + slack
    const uint64_t cells = (p.algo == AIM_ALGO_NW ? rs + 8 : rs + 1) * (rs + 2);
    const uint64_t cell_b = (p.algo == AIM_ALGO_NW) ? 2 : (swg_cell_bytes(p) == 1 ? 4ull : 8ull);   // SWG: one packed word per cell
    uint64_t per = cells * cell_b * kWave;
    per = (per + 255) & ~255ull;
    if (!(p.flags & AIM_FLAG_BACKTRACE)) per = 256;   // score-only: no table at all
    const uint32_t n_groups = (n_pairs + kWave - 1) / kWave;
    uint32_t g = resident_grid(kn, 12);   // capped below by what LDS admits
    const uint32_t need = ((n_groups + 7u) / 8u) * 8u;
    if (g > need) g = need < 8u ? 8u : need;
    while (g > 8 && per * g > budget) g -= 8;
    if (per * g > budget) return false;
    *grid = g;
    *block = kWave;
    const size_t img = (size_t)(p.read_size >> 2) * kWave * 4;   // pattern image only
    const size_t rows = p.algo == AIM_ALGO_NW ? (size_t)(p.read_size + 1) * kWave * 2
                                              : (size_t)2 * (p.read_size + 1) * kWave * swg_cell_bytes(p);   // SWG: M and I rows, CELL-typed
    *seq_lds = img + rows <= 150 * 1024;
    // SWG score-only does better with the pattern read from global memory and the LDS spent on residency instead (l = 100, 1 M
    // pairs, same box: image in LDS, 7 workgroups per CU 7.12 ms; no image, 8 / 10 / 11 per CU 6.66 / 6.28 / 6.30 ms). NW and the
    // CIGAR variants measure the other way (NW score-only 4.32 vs 5.24 ms, SWG CIGAR 12.5 vs 13.6 ms).
    if (p.algo == AIM_ALGO_SWG && !(p.flags & AIM_FLAG_BACKTRACE)) *seq_lds = false;
    // READ_SIZE <= 124: no image either -- the pattern row lives in REGISTERS (dp_lane_launch, SEQ = 2), the LDS it frees is residency
    // (10 instead of 7 workgroups per CU). l = 100, 1 M pairs, same box, image in LDS / registers / global memory: NW score-only 4.30 / 3.27 /
    // 5.25 ms, NW with CIGAR 7.60 / 6.86 / 11.8, SWG score-only - / 5.89 / 6.25, SWG with CIGAR 12.7 / 11.0 / 14.5.
    if (p.read_size <= 124 && !kn.dpl_no_reg && kn.dpl_seq_lds < 0) *seq_lds = false;
    // experiments (AIM_DPL_SEQ_LDS): 0 = pattern from global memory, 1 = image in LDS (where it fits), 2 = registers (READ_SIZE <= 124);
    // an explicit value overrides the defaults above
    if (kn.dpl_seq_lds == 0 || (kn.dpl_seq_lds == 2 && p.read_size <= 124)) *seq_lds = false;
    else if (kn.dpl_seq_lds == 1) *seq_lds = img + rows <= 150 * 1024;
    *lds = rows + (*seq_lds ? img : 0);
    if (*lds > 160 * 1024) return false;
    uint32_t per_cu = (uint32_t)std::min<size_t>(12, lds_workgroups_per_cu(*lds));
    if (kn.dpl_per_cu >= 0) {   // experiments: residency sweep (also lifts the 8-per-CU start value)
        per_cu = (uint32_t)std::min<size_t>((size_t)std::max(1, kn.dpl_per_cu), lds_workgroups_per_cu(*lds));
        g = std::min<uint32_t>(resident_grid(kn, per_cu), need < 8u ? 8u : need);
        while (g > 8 && per * g > budget) g -= 8;
        *grid = g;
    }
    if (g > resident_grid(kn, per_cu)) {
        g = resident_grid(kn, per_cu);
        *grid = g;
    }
    *scratch_per_wg = per;
    *scratch_total = (size_t)(per * g);
    return true;
}

// Kernels are instantiated in ONE translation unit (tu_*.hip defines AIM_TU_DP_LANE); every other includer sees the declaration only.
#ifdef AIM_TU_DP_LANE
void dp_lane_launch(const aim_params_t &p, const Knobs &kn, uint32_t grid, size_t lds, bool seq_lds, const KArgs &ka, hipStream_t s)
{
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    const int seq = seq_lds ? 1 : (p.read_size <= 124 && !kn.dpl_no_reg && kn.dpl_seq_lds != 0) ? 2 : 0;   // no image in LDS: registers when the row fits 31 dwords (preg[w0 + 1] stays inside the vector)
#define AIM_DP_LAUNCH(KERNEL)                                                                                             \
    do {                                                                                                                  \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(kWave), lds, s, ka);                                                  \
    } while (0)
    if (p.algo == AIM_ALGO_NW) {
        const bool nowrap = nw_lane_nowrap(p);
#define AIM_NW_LAUNCH(BTV, SLV) do { if (nowrap) AIM_DP_LAUNCH((nw_lane_kernel<BTV, SLV, true>)); else AIM_DP_LAUNCH((nw_lane_kernel<BTV, SLV, false>)); } while (0)
#define AIM_NW_SEQ(BTV) do { if (seq == 1) AIM_NW_LAUNCH(BTV, 1); else if (seq == 2) AIM_NW_LAUNCH(BTV, 2); else AIM_NW_LAUNCH(BTV, 0); } while (0)
        if (bt) AIM_NW_SEQ(true); else AIM_NW_SEQ(false);
#undef AIM_NW_SEQ
#undef AIM_NW_LAUNCH
    } else {
#define AIM_SWG_SEQ(CELLT, BTV) do { if (seq == 1) AIM_DP_LAUNCH((swg_lane_kernel<CELLT, BTV, 1>)); else if (seq == 2) AIM_DP_LAUNCH((swg_lane_kernel<CELLT, BTV, 2>)); \
                                     else AIM_DP_LAUNCH((swg_lane_kernel<CELLT, BTV, 0>)); } while (0)
        if (swg_cell_bytes(p) == 1) { if (bt) AIM_SWG_SEQ(int8_t, true); else AIM_SWG_SEQ(int8_t, false); }
        else { if (bt) AIM_SWG_SEQ(int16_t, true); else AIM_SWG_SEQ(int16_t, false); }
#undef AIM_SWG_SEQ
    }
#undef AIM_DP_LAUNCH
}
#else
void dp_lane_launch(const aim_params_t &p, const Knobs &kn, uint32_t grid, size_t lds, bool seq_lds, const KArgs &ka, hipStream_t s);
#endif

}  // namespace aim
