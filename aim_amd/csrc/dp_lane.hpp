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
// flat table literally.  The table lives in a per-wave HBM scratch slab,
// lane-interleaved (cell idx of lane l at [idx*64 + l]) so lanes of equal tlen
// touch contiguous 64-element segments; sequences are staged per wave in LDS,
// transposed to [dword][lane] so every per-lane read is bank-conflict free.
#pragma once

#include "aim_device.hpp"

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

template <bool BT, bool SEQ_LDS>
__global__ __launch_bounds__(64) void nw_lane_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    typedef int16_t cell_t;   // NW_W16, NW/DPU-WRAM/common/common.h:87-97
    const int lane = threadIdx.x;
    const int rs = a.p.read_size, rsw = rs >> 2;
    uint32_t *imgP = reinterpret_cast<uint32_t *>(smem);
    uint32_t *imgT = imgP + (SEQ_LDS ? rsw * kWave : 0);
    cell_t *tb = reinterpret_cast<cell_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave);
    const int GAP_D = a.p.gap_d, GAP_I = a.p.gap_i, MISMATCH = a.p.mismatch;
    const uint32_t n_groups = (a.n_pairs + kWave - 1) / kWave;
#define TB(idx) tb[(size_t)(idx) * kWave + lane]

    for (uint32_t it = 0;; ++it) {
        uint32_t grp;
        if (!xcd_unit(n_groups, it, &grp)) break;
        const uint32_t pair0 = grp * kWave;
        const uint32_t pair = pair0 + lane;
        const bool active = pair < a.n_pairs;
        const int n_rows = min((uint32_t)kWave, a.n_pairs - pair0);
        __syncthreads();
        if (SEQ_LDS) {
            stage_rows_transposed(imgP, a.patterns + (uint64_t)pair0 * rs, rsw, n_rows, lane);
            stage_rows_transposed(imgT, a.texts + (uint64_t)pair0 * rs, rsw, n_rows, lane);
        }
        __syncthreads();
        if (!active) continue;
        const aim_request_t rq = a.req[pair];
        const int plen = rq.pattern_len, tlen = rq.text_len;
        const unsigned char *gP = reinterpret_cast<const unsigned char *>(a.patterns + (uint64_t)pair * rs);
        const unsigned char *gT = reinterpret_cast<const unsigned char *>(a.texts + (uint64_t)pair * rs);
        const int num_cols = tlen + 1;

        // nw_compute, nw.c:109-153
        int cell = 0;
        TB(0) = 0;
        for (int v = 1; v <= plen; ++v) { cell += GAP_D; TB(v) = (cell_t)cell; }
        cell = 0;
        for (int h = 1; h <= tlen; ++h) { cell += GAP_I; TB(num_cols * h) = (cell_t)cell; }
        cell_t score = 0;
        for (int h = 1; h <= tlen; ++h) {
            const uint32_t tch = SEQ_LDS ? seq_byte(imgT, h - 1, lane) : gT[h - 1];
            const int row = num_cols * h, prow = row - num_cols;
            // flat[row + v - 1] is the cell just written (v >= 2) and flat[prow + v - 1] is the
            // previous iteration's `ins` source: both carried in registers (same values the
            // reference re-reads; no store can intervene, see DESIGN.md "NW/SWG").
            cell_t up = TB(row);
            cell_t diag = TB(prow);
            for (int v = 1; v <= plen; ++v) {
                const cell_t left = TB(prow + v);
                const uint32_t pch = SEQ_LDS ? seq_byte(imgP, v - 1, lane) : gP[v - 1];
                const cell_t del = (cell_t)(up + GAP_D);
                const cell_t ins = (cell_t)(left + GAP_I);
                const cell_t mm = (cell_t)(diag + ((pch == tch) ? 0 : MISMATCH));
                const cell_t m = min(mm, min(ins, del));
                TB(row + v) = m;
                score = m;
                up = m;
                diag = left;
            }
        }
        int begin_offset = plen + tlen - 1;
        const int end_offset = plen + tlen;
        if (BT) {   // nw_traceback, nw.c:67-107
            char *ops = a.ops + (uint64_t)pair * 2 * rs;
            int sentinel = end_offset - 1;
            int h = num_cols - 1, v = plen;
            while (h > 0 && v > 0) {
                const int at = num_cols * h + v;
                const int c = TB(at);
                if (c == TB(at - 1) + GAP_D) { ops[sentinel--] = 'D'; --v; }
                else if (c == TB(at - num_cols) + GAP_I) { ops[sentinel--] = 'I'; --h; }
                else { ops[sentinel--] = (c == TB(at - num_cols - 1) + MISMATCH) ? 'X' : 'M'; --h; --v; }
            }
            while (h > 0) { ops[sentinel--] = 'I'; --h; }
            while (v > 0) { ops[sentinel--] = 'D'; --v; }
            begin_offset = sentinel + 1;
        }
        aim_result_t r;
        r.max_operations = plen + tlen;
        r.begin_offset = begin_offset;
        r.end_offset = end_offset;
        r.score = (int)score;
        r.status = AIM_PAIR_OK;
        r.idx = rq.idx;
        a.res[pair] = r;
    }
#undef TB
}

template <typename CELL, bool BT, bool SEQ_LDS>
__global__ __launch_bounds__(64) void swg_lane_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    const int rs = a.p.read_size, rsw = rs >> 2;
    uint32_t *imgP = reinterpret_cast<uint32_t *>(smem);
    uint32_t *imgT = imgP + (SEQ_LDS ? rsw * kWave : 0);
    CELL *tb = reinterpret_cast<CELL *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave);
    const int GAP_O = a.p.gap_o, GAP_E = a.p.gap_e, MATCH = a.p.match, MISMATCH = a.p.mismatch;
    const int MAX_SCORE = a.p.max_score;
    const uint32_t n_groups = (a.n_pairs + kWave - 1) / kWave;
    // dp_cell_t {M, I, D} (SWG/DPU-WRAM/common/common.h:112-118) as three lane-interleaved planes per cell
#define TM(idx) tb[((size_t)(idx) * 3 + 0) * kWave + lane]
#define TI(idx) tb[((size_t)(idx) * 3 + 1) * kWave + lane]
#define TD(idx) tb[((size_t)(idx) * 3 + 2) * kWave + lane]

    for (uint32_t it = 0;; ++it) {
        uint32_t grp;
        if (!xcd_unit(n_groups, it, &grp)) break;
        const uint32_t pair0 = grp * kWave;
        const uint32_t pair = pair0 + lane;
        const bool active = pair < a.n_pairs;
        const int n_rows = min((uint32_t)kWave, a.n_pairs - pair0);
        __syncthreads();
        if (SEQ_LDS) {
            stage_rows_transposed(imgP, a.patterns + (uint64_t)pair0 * rs, rsw, n_rows, lane);
            stage_rows_transposed(imgT, a.texts + (uint64_t)pair0 * rs, rsw, n_rows, lane);
        }
        __syncthreads();
        if (!active) continue;
        const aim_request_t rq = a.req[pair];
        const int plen = rq.pattern_len, tlen = rq.text_len;
        const unsigned char *gP = reinterpret_cast<const unsigned char *>(a.patterns + (uint64_t)pair * rs);
        const unsigned char *gT = reinterpret_cast<const unsigned char *>(a.texts + (uint64_t)pair * rs);
        const int num_cols = tlen + 1;

        // swg_compute, swg.c:121-171
        TD(0) = (CELL)MAX_SCORE;
        TI(0) = (CELL)MAX_SCORE;
        TM(0) = 0;
        for (int v = 1; v <= plen; ++v) {
            const CELL d = (CELL)(GAP_O + v * GAP_E);
            TD(v) = d;
            TI(v) = (CELL)MAX_SCORE;
            TM(v) = d;
        }
        for (int h = 1; h <= tlen; ++h) {
            const CELL i = (CELL)(GAP_O + h * GAP_E);
            TD(num_cols * h) = (CELL)MAX_SCORE;
            TI(num_cols * h) = i;
            TM(num_cols * h) = i;
        }
        int score = 0;
        for (int h = 1; h <= tlen; ++h) {
            const uint32_t tch = SEQ_LDS ? seq_byte(imgT, h - 1, lane) : gT[h - 1];
            const int row = num_cols * h, prow = row - num_cols;
            CELL upM = TM(row), upD = TD(row);
            CELL diagM = TM(prow);
            for (int v = 1; v <= plen; ++v) {
                const CELL leftM = TM(prow + v), leftI = TI(prow + v);
                const uint32_t pch = SEQ_LDS ? seq_byte(imgP, v - 1, lane) : gP[v - 1];
                const CELL del_new = (CELL)(upM + GAP_O + GAP_E);
                const CELL del_ext = (CELL)(upD + GAP_E);
                const CELL del = min(del_new, del_ext);
                const CELL ins_new = (CELL)(leftM + GAP_O + GAP_E);
                const CELL ins_ext = (CELL)(leftI + GAP_E);
                const CELL ins = min(ins_new, ins_ext);
                const CELL mm = (CELL)(diagM + ((pch == tch) ? MATCH : MISMATCH));
                const CELL m = (CELL)min(mm, min(ins, del));
                TD(row + v) = del;
                TI(row + v) = ins;
                TM(row + v) = m;
                score = m;
                upM = m;
                upD = del;
                diagM = leftM;
            }
        }
        int begin_offset = plen + tlen - 1;
        const int end_offset = plen + tlen;
        int status = AIM_PAIR_OK;
        if (BT) {   // swg_traceback, swg.c:45-119 (ops row was memset to 'M', swg.c:261)
            char *ops = a.ops + (uint64_t)pair * 2 * rs;
            {
                uint32_t *o4 = reinterpret_cast<uint32_t *>(ops);
                for (int w = 0; w < (rs >> 1); ++w) o4[w] = 0x4D4D4D4Du;
            }
            enum { L_M, L_I, L_D };
            int sentinel = end_offset - 1;
            int h = num_cols - 1, v = plen;
            int layer = L_M;
            while (h > 0 && v > 0) {
                const int at = num_cols * h + v;
                if (layer == L_D) {
                    ops[sentinel--] = 'D';
                    if ((int)TD(at) == (int)TM(at - 1) + GAP_O + GAP_E) layer = L_M;
                    --v;
                } else if (layer == L_I) {
                    ops[sentinel--] = 'I';
                    if ((int)TI(at) == (int)TM(at - num_cols) + GAP_O + GAP_E) layer = L_M;
                    --h;
                } else {
                    const int m = TM(at);
                    if (m == (int)TD(at)) layer = L_D;
                    else if (m == (int)TI(at)) layer = L_I;
                    else if (m == (int)TM(at - num_cols - 1) + MATCH) { ops[sentinel--] = 'M'; --h; --v; }
                    else if (m == (int)TM(at - num_cols - 1) + MISMATCH) { ops[sentinel--] = 'X'; --h; --v; }
                    else { status = AIM_PAIR_SWG_NO_OP; break; }
                }
            }
            if (status == AIM_PAIR_OK) {
                while (h > 0) { ops[sentinel--] = 'I'; --h; }
                while (v > 0) { ops[sentinel--] = 'D'; --v; }
            }
            begin_offset = sentinel + 1;
        }
        aim_result_t r;
        r.max_operations = plen + tlen;
        r.begin_offset = begin_offset;
        r.end_offset = end_offset;
        r.score = score;
        r.status = status;
        r.idx = rq.idx;
        a.res[pair] = r;
    }
#undef TM
#undef TI
#undef TD
}

inline int swg_cell_bytes(const aim_params_t &p)
{
    if (p.flags & AIM_FLAG_SWG_W16) return 2;
    return p.max_score < 127 ? 1 : 2;   // SWG/DPU-WRAM/common/common.h:71-75
}

// Returns false when even the smallest grid does not fit the scratch budget.
inline bool dp_lane_plan(const aim_params_t &p, uint32_t n_pairs, uint64_t budget, uint32_t *grid, uint32_t *block,
                         size_t *lds, uint64_t *scratch_per_wg, size_t *scratch_total, bool *seq_lds)
{
    const uint64_t rs = (uint64_t)p.read_size;
    const uint64_t cells = (rs + 1) * rs + rs + 2;   // max flat index num_cols*tlen + plen, lengths <= read_size
    const uint64_t cell_b = (p.algo == AIM_ALGO_NW) ? 2 : 3ull * swg_cell_bytes(p);
    uint64_t per = cells * cell_b * kWave;
    per = (per + 255) & ~255ull;
    const uint32_t n_groups = (n_pairs + kWave - 1) / kWave;
    uint32_t g = 256 * 8;
    const uint32_t need = ((n_groups + 7u) / 8u) * 8u;
    if (g > need) g = need < 8u ? 8u : need;
    while (g > 8 && per * g > budget) g -= 8;
    if (per * g > budget) return false;
    *grid = g;
    *block = kWave;
    const size_t img = 2 * (size_t)(p.read_size >> 2) * kWave * 4;
    *seq_lds = img <= 64 * 1024;
    *lds = *seq_lds ? img : 0;
    *scratch_per_wg = per;
    *scratch_total = (size_t)(per * g);
    return true;
}

inline void dp_lane_launch(const aim_params_t &p, uint32_t grid, size_t lds, bool seq_lds, const KArgs &ka, hipStream_t s)
{
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
#define AIM_DP_LAUNCH(KERNEL) hipLaunchKernelGGL((KERNEL), dim3(grid), dim3(kWave), lds, s, ka)
    if (p.algo == AIM_ALGO_NW) {
        if (bt) { if (seq_lds) AIM_DP_LAUNCH((nw_lane_kernel<true, true>)); else AIM_DP_LAUNCH((nw_lane_kernel<true, false>)); }
        else    { if (seq_lds) AIM_DP_LAUNCH((nw_lane_kernel<false, true>)); else AIM_DP_LAUNCH((nw_lane_kernel<false, false>)); }
    } else if (swg_cell_bytes(p) == 1) {
        if (bt) { if (seq_lds) AIM_DP_LAUNCH((swg_lane_kernel<int8_t, true, true>)); else AIM_DP_LAUNCH((swg_lane_kernel<int8_t, true, false>)); }
        else    { if (seq_lds) AIM_DP_LAUNCH((swg_lane_kernel<int8_t, false, true>)); else AIM_DP_LAUNCH((swg_lane_kernel<int8_t, false, false>)); }
    } else {
        if (bt) { if (seq_lds) AIM_DP_LAUNCH((swg_lane_kernel<int16_t, true, true>)); else AIM_DP_LAUNCH((swg_lane_kernel<int16_t, true, false>)); }
        else    { if (seq_lds) AIM_DP_LAUNCH((swg_lane_kernel<int16_t, false, true>)); else AIM_DP_LAUNCH((swg_lane_kernel<int16_t, false, false>)); }
    }
#undef AIM_DP_LAUNCH
}

}  // namespace aim
