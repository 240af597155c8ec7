// genasm_wave.hpp -- GenASM (bit-vector approximate string matching, windowed traceback) for long reads: ONE PAIR PER
// WAVEFRONT, the 64 lanes are the 64 error levels of a window.  BASELINE config 5.
//
// PARITY UNPINNED: AIM's GenASM is an un-vendored, un-pinned submodule (/root/reference/.gitmodules:1-3, empty directory,
// no call sites or tests).  This kernel implements the PUBLISHED algorithm (Senol Cali et al., MICRO 2020: GenASM-DC,
// Algorithm 1; GenASM-TB and the W = 64 / O = 24 windows, Section 6) exactly as oracle/genasm_oracle.c restates it --
// every open choice is fixed there and marked [spec] -- and is tested bit for bit against that restatement.
//
// Mapping.  A pair of 100 kb reads is ~2 500 dependent windows, so the parallelism has to come from inside a window.
// GenASM-DC's levels are coupled only through the pattern-only edit (R_a[d] needs the NEW R_a[d-1] << 1):
//     R_a[d] = X_a[d] & (R_a[d-1] << 1),   X_a[d] = match & substitution & text-only edit   (all from column a+1)
//  => R_a[d] = AND_{j <= d} (X_a[d-j] << j), a prefix-AND with a shift per step: six Hillis-Steele steps over the lanes
//     (stride 1, 2, 4, ..., 32; the shift of a step equals its stride), instead of GenASM's 64-deep hardware chain.
// Almost every window needs far fewer than 16 edits, so the column loop first runs levels 0..15 only, in one 16-lane DPP row
// (four row_shr steps, no LDS crossbar), and falls back to all 64 levels (six ds_bpermute steps) when that finds no alignment.
// Lane d keeps R[d] of the current column in two VGPRs; every column is also kept ([a][d]: 16 levels in LDS, 8.3 KB per
// wavefront; all 64 in an HBM slab on the slow path) because the traceback -- a wave-uniform walk of <= ~80 steps per window -- reads R_a[d], R_{a+1}[d] and
// R_{a+1}[d-1] along its path.  Pattern masks are not tabulated: PM[c] = ~ballot(reversed pattern char == c) is one
// compare per text character and works for ANY byte values (the reference family compares raw bytes).
// Integer / bit work only; HBM sees each sequence byte once and the ops once.
#pragma once

#include "aim_device.hpp"

namespace aim {

constexpr int kGaW = 64;        // window
constexpr int kGaCommit = 40;   // W - O

__device__ __forceinline__ uint64_t ga_shfl_up(uint64_t v, int delta, int lane)
{
    // lanes < delta receive their own value (they ignore it)
    const int src = (lane - delta) < 0 ? lane : lane - delta;
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src << 2, (int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}
// lane L receives the value of lane L - S of its own 16-lane row (DPP row_shr: no LDS crossbar); lanes whose source would
// lie outside the row receive `fill`
template <int S>
__device__ __forceinline__ uint64_t ga_row_shr(uint64_t v, uint64_t fill)
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)fill, (int)(uint32_t)v, 0x110 + S, 0xf, 0xf, false);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(fill >> 32), (int)(uint32_t)(v >> 32), 0x110 + S, 0xf, 0xf, false);
    return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint64_t ga_uniform(uint64_t v)   // value known to be wave-uniform -> SGPRs
{
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

template <bool BT>
__global__ __launch_bounds__(64) void genasm_wave_kernel(KArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    // Columns of the window for the traceback. The fast path (16 levels) keeps them in LDS, [kGaW + 1][16] = 8.3 KB, so that
    // 8 wavefronts are resident per CU (all 64 levels in LDS were 33 KB: 4 per CU, one per SIMD, nothing to overlap the
    // dependent column chain with); the rare slow path (a window needing 16..63 edits) writes [kGaW + 1][64] to this
    // wavefront's slab of HBM scratch instead.
    uint64_t *Rs = reinterpret_cast<uint64_t *>(smem);
    uint64_t *Rg = reinterpret_cast<uint64_t *>(a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave);
    unsigned char *wops = reinterpret_cast<unsigned char *>(Rs + (kGaW + 1) * 16);   // ops of the current window (<= 128)
    unsigned char *pwin = wops + 192, *twin = wops + 256;                            // the window's characters, for the traceback's run test
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

        while (pi < plen && ti < tlen) {
            const int m = min(kGaW, plen - pi), n = min(kGaW, tlen - ti);
            const bool last = (m == plen - pi) && (n == tlen - ti);
            // lane j: p[m-1-j] (reversed, for the pattern masks), p[j] and t[j] (forward, for the traceback)
            const int prev = lane < m ? (int)gP[pi + m - 1 - lane] : 0x100;     // 0x100 never equals a byte
            const int pfwd = lane < m ? (int)gP[pi + lane] : 0x200;
            const int tfwd = lane < n ? (int)gT[ti + lane] : 0x300;
            pwin[lane] = (unsigned char)pfwd;
            twin[lane] = (unsigned char)tfwd;
            // FAST PATH: levels 0..15 only, in lanes 0..15, with DPP row shifts (four scan steps, no LDS crossbar). Level d of a
            // column depends on levels <= d only, so these 16 levels are exactly the first 16 of the full computation; if the
            // window aligns within 15 edits (at e = 10 % a 64-character window carries ~6) the traceback never looks further.
            uint64_t R = ONES << lane;                                           // R_n[d] = ~0 << d
            if (lane < 16) Rs[n * 16 + lane] = R;
            const int dl = lane & 15;
            for (int col = n - 1; col >= 0; --col) {
                const int c = __builtin_amdgcn_readlane(tfwd, col);
                const uint64_t pm = ~__ballot(prev == c);                       // bit j = 0 <=> p[m-1-j] == t[col]
                const uint64_t old = R;
                const uint64_t oldm1 = ga_row_shr<1>(old, ONES);
                uint64_t y = (old << 1) | pm;                                   // match
                if (dl > 0) y &= (oldm1 << 1) & oldm1;                          // substitution, text-only edit (from level d-1)
                uint64_t up;
                up = ga_row_shr<1>(y, ONES); if (dl >= 1) y &= up << 1;         // pattern-only edit: prefix-AND with shift
                up = ga_row_shr<2>(y, ONES); if (dl >= 2) y &= up << 2;
                up = ga_row_shr<4>(y, ONES); if (dl >= 4) y &= up << 4;
                up = ga_row_shr<8>(y, ONES); if (dl >= 8) y &= up << 8;
                R = y;
                if (lane < 16) Rs[col * 16 + lane] = R;
            }
            uint64_t hit = __ballot(lane < 16 && !((R >> (m - 1)) & 1ull));
            const bool slow = !hit;            // wave-uniform
            if (slow) {
                // SLOW PATH (a window that needs 16..63 edits): all 64 levels, one per lane, scan steps through ds_bpermute;
                // columns go to this wavefront's HBM slab
                R = ONES << lane;
                Rg[n * 64 + lane] = R;
                for (int col = n - 1; col >= 0; --col) {
                    const int c = __builtin_amdgcn_readlane(tfwd, col);
                    const uint64_t pm = ~__ballot(prev == c);
                    const uint64_t old = R;
                    const uint64_t oldm1 = ga_shfl_up(old, 1, lane);
                    uint64_t y = (old << 1) | pm;
                    if (lane > 0) y &= (oldm1 << 1) & oldm1;
#pragma unroll
                    for (int s = 1; s < 64; s <<= 1) {
                        const uint64_t up = ga_shfl_up(y, s, lane);
                        if (lane >= s) y &= up << s;
                    }
                    R = y;
                    Rg[col * 64 + lane] = R;
                }
                hit = __ballot(!((R >> (m - 1)) & 1ull));
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront reads its own slab back below
            }
            auto R_at = [&](int col, int lvl) -> uint64_t { return slow ? Rg[col * 64 + lvl] : Rs[col * 16 + lvl]; };
            // d0 = smallest level whose bit m-1 is clear in column 0
            int d = hit ? (int)__builtin_ctzll(hit) : -1;
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the columns are in LDS (one wavefront: in-order LDS)
            int ca = 0, cb = 0, wn = 0;           // consumed text / pattern characters, ops of this window (uniform)
            auto emit = [&](unsigned char ch) { if (lane == 0) wops[wn] = ch; ++wn; };
            if (d < 0) {   // [spec] no alignment of this window within 63 edits: diagonal steps
                int steps = min(min(m, n), kGaCommit);
                for (; ca < steps; ++ca, ++cb) {
                    const bool eq = __builtin_amdgcn_readlane(pfwd, cb) == __builtin_amdgcn_readlane(tfwd, ca);
                    emit(eq ? 'M' : 'X');
                    dist += eq ? 0 : 1;
                }
            } else {
                for (;;) {
                    // A run of matches in ONE step: lane i tests what the sequential walk would test at (ca+i, cb+i) -- window
                    // limits, equal characters, and bit (m-2-cb-i) of R_{ca+i+1}[d] clear ('M' never changes d) -- and the run
                    // is as long as the leading true lanes. The walk below then takes the one non-match step that ends it.
                    {
                        const int ai = ca + lane, bi = cb + lane;
                        bool cond = bi < m && ai < n && (last || (ai < kGaCommit && bi < kGaCommit));
                        const int aic = cond ? ai : 0, bic = cond ? bi : 0;
                        const uint64_t rr = R_at(aic + 1, d);
                        cond = cond && pwin[bic] == twin[aic] && (bic + 1 >= m || !((rr >> (m - 2 - bic)) & 1ull));
                        const uint64_t bad = ~__ballot(cond);
                        const int run = bad ? (int)__builtin_ctzll(bad) : 64;
                        if (run) {
                            if (lane < run) wops[wn + lane] = 'M';
                            wn += run; ca += run; cb += run;
                        }
                    }
                    if (cb == m) break;
                    if (!last && (ca >= kGaCommit || cb >= kGaCommit)) break;
                    if (ca == n) { emit('D'); ++cb; --d; ++dist; continue; }
                    // the three vectors a step can look at, fetched together (one LDS round trip per step instead of up to four)
                    const int dm1 = d > 0 ? d - 1 : 0;
                    const uint64_t r_next_d = ga_uniform(R_at(ca + 1, d));       // R_{a+1}[d]   : match
                    const uint64_t r_next_dm1 = ga_uniform(R_at(ca + 1, dm1));   // R_{a+1}[d-1] : substitution (b+1), text-only edit (b)
                    const uint64_t r_cur_dm1 = ga_uniform(R_at(ca, dm1));          // R_a[d-1]     : pattern-only edit (b+1)
                    auto clear = [&](uint64_t r, int b) -> bool { return b >= m || !((r >> (m - 1 - b)) & 1ull); };
                    const bool eq = __builtin_amdgcn_readlane(pfwd, cb) == __builtin_amdgcn_readlane(tfwd, ca);
                    if (eq && clear(r_next_d, cb + 1)) { emit('M'); ++ca; ++cb; continue; }
                    if (d > 0 && clear(r_next_dm1, cb + 1)) { emit('X'); ++ca; ++cb; --d; ++dist; continue; }
                    if (d > 0 && clear(r_cur_dm1, cb + 1)) { emit('D'); ++cb; --d; ++dist; continue; }
                    if (d > 0 && clear(r_next_dm1, cb)) { emit('I'); ++ca; --d; ++dist; continue; }
                    status = AIM_PAIR_WFA_NO_LINK;   // cannot happen (the recurrence guarantees one rule applies)
                    break;
                }
            }
            if (BT) {   // the window's ops leave as coalesced byte stores
                __builtin_amdgcn_s_waitcnt(0xC07F);
                for (int i = lane; i < wn; i += kWave)
                    if (nops + i < cap) ops[nops + i] = (char)wops[i];
            }
            nops += wn;
            pi += cb;
            ti += ca;
            if (status != AIM_PAIR_OK) break;
        }
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

inline void genasm_plan(const aim_params_t &p, uint32_t n_pairs, uint32_t *grid, uint32_t *block, size_t *lds)
{
    (void)p;
    *block = kWave;
    *lds = (size_t)(kGaW + 1) * 16 * 8 + 384;
    const uint32_t per_cu = (uint32_t)std::min<size_t>(8, lds_workgroups_per_cu(*lds));
    uint32_t g = 256 * per_cu;
    const uint32_t need = ((n_pairs + 7u) / 8u) * 8u;
    if (g > need) g = need < 8u ? 8u : need;
    *grid = g;
}

inline void genasm_launch(const aim_params_t &p, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s)
{
    if (p.flags & AIM_FLAG_BACKTRACE) hipLaunchKernelGGL((genasm_wave_kernel<true>), dim3(grid), dim3(kWave), lds, s, ka);
    else hipLaunchKernelGGL((genasm_wave_kernel<false>), dim3(grid), dim3(kWave), lds, s, ka);
}

}  // namespace aim
