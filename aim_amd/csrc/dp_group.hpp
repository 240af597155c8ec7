// dp_group.hpp -- NW / SWG on MEDIUM reads (READ_SIZE 177 .. 1024, round 6: score-only to 1536, with CIGAR also 1440 .. 2048 -- dp_group_rs_ok: between the per-lane register kernels of dp_reg.hpp and the multi-wavefront strips
// of dp_strip.hpp): G CONSECUTIVE LANES OWN A PAIR, 64 / G pairs per wavefront, every lane K = 32 consecutive cells of its pair's row in registers as
// packed int16 pairs -- dp_strip.hpp's row body (previous row in registers, two cells per vector instruction, the in-row gap chain as a prefix minimum)
// with the strip boundaries INSIDE the wavefront: what crosses a lane boundary travels by DPP, the prefix minimum over a pair's lanes is ONE wave scan on
// keys that carry the pair's rank in their high bits, and nothing waits for a mailbox. Round 5: the length sweep (profiles/r05/length_sweep.txt) showed
// l = 180 .. 700 at 300 - 1 600 GCUPS (NW) / 180 - 840 (SWG) on nw_lane / swg_lane / single-wavefront strips, against 5 500 at l = 150 and 2 000 - 2 900
// from l = 1000 up.
//
// Same results as nw_compute / swg_compute (NW/DPU-WRAM/dpu/nw.c:109-153, SWG/DPU-WRAM/dpu/swg.c:121-171) by dp_strip.hpp's argument (dp_strip_exact_ok:
// no int16 store of the reference can wrap, so the prefix-minimum form equals its cell-by-cell arithmetic), including the flat table's aliasing for
// plen > tlen: cell (h, W) is the boundary cell of row h + 1 (computed by the lane that owns column W - 1, handed to the pair's lanes by ds_bpermute),
// and the last row's tail cells are walked by the pair's first lane over the row's LDS image. Pairs outside (a sequence empty) go to the
// to-do list: dp_lane.hpp's kernels (READ_SIZE <= 320) or dp_strip_kernel in to-do mode drain it behind this kernel.
#pragma once

#include "aim_device.hpp"
#include "dp_strip.hpp"
#include "wfa_lane.hpp"   // LANE_TODO_*
#include "dp_lane.hpp"    // swg_cell_bytes

namespace aim {

constexpr int kDpgMinRs = 177;
// Packed registers per lane and row (KP; 2 KP columns) and the READ_SIZE range: 16 everywhere up to READ_SIZE 1024, and with CIGAR (the direction bits' lane words are
// dp_traceback_swg_bits' 16-byte layout for 32 columns); round 6: score-only READ_SIZE 1025 .. 1280 / .. 1536 with 20 / 24 registers -- two pairs of <= 32 lanes per
// wavefront where dp_strip ran one wavefront per pair (NW l = 1000: 1 980 GCUPS) -- and with CIGAR READ_SIZE 1440 .. 2048 with ONE pair of 45 .. 64 lanes per wavefront
// (dp_strip's two-wavefront strips with the walk on one of them: 900 / 730 GCUPS at l = 2000).
// Registers per lane and row (KP; 2 KP columns per lane): 16, 20 or 24. A wavefront holds P = floor(64 / G) pairs of G = ceil(READ_SIZE / 2 KP) lanes and a row costs ~KP, so the
// cells per unit of time go as P / KP: READ_SIZE 728 is two pairs of 23 lanes at 16 registers (46 of 64 lanes), three of 19 at 20, FOUR of 16 at 24 -- the dips of round 5's
// length sweep at l = 320 / 700 were idle lanes. The shape with the largest P / KP among those the configuration has registers for: NW score-only 16 / 20 / 24 (156 / 184 / 213
// VGPRs; 28: 241), SWG score-only 16 / 20 / 24 (256 VGPRs and twenty spill instructions inside the row at 24 -- still 2 806 GCUPS at l = 1450 against dp_strip's 2 394), NW with CIGAR 16 / 20
// (the lane words are 16 bytes for 40 columns of NW, 251 VGPRs), SWG with CIGAR 16.
// (... times the wavefronts per CU the pairs' LDS slots leave, up to the eight the registers admit: READ_SIZE 192 as sixteen pairs of four lanes is 23 KB per wavefront = six per CU,
//  and measured 10 % SLOWER than ten pairs of six.)
// LDS of one pair slot: pattern | text | the last row's M, I (int16) | 16 B {M, D of cell W - 1, M of the cell above it}
__host__ __device__ inline int dp_group_slot_bytes(int rs) { return 2 * ((rs + 79) & ~15) + 2 * 2 * ((rs + 47) & ~7) + 16; }
__host__ __device__ inline int dp_group_kp(int read_size, bool bt, bool swg)
{
    const int kmax = bt ? (swg ? 16 : 20) : (swg ? 24 : 28);
    int best = 16;
    long best_num = 0, best_den = 1;
    for (int kp = 16; kp <= kmax; kp += 4) {
        const int g = (read_size + 2 * kp - 1) / (2 * kp);
        if (g > kWave) continue;
        const int p = kWave / g;
        const long lds = (((long)p * dp_group_slot_bytes(read_size) + 15) & ~15L) + (bt ? 64 * 3 * 16 : 0) + 64;
        const long gran = (lds + 1279) / 1280 * 1280;          // (LDS is handed out in 1 280-byte granules: lds_workgroups_per_cu)
        long waves = 160L * 1024 / gran;
        waves = waves > 8 ? 8 : (waves < 1 ? 1 : waves);
        const long num = (long)p * waves, den = kp;
        if (num * best_den > best_num * den) { best = kp; best_num = num; best_den = den; }   // p waves / kp > best
    }
    return best;
}
__host__ __device__ inline bool dp_group_rs_ok(int read_size, bool bt, bool swg)
{
    if (read_size < kDpgMinRs) return false;
    if (!bt) return read_size <= (swg ? 1536 : 1792);         // (two pairs of <= 32 lanes at 20 / 24 / 28 registers; SWG stops at 24)
    return read_size <= 1024 || (!swg && read_size <= 1280) || (read_size >= 1440 && read_size <= (swg ? 2048 : 2560));   // (NW at 20 registers: 40 columns per lane)
}

__host__ __device__ inline int dp_group_lanes(int read_size, bool bt, bool swg) { const int k = 2 * dp_group_kp(read_size, bt, swg); return (read_size + k - 1) / k; }   // G: lanes per pair (6 .. 64)

constexpr int kDpgTileRows = 64;                 // the traceback's window (dp_traceback_swg_bits): 64 rows x 3 lane words x 16 B = 3 KB (128 rows measured slower: 3.53 -> 3.65 ms at NW l = 300, LDS residency at READ_SIZE 192)
__host__ __device__ inline size_t dp_group_lds_bytes(int rs, bool bt, bool swg)
{
    return (((size_t)(kWave / dp_group_lanes(rs, bt, swg)) * (size_t)dp_group_slot_bytes(rs) + 15) & ~(size_t)15) + (bt ? (size_t)kDpgTileRows * 3 * 16 : 0) + 64;
}
// BACKTRACE: a pair's slab of direction bits, dp_strip.hpp's layout -- FLW [READ_SIZE + 3 rows][FS lane words], then the boundary cells' bytes [row]. Lane words (DpBits): SWG 16 bytes
// (32 columns x 4 bits), NW 8 bytes at 32 columns, 16 at 40.
__host__ __device__ inline int dp_group_fs(int rs, bool swg) { return dp_group_lanes(rs, true, swg); }   // (exactly the pair's lanes: a row of the slab is one contiguous run of lane words, rows follow each other without gaps)
__host__ __device__ inline int dp_group_word_bytes(int rs, bool swg) { return (swg || dp_group_kp(rs, true, swg) > 16) ? 16 : 8; }
__host__ __device__ inline size_t dp_group_slab_bytes(int rs, bool swg) { return (((size_t)(rs + 3) * (size_t)dp_group_fs(rs, swg) * (size_t)dp_group_word_bytes(rs, swg) + (size_t)(rs + 3) + 64) + 255) & ~(size_t)255; }

inline bool dp_group_supported(const aim_params_t &p, const Knobs &kn)
{
    if (kn.no_dp_group || kn.force_dpwave || kn.dpw_legacy || kn.strip_k > 0 || kn.dpw_nw > 0) return false;   // (the long-read kernels' own knobs ask for those kernels)
    if (p.algo != AIM_ALGO_NW && p.algo != AIM_ALGO_SWG) return false;
    if (!dp_group_rs_ok(p.read_size, (p.flags & AIM_FLAG_BACKTRACE) != 0, p.algo == AIM_ALGO_SWG)) return false;
    if (p.algo == AIM_ALGO_SWG && swg_cell_bytes(p) == 1) return false;   // int8 cells wrap by design: the literal kernels
    return dp_strip_exact_ok(p, false);
}

template <int ALGO, bool BT, int KP>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void dp_group_kernel(KArgs a, int G)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    constexpr bool SWG = (ALGO == AIM_ALGO_SWG);
    constexpr int K = 2 * KP;
    const int lane = threadIdx.x;
    const int rs = a.p.read_size;
    const int P = kWave / G;                                   // pairs per wavefront
    const int q = lane / G, g = lane - q * G;                  // pair slot, lane of the pair
    const bool lane_on = q < P;                                // (64 - P G lanes idle)
    const int seqcap = (rs + 79) & ~15, rowcap = (rs + 47) & ~7;
    char *slot = smem + (size_t)(lane_on ? q : 0) * (size_t)dp_group_slot_bytes(rs);
    unsigned char *ldsP = reinterpret_cast<unsigned char *>(slot), *ldsT = ldsP + seqcap;
    int16_t *rowM = reinterpret_cast<int16_t *>(ldsT + seqcap), *rowI = rowM + rowcap;
    int *tl = reinterpret_cast<int *>(rowI + rowcap);
    uint32_t *tile = reinterpret_cast<uint32_t *>(smem + (((size_t)P * (size_t)dp_group_slot_bytes(rs) + 15) & ~(size_t)15));   // (BT) the traceback's window
    const int FS = dp_group_fs(rs, SWG);
    const size_t slab = dp_group_slab_bytes(rs, SWG);
    constexpr int NQS = DpBits<K, SWG>::NQS, RSH = DpBits<K, SWG>::RSH, RM = DpBits<K, SWG>::RM;   // (BT) dwords per lane word of direction bits
    char *slab0 = a.scratch + (uint64_t)blockIdx.x * a.scratch_per_wave;
    uint32_t *FLW = reinterpret_cast<uint32_t *>(slab0 + (size_t)(lane_on ? q : 0) * slab);                 // (BT) this pair's direction bits
    unsigned char *BF = reinterpret_cast<unsigned char *>(FLW + (size_t)(rs + 3) * FS * NQS);
    uint32_t *todo = const_cast<uint32_t *>(a.todo);
    const int O = a.p.gap_o, E = a.p.gap_e, OE = O + E, MATCH = a.p.match, MISMATCH = a.p.mismatch;
    const int GD = a.p.gap_d, GI = a.p.gap_i, MAXS = a.p.max_score;
    const int GE = SWG ? E : GD;                               // step of the in-row chain
    const int v0 = 1 + g * K;                                  // first column of this lane
    const uint32_t n_units = (a.n_pairs + (uint32_t)P - 1u) / (uint32_t)P;
    const int rank_hi = lane_on ? (P - 1 - q) : (P + 1);       // scan keys: pairs on the left carry LARGER high bits and lose every minimum

    for (uint32_t it = 0;; ++it) {
        uint32_t unit;
        if (!xcd_unit(n_units, it, &unit)) break;
        const uint32_t pair = unit * (uint32_t)P + (uint32_t)q;
        const bool valid = lane_on && pair < a.n_pairs;
        aim_request_t rq;
        rq.pattern_len = rq.text_len = 0; rq.padding = 0; rq.idx = 0;
        if (valid) rq = load_request(a, pair);
        int plen = rq.pattern_len, tlen = rq.text_len;
        // not this kernel's: an empty sequence, aliasing beyond one row (plen > 2 tlen), lengths beyond the rows (the literal kernels report those)
        const bool outl = valid && (plen < 1 || tlen < 1 || plen > rs || tlen > rs);   // (until round 6 also plen > 2 tlen: the last row's tail cells below take any plen now)
        {
            const unsigned long long m = __ballot(outl && g == 0), below = (1ull << lane) - 1ull;
            if (m) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(&todo[LANE_TODO_COUNT], (uint32_t)__builtin_popcountll(m));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (outl && g == 0) todo[LANE_TODO_LIST + base + (uint32_t)__builtin_popcountll(m & below)] = pair + a.pair_base;
            }
        }
        const bool act = valid && !outl;
        if (!act) { plen = 0; tlen = 0; }
        const int hmax = -wave_min_i32(-tlen);                 // rows of this wavefront (wave-uniform)
        if (hmax == 0) continue;
        const int W = tlen + 1;
        __syncthreads();                                       // (single wavefront: the previous unit's tail walks are done with the slots)
        if (act) {   // sequences into the pair's slot, zero beyond their length (dp_strip.hpp)
            const unsigned char *gP = reinterpret_cast<const unsigned char *>(a.patterns + (uint64_t)pair * rs);
            const unsigned char *gT = reinterpret_cast<const unsigned char *>(a.texts + (uint64_t)pair * rs);
            for (int i = g * 4; i < seqcap; i += G * 4) {
                uint32_t wp = 0, wt = 0;
                if (i + 3 < plen && (rs & 3) == 0) wp = *reinterpret_cast<const uint32_t *>(gP + i);
                else for (int b = 0; b < 4; ++b) if (i + b < plen) wp |= (uint32_t)gP[i + b] << (8 * b);
                if (i + 3 < tlen && (rs & 3) == 0) wt = *reinterpret_cast<const uint32_t *>(gT + i);
                else for (int b = 0; b < 4; ++b) if (i + b < tlen) wt |= (uint32_t)gT[i + b] << (8 * b);
                *reinterpret_cast<uint32_t *>(ldsP + i) = wp;
                *reinterpret_cast<uint32_t *>(ldsT + i) = wt;
            }
        }
        if (BT && act) {
            if (SWG) {   // memset(cigar->operations, 'M', 2*READ_SIZE), swg.c:261; row-init boundary cells {M = I = o + h e, D = MAX_SCORE}: the walk only asks whether column 1's D was extended from them
                uint32_t *o4 = reinterpret_cast<uint32_t *>(a.ops + (uint64_t)pair * 2 * rs);
                for (int w = g; w < (rs >> 1); w += G) o4[w] = 0x4D4D4D4Du;
                for (int h = 1 + g; h <= tlen + 1; h += G) BF[h] = (unsigned char)((O + h * E) + O <= MAXS ? 0 : 16);
            }
        }
        if (BT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the tail owner's stores to the same boundary bytes come later in program order: made explicit, dp_strip.hpp)
        __syncthreads();
        const int Rr = min(plen, W - 1);                       // regular columns 1 .. Rr
        const bool has_tail = act && plen >= W;
        const bool any_tail = __ballot(has_tail) != 0ull;      // wave-uniform
        dps2 Mp[KP], Ip[KP], cD[KP], vmask[KP];
        int nvalid = Rr - v0 + 1;                              // cells of this lane inside the row
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
        uint32_t pc16[KP];                                     // this lane's pattern characters as 16-bit fields
#pragma unroll
        for (int j = 0; j < KP; ++j) pc16[j] = (uint32_t)ldsP[v0 - 1 + 2 * j] | ((uint32_t)ldsP[v0 + 2 * j] << 16);
        const bool tail_owner = has_tail && v0 <= Rr && Rr < v0 + K;   // this lane owns column W - 1 = Rr
        const int tail_t = Rr - v0;                                     // ... as its cell tail_t
        const int own_addr = (q * G + (has_tail ? (Rr - 1) / K : g)) * 4;   // ds_bpermute address of the pair's tail owner (no tail: oneself)
        const int pchW = has_tail ? (int)ldsP[W - 1] : 0;
        int BM = 0, BI = 0, BD = 0, BMprev = 0;                         // B(h), and B(h - 1).M (row 0: 0)
        int nBM = 0, nBI = 0, nBD = 0;                                  // B(h + 1) of a pair with tail cells, as its owner lane computed it
        const dps2 OEp = dps_splat(OE), Ep = dps_splat(E), GIp = dps_splat(GI);
        dps2 c1[KP];                                                    // G = A - c1: SWG (v + 1) e - (o + e); NW v g
#pragma unroll
        for (int j = 0; j < KP; ++j) c1[j] = cD[j] + dps_splat(SWG ? (E - OE) : 0);
        const dps2 costD = dps_splat(MISMATCH - (SWG ? MATCH : 0)), costM = dps_splat(SWG ? MATCH : 0);
        uint32_t ones = 0x00010001u;
        opaque(ones);
        int tailM_M = 0, tail_diag = 0;                                 // the tail owner's M of cell (h, W - 1) and of (h - 1, W - 1)
        int upM_prev = SWG ? O + Rr * E : Rr * GD;                      // M[h - 1][W - 1] (row 0: its initial value)
        auto pick = [&](const dps2 (&arr)[KP], int t) {                 // cell t of a packed register row, t per lane: a binary select tree (dp_strip.hpp)
            uint32_t v[KP];
#pragma unroll
            for (int j = 0; j < KP; ++j) v[j] = dps_bits(arr[j]);
            const int d = t >> 1;
#pragma unroll
            for (int step = 1; step < KP; step <<= 1) {
                const bool odd = (d & step) != 0;
#pragma unroll
                for (int j = 0; j + step < KP; j += 2 * step) v[j] = odd ? v[j + step] : v[j];
            }
            const uint32_t w = v[0];
            return (t & 1) ? (int)(int16_t)(w >> 16) : (int)(int16_t)(w & 0xffffu);
        };

        for (int h = 1; h <= hmax; ++h) {
            const uint32_t tch2 = (uint32_t)ldsT[h - 1] * 0x00010001u;
            // ---- diagonal input of this lane's first cell: M[h-1][v0 - 1] (the lane on the left; the pair's first lane: B(h - 1).M)
            int dfirst = __builtin_amdgcn_update_dpp(0, (int)Mp[KP - 1].y, 0x138, 0xf, 0xf, false);   // wave_shr:1
            if (g == 0) dfirst = BMprev;
            // ---- pre-carry: I, A, G of this lane's K cells, two per instruction
            dps2 A[KP], Iv[KP], Gv[KP];
            uint32_t fw[4] = {0u, 0u, 0u, 0u};                 // (BT) the row's direction bits, assembled as the tests become known (layout: dp_traceback_swg_bits)
            dps2 gmin = dps_splat(kInf16);
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const uint32_t up = dps_bits(Mp[j]);
                const uint32_t prev = j ? dps_bits(Mp[j - 1]) : ((uint32_t)(uint16_t)dfirst << 16);
                const dps2 diag = dps_from(__builtin_amdgcn_alignbit(up, prev, 16));       // {M[v-1], M[v]} of the previous row
                const dps2 f = dps_from(pk_ne01(pc16[j], tch2, ones));
                const dps2 sub = f * costD + (diag + costM);
                dps2 ins;
                if (SWG) {
                    const dps2 insn = Mp[j] + OEp, inse = Ip[j] + Ep;
                    ins = dps_min(insn, inse);
                    if (BT) {   // "I extended": I_up + e < M_up + o + e
                        const uint32_t sI = __builtin_amdgcn_perm(0u, dps_bits(__builtin_elementwise_sub_sat(inse, insn)), 0x09080c0cu);
                        fw[j >> 2] |= sI & (0x10100000u << (j & 3));
                    }
                }
                else ins = Mp[j] + GIp;
                Iv[j] = ins;
                A[j] = dps_min(sub, ins);
                dps2 gg = A[j] - c1[j];
                gg = dps_from((dps_bits(gg) & dps_bits(vmask[j])) | (0x7fff7fffu & ~dps_bits(vmask[j])));
                Gv[j] = gg;
                gmin = dps_min(gmin, gg);
            }
            // ---- exclusive prefix minimum over the lanes of the pair: one wave scan on (rank of the pair, value) keys
            const int lane_min = min((int)gmin.x, (int)gmin.y);
            int total;
            const int key = (rank_hi << 16) | (lane_min + 0x8000);
            const int sk = wave_excl_scan_min(key, lane, &total);
            const int lane_pre = ((sk >> 16) == rank_hi) ? (sk & 0xffff) - 0x8000 : kDpInf;
            // ---- B(h): the boundary cell of this row
            if (h == 1 || !has_tail) {
                if (SWG) { BM = O + h * E; BI = BM; BD = MAXS; }
                else { BM = h * GI; BI = BD = 0; }
            } else { BM = nBM; BI = nBI; BD = nBD; }
            const int carry_in = SWG ? min(BD, BM + O) : BM;   // G[0]
            // ---- post-carry: D / R and M of the K cells; the new row replaces the old one in the registers
            const int pre = min(carry_in, lane_pre);
            dps2 c = dps_splat(pre);
            dps2 Do[KP];
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                dps2 s_; s_.x = kInf16; s_.y = Gv[j].x;
                const dps2 prej = dps_min(c, s_);                      // {pre(2j), pre(2j+1)}
                c = dps_splat(min((int)prej.y, (int)Gv[j].y));
                if (BT && SWG) {   // "the next cell's D was extended": pre(v) < G(v)
                    const uint32_t sD = __builtin_amdgcn_perm(0u, dps_bits(__builtin_elementwise_sub_sat(prej, Gv[j])), 0x0c0c0908u);
                    fw[j >> 2] |= sD & (0x00001010u << (j & 3));
                }
                Do[j] = prej + cD[j];
                Mp[j] = dps_min(A[j], Do[j]);
                if (SWG) Ip[j] = Iv[j];
            }
            BMprev = BM;
            // ---- first tail cell (h, W): the boundary cell of row h + 1 (rows before the last; the last row's tail is walked below)
            // (its D / R is the in-row chain one column further: the prefix minimum over ALL of the owner lane's cells -- cells beyond the row carry +inf -- plus W steps;
            //  one select tree per row, for M[h][W-1], instead of three)
            if (any_tail) {
                const int upM = pick(Mp, tail_t);
                int cM = 0, cI = 0, cDd = 0;
                if (tail_owner && h <= tlen) {
                    const int diag_keep = upM_prev;
                    tailM_M = upM; tail_diag = diag_keep;
                    const int tch = (int)(tch2 & 0xffu);
                    cDd = (int)c.x + W * GE;
                    if (SWG) {
                        cI = min(BM + OE, BI + E);
                        cM = min(diag_keep + ((pchW == tch) ? MATCH : MISMATCH), min(cI, cDd));
                    } else {
                        cI = BM + GI;
                        cM = min(diag_keep + ((pchW == tch) ? 0 : MISMATCH), min(cI, cDd));
                    }
                }
                upM_prev = upM;
                if (BT && tail_owner && h < tlen) {
                    if (SWG) {   // "D extended" of the tail cell = "the next cell's D was extended" of cell W - 1: in this row's bits already
                        const int tj = tail_t >> 1;
                        const uint32_t fq = (tj >> 2) == 0 ? fw[0] : ((tj >> 2) == 1 ? fw[1] : ((tj >> 2) == 2 ? fw[2] : fw[3]));
                        const uint32_t xD = (fq >> (8 * (tail_t & 1) + 4 + (tj & 3))) & 1u;
                        BF[h + 1] = (unsigned char)((cM != cDd ? 1 : 0) | (cM != cI ? 2 : 0) | (xD << 2) | (BI + E < BM + OE ? 8 : 0) | (cM + O <= cDd ? 0 : 16));
                    }
                    else BF[h + 1] = (unsigned char)((cM != cDd ? 1 : 0) | (cM != cI ? 2 : 0));   // NW: "not D", "not I"
                }
                nBM = __builtin_amdgcn_ds_bpermute(own_addr, cM);
                nBI = __builtin_amdgcn_ds_bpermute(own_addr, cI);
                nBD = __builtin_amdgcn_ds_bpermute(own_addr, cDd);
            }
            // ---- (BT) four direction bits per cell (NW: the first two), one 16-byte store per lane and row
            if (BT) {
#pragma unroll
                for (int j = 0; j < KP; ++j) {
                    const uint32_t dA = dps_bits(__builtin_elementwise_sub_sat(A[j], Do[j])), dB = dps_bits(__builtin_elementwise_sub_sat(A[j], Iv[j]));
                    const uint32_t w1 = __builtin_amdgcn_perm(dB, dA, 0x0b0a0908u);
                    fw[j >> RSH] |= w1 & (0x01010101u << (j & RM));
                }
                if (act && h <= tlen && nvalid > 0 && !(a.dbg_flags & 4u)) {
                    if constexpr (NQS == 4) *reinterpret_cast<uint4 *>(FLW + ((size_t)h * FS + g) * 4) = make_uint4(fw[0], fw[1], fw[2], fw[3]);
                    else *reinterpret_cast<uint2 *>(FLW + ((size_t)h * FS + g) * 2) = make_uint2(fw[0], fw[1]);
                }
            }
            // ---- a pair's last row: its regular part into the slot (score; the tail walk reads it)
            if (__ballot(act && h == tlen) != 0ull) {
                if (act && h == tlen) {
#pragma unroll
                    for (int t = 0; t < K; ++t)
                        if (t < nvalid) {
                            rowM[v0 + t] = (t & 1) ? Mp[t >> 1].y : Mp[t >> 1].x;
                            if (SWG) rowI[v0 + t] = (t & 1) ? Ip[t >> 1].y : Ip[t >> 1].x;
                        }
                    if (tail_owner) { tl[0] = tailM_M; tl[1] = pick(Do, tail_t); tl[2] = tail_diag; }
                    if (g == 0) { rowM[0] = (int16_t)BM; if (SWG) rowI[0] = (int16_t)BI; }
                }
            }
        }
        __syncthreads();
        // ---- after the last row: the reference's tail cells v = W .. plen of the LAST row, sequentially, by the pair's first lane (dp_strip.hpp)
        if (act && g == 0) {
            int score;
            if (has_tail) {
                const int h = tlen;
                const int tch = ldsT[h - 1];
                const int bM = rowM[0], bI = SWG ? (int)rowI[0] : 0;
                int upM = tl[0], upD = tl[1];
                int lastM = 0;
                int tw_g = -1, tw_R = -1;         // (BT) lane word being assembled for the tail cells, and its canonical row (dp_strip.hpp: row tlen + v / W, column v mod W)
                uint32_t tw[4] = {0u, 0u, 0u, 0u};
                auto tw_flush = [&]() {
                    if (tw_g >= 0) for (int d = 0; d < NQS; ++d) FLW[((size_t)tw_R * FS + tw_g) * NQS + d] = tw[d];
                };
                int Rt = tlen + 1, C = 0;
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
                    if (BT) {   // the direction bits of this cell: column C = v - W of row tlen + 1 (C = 0: the boundary array), dp_strip.hpp
                        const uint32_t nD = cM != cDd ? 1u : 0u, nI = cM != cI ? 1u : 0u, xD = (SWG && upD + E < upM + OE) ? 1u : 0u, xI = (SWG && leftI + E < leftM + OE) ? 1u : 0u;
                        if (C == 0) BF[Rt] = (unsigned char)(nD | (nI << 1) | (xD << 2) | (xI << 3) | ((!SWG || cM + O <= cDd) ? 0u : 16u));
                        else {
                            if (C >= 2) {   // this cell's "D extended" is kept at the cell on its left
                                const int t = (C - 2) - tw_g * K, j = t >> 1;
                                tw[j >> RSH] |= xD << (8 * (t & 1) + 4 + (j & RM));
                            }
                            const int gg = (C - 1) / K, t = (C - 1) - gg * K, j = t >> 1;
                            if (gg != tw_g || Rt != tw_R) {
                                tw_flush();
                                tw_g = gg; tw_R = Rt; tw[0] = tw[1] = tw[2] = tw[3] = 0u;
                            }
                            tw[j >> RSH] |= (nD << (8 * (t & 1) + (j & RM))) | (nI << (8 * (2 + (t & 1)) + (j & RM))) | (xI << (8 * (2 + (t & 1)) + 4 + (j & RM)));
                        }
                    }
                    if (v < plen) { rowM[v] = (int16_t)cM; if (SWG) rowI[v] = (int16_t)cI; }   // (plen > 2 tlen: read back W cells on, as the cell "above")
                    upM = cM; upD = cDd;
                    lastM = cM;
                }
                if (BT) tw_flush();
                score = lastM;
            } else score = (int)rowM[plen];
            if (BT) tl[3] = score;
            aim_result_t r;
            r.max_operations = plen + tlen;
            r.begin_offset = plen + tlen - 1;
            r.end_offset = plen + tlen;
            r.score = score;
            r.status = AIM_PAIR_OK;
            r.idx = rq.idx;
            if (!BT) store_result(a, pair, r);
        }
        if (BT) {   // the tracebacks: the wavefront walks its pairs one after the other (dp_traceback_swg_bits: 64 cells of a diagonal run per step)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            for (int sq = 0; sq < P; ++sq) {
                const int l0 = sq * G;
                const int s_act = __builtin_amdgcn_readlane((int)act, l0);
                if (!s_act) continue;
                const int s_plen = __builtin_amdgcn_readlane(plen, l0), s_tlen = __builtin_amdgcn_readlane(tlen, l0);
                const uint32_t s_pair = (uint32_t)__builtin_amdgcn_readlane((int)pair, l0), s_idx = (uint32_t)__builtin_amdgcn_readlane((int)rq.idx, l0);
                char *sslot = smem + (size_t)sq * (size_t)dp_group_slot_bytes(rs);
                const unsigned char *sP = reinterpret_cast<const unsigned char *>(sslot), *sT = sP + seqcap;
                const int *stl = reinterpret_cast<const int *>(sslot + 2 * seqcap + 2 * 2 * rowcap);
                const uint32_t *sFLW = reinterpret_cast<const uint32_t *>(slab0 + (size_t)sq * slab);
                const unsigned char *sBF = reinterpret_cast<const unsigned char *>(sFLW + (size_t)(rs + 3) * FS * NQS);
                char *ops = a.ops + (uint64_t)s_pair * 2 * rs;
                int begin_offset = s_plen + s_tlen - 1;
                if (!(a.dbg_flags & 1u))
                    dp_traceback_swg_bits<K, SWG, true>(a.p, s_plen, s_tlen, FS, sFLW, sBF, sP, sT, tile, kDpgTileRows, ops, lane, begin_offset, false, 0, 0);
                if (lane == 0) {
                    aim_result_t r;
                    r.max_operations = s_plen + s_tlen;
                    r.begin_offset = begin_offset;
                    r.end_offset = s_plen + s_tlen;
                    r.score = stl[3];
                    r.status = AIM_PAIR_OK;
                    r.idx = s_idx;
                    store_result(a, s_pair, r);
                }
            }
        }
    }
}

// [to-do region only] one wavefront per workgroup, two per SIMD (the row body holds ~10 arrays of KP registers)
inline bool dp_group_plan(const aim_params_t &p, uint32_t n_pairs, const Knobs &kn, uint32_t *grid, size_t *lds, uint64_t *scratch_per_wg)
{
    const bool bt = (p.flags & AIM_FLAG_BACKTRACE) != 0;
    const bool swg = p.algo == AIM_ALGO_SWG;
    const int G = dp_group_lanes(p.read_size, bt, swg), P = kWave / G;
    *lds = dp_group_lds_bytes(p.read_size, bt, swg);
    *scratch_per_wg = bt ? (uint64_t)P * dp_group_slab_bytes(p.read_size, p.algo == AIM_ALGO_SWG) : 256;
    const uint32_t per_cu = (uint32_t)std::min<size_t>(kn.dpg_per_cu > 0 ? (size_t)kn.dpg_per_cu : 8, lds_workgroups_per_cu(*lds));
    uint32_t g = resident_grid(kn, per_cu);
    const uint32_t n_units = (n_pairs + (uint32_t)P - 1u) / (uint32_t)P;
    const uint32_t need = ((n_units + 7u) / 8u) * 8u;
    if (g > need) g = need < 8u ? 8u : need;
    *grid = g;
    return true;
}

#ifdef AIM_TU_DP_GROUP
void dp_group_launch(const aim_params_t &p, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s)
{
    const bool bt = (p.flags & AIM_FLAG_BACKTRACE) != 0;
    const int G = dp_group_lanes(p.read_size, bt, p.algo == AIM_ALGO_SWG), kp = dp_group_kp(p.read_size, bt, p.algo == AIM_ALGO_SWG);
#define AIM_DPG(ALGO, BT_, KP_) hipLaunchKernelGGL((dp_group_kernel<ALGO, BT_, KP_>), dim3(grid), dim3(kWave), lds, s, ka, G)
    if (p.algo == AIM_ALGO_NW) {
        if (bt && kp == 16) AIM_DPG(AIM_ALGO_NW, true, 16);
        else if (bt) AIM_DPG(AIM_ALGO_NW, true, 20);
        else if (kp == 16) AIM_DPG(AIM_ALGO_NW, false, 16);
        else if (kp == 20) AIM_DPG(AIM_ALGO_NW, false, 20);
        else if (kp == 24) AIM_DPG(AIM_ALGO_NW, false, 24);
        else AIM_DPG(AIM_ALGO_NW, false, 28);
    } else {
        if (bt) AIM_DPG(AIM_ALGO_SWG, true, 16);
        else if (kp == 16) AIM_DPG(AIM_ALGO_SWG, false, 16);
        else if (kp == 20) AIM_DPG(AIM_ALGO_SWG, false, 20);
        else AIM_DPG(AIM_ALGO_SWG, false, 24);
    }
#undef AIM_DPG
}
#else
void dp_group_launch(const aim_params_t &p, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s);
#endif

}  // namespace aim
