// wfa_group.hpp -- WFA / WFA-adaptive for ANY penalties / MAX_SCORE, score-only or with CIGAR: G LANES PER PAIR,
// 64/G pairs per wavefront, everything after the HBM->LDS DMA in LDS and registers.
//
// Same results as affine_wfa_compute (WFA/DPU-WRAM/dpu/wfa.c:342-379, with -DREDUCE wfa.c:69-140).  The static
// kernel (wfa_lane.hpp) covers the reference's default penalties at MAX_SCORE <= 5 (<= 10 score-only); this one covers
// the rest (READ_SIZE <= 16 368, MAX_SCORE <= 4000 as far as LDS admits the shape; beyond that wfa_wave.hpp) at run-time parameters:
//   * a pair is owned by G consecutive lanes (G = 1 .. 64, picked by wfa_group_plan from the LDS one pair's window
//     needs and the residency that leaves); the lanes of a group take the diagonals k = lo+g, lo+g+G, ...
//   * the live window of wavefronts -- M for the last max(x,o+e)+1 scores, I and D for the last e+1 -- lives in
//     LDS as int16, indexed [ring slot][k + MAX_SCORE + 1] (or modulo 128 / 96 in the narrow-window mode, GroupCfg::wlds),
//     so the +/-1 neighbours of affine_wfa_compute_offsets (wfa.c:231-266) are plain LDS reads; scores are counted in units
//     of gcd(x, o+e, e) (GroupCfg::unit), so the null wavefronts between reachable scores are never visited;
//   * with BACKTRACE every computed cell is streamed to the pair's history region in HBM and wfa_group_tb_kernel walks it, one
//     pair per lane (wfa_backtracing.c:210-351);
//   * sequences arrive by LDS-DMA as in wfa_lane.hpp (G < 8) or straight from global memory (G >= 8), are validated
//     (A/C/G/T only) and packed 2 bits per base into LDS; affine_wfa_extend (wfa.c:186-208) compares 32 bases per
//     trip with funnel shifts;
//   * per-pair descriptors (klo, khi, flags) sit in a small LDS ring, so WFA-adaptive's per-pair bounds
//     (wfa.c:96-139) cost nothing when they do not fire.
// Pairs with non-ACGT bytes go to the to-do list drained by wfa_wave_kernel, as in wfa_lane.hpp.
#pragma once

#include <cstdlib>

#include <cstdio>
#include "aim_device.hpp"
#include "wfa_lane.hpp"
#include "wfa_lane_packed.hpp"

#ifndef AIM_GROUP_DIRECT_G
#define AIM_GROUP_DIRECT_G 8      // groups of at least this many lanes pack their sequences straight from global memory (no LDS staging rows)
#endif
// Wavefronts per SIMD the register allocation must leave room for (20 single-wave workgroups per CU = 5 per SIMD = at
// most 96 VGPRs). The score-only variants use 61-68 registers; the CIGAR variants sit at 87-100, right at that step, and one
// register over it cost cfg3 with CIGAR 28 % (4.45 -> 5.71 ms, same box): the bound is stated instead of left to chance.
#ifndef AIM_GROUP_MIN_WAVES
#define AIM_GROUP_MIN_WAVES 5
#endif
#ifndef AIM_GROUP_MAX_PER_CU
#define AIM_GROUP_MAX_PER_CU 20   // cap on resident single-wave workgroups per CU (5 per SIMD at <= 102 VGPRs; 24 measured worse on cfg3 with CIGAR)
#endif

namespace aim {

struct GroupCfg {
    int kbias;        // MAX_SCORE + 1: slot index of diagonal k is k + kbias
    int wcap;         // 2*MAX_SCORE + 3 entries per ring row
    int ring_m;       // rows of the M / descriptor ring: max(x, o+e) + 1
    int ring_e;       // power of two > e
    int np;           // packed dwords per sequence (READ_SIZE/16 rounded up) + 1 pad
    int pair_dwords;  // LDS dwords per pair: window + descriptors + packed sequences (odd => conflict-free across pairs)
    int rows_per_wave;
    // BACKTRACE: every pair of the launch (chunk) owns a history region in HBM that outlives the compute kernel -- the traceback
    // is a kernel of its own (wfa_group_tb_kernel):  [TbHead 16 B][TbRow table, (MAX_SCORE/unit+2) x 8 B][pool of cells {M, I, D, -} int16:
    // the cell of (score s, diagonal k) sits at a CLOSED-FORM index -- narrow rows: s * wlds + home(k) (the LDS row's own image); one home per
    // diagonal: s * s + s + k (row s holds diagonals -s .. s) -- so the traceback needs no descriptor to ADDRESS a cell and fetches the
    // three descriptors and the five candidate cells of a step in ONE round trip (round 3: descriptors first, then cells: two)]
    // [run scratch: (2*MAX_SCORE+16) x 4 B, compact CIGAR only]
    int hist_pair_bytes;  // bytes of one pair's region (multiple of 256)
    int pool_off;         // byte offset of the pool inside the region
    int pool_cap;         // cells {M, I, D, -} (8 B) of the pool: rows x wlds, or rows^2
    int runs_off;         // byte offset of the run scratch
    int runs_cap;         // entries of it
    int wlds;         // entries per ring row IN LDS: wcap (every diagonal has its own home), or a power of two < wcap ("narrow
                      // window": homes are (k + kbias) & wmask; a pair whose wavefront outgrows wlds - 2 diagonals is handed
                      // to the general kernel through the to-do list, like a pair with non-ACGT bytes)
    int wmask;        // wlds - 1 in narrow mode, 0xffff otherwise (REDUCE = false kernels: home = (k + kbias) & wmask)
    int wmagic;       // MODW kernels (G = 16 with the reduction): rows that are no power of two -- home = t - ((t * wmagic) >> 16) * wlds with
                      // t = k + kbias, i.e. t mod wlds for wmagic = ceil(65536 / wlds). Rows of 96 hold cfg3 at 16 workgroups per CU where 128
                      // hold 12. 0 = the other kernels (mask).
    int unit;         // score unit: gcd(x, o+e, e). Only multiples of it have a wavefront (every score is a sum of penalties), so the
                      // score loop counts in units -- row s of the rings / of the history table is score s * unit -- and never
                      // visits the null wavefronts in between (x = 4, o = 6, e = 2: every second step of the reference's loop)
};

enum { GF_PRESENT = 1, GF_MNULL = 2, GF_INULL = 4, GF_DNULL = 8, GF_HASI = 16, GF_HASD = 32 };
constexpr int kGrpNull = -16384;

// Per-pair history (BACKTRACE): head + one 16-byte descriptor per score + the offsets (see GroupCfg).
struct TbHead { int32_t final_score; int32_t walk; int32_t pad[2]; };   // walk: 1 = the traceback kernel owns this pair, 0 = to-do list / not computed
struct TbRow { int16_t klo, khi, flags, pad; };   // the score's final (reduced) bounds and flags: what the traceback's range / null tests read
static_assert(sizeof(TbHead) == 16 && sizeof(TbRow) == 8, "history layout");

// Pool index of the cell of (score su in units, diagonal k): see GroupCfg. The traceback reads neighbours of cells that exist, so k may
// lie one or two outside the row: clamped here (such a cell is never selected -- the caller's range test fails).
template <bool MODW>
__device__ __forceinline__ int group_cell_index(const GroupCfg &c, int su, int k)
{
    if (c.wlds != c.wcap) {   // narrow rows: the row's LDS image
        const int t = max(k + c.kbias, 0);
        const int h = MODW ? t - (int)__umul24(__umul24((uint32_t)t, (uint32_t)c.wmagic) >> 16, (uint32_t)c.wlds) : (t & c.wmask);
        return su * c.wlds + h;
    }
    const int kc = min(max(k, -su), su);
    return su * su + su + kc;
}

// minimum over the G lanes of a group (G-aligned inside a 16-lane DPP row); every lane of the group gets it
template <int G>
__device__ __forceinline__ int group_min(int v)
{
    constexpr int big = 0x7fffffff;
    if (G == 64) return wave_min_i32(v);   // a whole wavefront per pair
    if (G == 32) {                         // two DPP rows per pair: row minimum, then the partner row's through the LDS crossbar
        v = min(v, __builtin_amdgcn_update_dpp(big, v, 0xB1, 0xf, 0xf, false));
        v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x4E, 0xf, 0xf, false));
        v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x141, 0xf, 0xf, false));
        v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x140, 0xf, 0xf, false));
        return min(v, __builtin_amdgcn_ds_bpermute((int)((threadIdx.x ^ 16u) << 2), v));
    }
    if (G >= 2) v = min(v, __builtin_amdgcn_update_dpp(big, v, 0xB1, 0xf, 0xf, false));    // quad_perm [1,0,3,2]
    if (G >= 4) v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x4E, 0xf, 0xf, false));    // quad_perm [2,3,0,1]
    if (G >= 8) v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x141, 0xf, 0xf, false));   // row_half_mirror
    if (G >= 16) v = min(v, __builtin_amdgcn_update_dpp(big, v, 0x140, 0xf, 0xf, false));  // row_mirror
    return v;
}

#ifdef AIM_GROUP_STAMPS   // diagnostic builds only: s_memtime per phase of a score step, summed per wave, dumped behind the to-do region
#define AIM_GSTAMP(i) do { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); \
    __builtin_amdgcn_sched_barrier(0); gsum[i] += t_ - glast; glast = t_; } while (0)
#else
#define AIM_GSTAMP(i) do { } while (0)
#endif
template <int G, bool REDUCE, bool BT, bool MODW = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(AIM_GROUP_MIN_WAVES))) void wfa_group_kernel(KArgs a, GroupCfg c)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    debug_poison_lds(a, smem);
    constexpr int PPW = kWave / G;                       // pairs per wavefront
    const int lane = threadIdx.x;
    const int g = lane % G, q = lane / G;
    const int rs = a.p.read_size;
    // raw rows of one array, whole 16-B chunks; a whole-wavefront group (G == 64) packs straight from global memory
    // instead (one pair per ~2 ms of compute: nothing to hide, and 2*READ_SIZE bytes of LDS buy residency)
    const bool pk = a.packedP != nullptr;               // wave-uniform: the batch arrived packed (2 bits per base): no staging, no validation
    const int rows_dw = (G >= AIM_GROUP_DIRECT_G || pk) ? 0 : ((PPW * rs + 15) / 16) * 4;
    uint32_t *rowsP = reinterpret_cast<uint32_t *>(smem);
    uint32_t *rowsT = rowsP + rows_dw;
    uint32_t *pairmem = rowsT + rows_dw + 1;
    // per-pair region
    uint32_t *mine = pairmem + q * c.pair_dwords;
    int16_t *Mw = reinterpret_cast<int16_t *>(mine);                        // [ring_m][wlds]
    int16_t *Iw = Mw + c.ring_m * c.wlds;                                    // [ring_e][wlds]
    int16_t *Dw = Iw + c.ring_e * c.wlds;                                    // [ring_e][wlds]
    int16_t *meta = Dw + c.ring_e * c.wlds;                                  // [ring_m][4] = klo, khi, flags, pad
    uint32_t *packed = mine + ((c.ring_m + 2 * c.ring_e) * c.wlds * 2 + c.ring_m * 8 + 3) / 4;   // P then T, np dwords each
    uint32_t *pkP = packed, *pkT = packed + c.np;

    const int U = c.unit;                                // scores below are in units of U (GroupCfg::unit)
    const int X = a.p.mismatch / U, OE = (a.p.gap_o + a.p.gap_e) / U, E = a.p.gap_e / U, MS = a.p.max_score;
    const int kb = c.kbias;
    uint32_t *todo = reinterpret_cast<uint32_t *>(a.scratch);
    // BACKTRACE: every wavefront is also streamed to the pair's history region in HBM (GroupCfg); wfa_group_tb_kernel walks it.
    char *hist_base = BT ? a.scratch + a.scratch_per_wave : nullptr;
    const uint32_t n_units = (a.n_pairs + PPW - 1) / PPW;
    const int nchunk_total = (PPW * rs + 15) / 16;       // 16-B chunks per array per unit (the last one may run into the next row / tail slack)

    auto dma = [&](uint32_t unit) {
        if (G >= AIM_GROUP_DIRECT_G || pk) return;
        const uint32_t pair0 = unit * PPW;
        const uint32_t rows = min((uint32_t)PPW, a.n_pairs - pair0);
        const int nchunks = (int)((rows * rs + 15) / 16);
        const char *gp = a.patterns + (uint64_t)pair0 * rs, *gt = a.texts + (uint64_t)pair0 * rs;
        for (int base = 0; base < nchunk_total; base += kWave) {
            const int ch = base + lane;
            if (ch < nchunks) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp + (size_t)ch * 16),
                                                 (__attribute__((address_space(3))) void *)(rowsP + base * 4), 16, 0, AIM_LANE_DMA_AUX);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gt + (size_t)ch * 16),
                                                 (__attribute__((address_space(3))) void *)(rowsT + base * 4), 16, 0, AIM_LANE_DMA_AUX);
            }
        }
    };
    // M / descriptor ring of exactly max(x, o+e) + 1 rows, addressed by row index. The score counter is wave-uniform (all
    // pairs of a wavefront step together), so sm = score % ring_m is a running scalar and the rows of score - x,
    // score - (o+e) and score - e (the only ones ever read; all < ring_m back) are derived from it once per score step.
    // Computing the index at every use instead measured 3-5 % slower on the G <= 16 plans.
    int score = 0, sm = 0, i_x = 0, i_oe = 0, i_e = 0;
    auto back = [&](int d) { const int i = sm - d; return i < 0 ? i + c.ring_m : i; };   // 0 <= d < ring_m
    const int wmask = c.wmask;
    const uint32_t wmagic = (uint32_t)c.wmagic;
    const int wl = c.wlds;
    auto H = [&](int k) {                                                            // home of diagonal k inside a ring row
        const int t = k + kb;                                                        // 0 <= t < wcap
        if constexpr (MODW) return t - (int)__umul24(__umul24((uint32_t)t, wmagic) >> 16, (uint32_t)wl);   // t mod wlds (GroupCfg::wmagic); 24-bit multiplies are full rate, v_mul_lo_u32 is not
        else return t & wmask;
    };
    auto mrow_at = [&](int i) { return Mw + i * c.wlds; };                           // row base: row[H(k)]
    auto meta_at = [&](int i) { return meta + i * 4; };
    auto islot = [&](int s) { return Iw + (s & (c.ring_e - 1)) * c.wlds; };
    auto dslot = [&](int s) { return Dw + (s & (c.ring_e - 1)) * c.wlds; };
    auto fence = [&]() { asm volatile("" ::: "memory"); };   // same-wave LDS traffic is ordered; compiler fence only

#ifdef AIM_GROUP_STAMPS
    unsigned long long gsum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, glast;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(glast) :: "memory");
#endif
    uint32_t unit;
    bool have = xcd_unit(n_units, 0, &unit);
    aim_request_t rq_next;
    rq_next.pattern_len = rq_next.text_len = 0; rq_next.padding = 0; rq_next.idx = 0;
    if (have) {
        dma(unit);
        if (unit * PPW + q < a.n_pairs) rq_next = load_request(a, unit * PPW + q);
    }
    for (uint32_t it = 0; have; ++it) {
        const uint32_t pair = unit * PPW + q;
        const bool active = pair < a.n_pairs;
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        const aim_request_t rq = rq_next;
        // (Tried in round 2 and reverted: for G == 64 every bound / flag is wave-uniform, and moving that bookkeeping to SGPRs
        // with v_readfirstlane cut 28 static VALU instructions per step -- and made cfg3 11 % SLOWER (8.21 -> 9.15 ms, same box):
        // the scalar chain lengthened the wave's critical path more than the freed vector slots were worth.)
        const int plen = rq.pattern_len, tlen = rq.text_len;
        // ---- validate + pack: the G lanes of a group split the packed dwords of their pair ------------------------
        uint32_t bad = 0;
        if (pk) {   // the wire image IS the LDS image: the group's lanes copy the ceil(READ_SIZE/16) dwords of each row
            const int npw = (rs + 15) / 16;
            const uint32_t *gp = a.packedP + (uint64_t)pair * npw, *gt = a.packedT + (uint64_t)pair * npw;
            for (int j = g; j < npw; j += G) {
                pkP[j] = active ? __builtin_nontemporal_load(gp + j) : 0u;
                pkT[j] = active ? __builtin_nontemporal_load(gt + j) : 0u;
            }
            if (g == 0) { pkP[c.np - 1] = 0u; pkT[c.np - 1] = 0u; }
        } else {
            const uint32_t *rp = G >= AIM_GROUP_DIRECT_G ? reinterpret_cast<const uint32_t *>(a.patterns + (uint64_t)pair * rs) : rowsP + (q * rs) / 4;
            const uint32_t *rt = G >= AIM_GROUP_DIRECT_G ? reinterpret_cast<const uint32_t *>(a.texts + (uint64_t)pair * rs) : rowsT + (q * rs) / 4;
            const int npw = (rs + 15) / 16;
            for (int j = g; j < npw; j += G) {
#pragma unroll
                for (int side = 0; side < 2; ++side) {
                    const uint32_t *r = side ? rt : rp;
                    const int len = side ? tlen : plen;
                    uint32_t out = 0;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int w = 4 * j + i;
                        const uint32_t av = (4 * w < rs && (G < AIM_GROUP_DIRECT_G || active)) ? r[w] : 0u;
                        const uint32_t t = (av >> 1) & 0x03030303u;
                        const uint32_t rec = __builtin_amdgcn_perm(0u, 0x47544341u, t);
                        const int rem = len - 4 * w;
                        const uint32_t mask = rem >= 4 ? ~0u : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
                        bad |= (rec ^ av) & mask;
                        out |= __builtin_amdgcn_udot4(t, 0x40100401u, 0u, false) << (8 * i);
                    }
                    (side ? pkT : pkP)[j] = out;
                }
            }
            if (g == 0) { pkP[c.np - 1] = 0u; pkT[c.np - 1] = 0u; }
        }
        // group-wide "bad": OR over the G lanes
        {
            const unsigned long long bm = __ballot(bad != 0u);
            const unsigned long long gm = (G == 64) ? ~0ull : (((1ull << G) - 1ull) << (q * G));
            bad = (bm & gm) ? 1u : 0u;
        }
        // the raw image is consumed: next unit's DMA flies under the compute
        uint32_t nunit = 0;
        const bool nhave = xcd_unit(n_units, it + 1, &nunit);
        __builtin_amdgcn_s_waitcnt(0xC07F);
        __syncthreads();
        if (nhave) {
            dma(nunit);
            if (nunit * PPW + q < a.n_pairs) rq_next = load_request(a, nunit * PPW + q);
        }
        __builtin_amdgcn_sched_barrier(0);

#ifdef AIM_GROUP_COUNT_TRIPS    // diagnostic builds only (tools/group_trips.py): what the WAVEFRONT executes -- score steps, trips of the k-loop, trips of
        // extend's while loop -- counted once per trip by whichever lane is the first active one, summed over the lanes at the end and
        // reported through the result of the wavefront's first pair (max_operations = k-trips, begin_offset = extend trips, end_offset = score steps)
        int dbg_kt = 0, dbg_et = 0, dbg_st = 0;
#define AIM_GTRIP(c) do { const unsigned long long ex_ = __ballot(1); if (lane == __ffsll((long long)ex_) - 1) ++(c); } while (0)
#else
#define AIM_GTRIP(c) do { } while (0)
#endif
        // ---- affine_wfa_extend on packed words (wfa.c:186-208) -----------------------------------------------------
        auto extend = [&](int k, int off) -> int {
            int v = off - k, h = off;
            if (off < 0 || v < 0) return off;
            int rem = min(plen - v, tlen - h);
            while (rem > 0) {
                AIM_GTRIP(dbg_et);
                // 32 bases per trip: the trip count of a wavefront is its longest run, and every pair has one diagonal on
                // which runs average 1/e bases (same box, 16 -> 32 bases: l=100 e=5 % +3.3 %, l=150 e=2 % +6 %, cfg3 +2-3 %).
                // Word wp + 2 may lie one dword past the pair's packed image (the next array, or the workgroup's 64-B LDS
                // slack): whatever it holds sits >= 128 - v bases ahead, beyond `rem`.
                const int wp = v >> 4, wt = h >> 4;
                const uint32_t sp = (uint32_t)((v & 15) * 2), st = (uint32_t)((h & 15) * 2);
                const uint32_t p0 = pkP[wp], p1 = pkP[wp + 1], p2 = pkP[wp + 2];
                const uint32_t t0 = pkT[wt], t1 = pkT[wt + 1], t2 = pkT[wt + 2];
                const uint32_t xlo = __builtin_amdgcn_alignbit(p1, p0, sp) ^ __builtin_amdgcn_alignbit(t1, t0, st);
                const uint32_t xhi = __builtin_amdgcn_alignbit(p2, p1, sp) ^ __builtin_amdgcn_alignbit(t2, t1, st);
                const int n = xlo ? (__builtin_ctz(xlo) >> 1) : (xhi ? 16 + (__builtin_ctz(xhi) >> 1) : 32);
                const int adv = min(n, rem);       // stop at the first mismatch or at the end of either sequence
                h += adv; v += adv; rem -= adv;
                if (adv < 32) break;
            }
            return h;
        };

        const int ak = tlen - plen;
        score = 0; sm = 0; i_x = i_oe = i_e = 0;
        int final_score = -1;
        bool done = !active || bad != 0u;
#ifdef AIM_GROUP_COUNT_WIDTHS   // diagnostic builds only: wavefront width statistics into the result (max_operations = sum of widths, begin_offset = steps with width > 32, end_offset = steps with width > 64)
        int dbg_wsum = 0, dbg_w32 = 0, dbg_w64 = 0;
#endif
        // wavefronts[0]: lo = hi = 0, M[0] = 0 (wfa.c:347-348)
        int klo = 0, khi = 0, flags = GF_PRESENT | GF_INULL | GF_DNULL;
        // Every later wavefront is extended by the lane that computes it (below); score 0 has no compute step.
        int part = 0x7fffffff;                     // min over my diagonals of the distance to the end (for the reduction)
        int dist0 = 0x7fffffff, dist1 = 0x7fffffff, dist2 = 0x7fffffff; // that distance on my first, second and third diagonal of the current row
        // history region of this pair (BACKTRACE): descriptor table + pool
        char *hreg = BT ? hist_base + (size_t)(active ? pair : 0u) * (size_t)c.hist_pair_bytes : nullptr;
        TbRow *htab = reinterpret_cast<TbRow *>(hreg + sizeof(TbHead));
        uint2 *hpool = reinterpret_cast<uint2 *>(hreg + c.pool_off);   // cells {M, I, D, -}: ONE 8-byte store per computed cell, at its closed-form index (GroupCfg)
        const bool hnarrow = c.wlds != c.wcap;
        if (g == 0) {
            const int m00 = done ? 0 : extend(0, 0);
            mrow_at(0)[H(0)] = (int16_t)m00;
            if (BT && !done) hpool[hnarrow ? H(0) : 0] = make_uint2((uint32_t)(uint16_t)m00, 0u);
            meta[0] = 0; meta[1] = 0; meta[2] = (int16_t)flags;
        }
        if (BT && a.cig == nullptr && active && bad == 0u) {   // memset(cigar->operations, 'M', 2*READ_SIZE), wfa.c:465 (ops-row output only)
            uint4 *orow = reinterpret_cast<uint4 *>(a.ops + (uint64_t)pair * 2 * rs);
            const uint4 mm = make_uint4(0x4D4D4D4Du, 0x4D4D4D4Du, 0x4D4D4D4Du, 0x4D4D4D4Du);
            // (only the pieces that can hold a printed operation: begin_offset >= min(plen, tlen) - MAX_SCORE / e, wfa_lane.hpp)
            // (gap_e == 0: gaps cost nothing to extend and MAX_SCORE bounds no length -- the whole row is written)
            const int p_lo = a.p.gap_e > 0 ? max(0, min(plen, tlen) - a.p.max_score / a.p.gap_e) >> 4 : 0, p_hi = min((plen + tlen + 15) >> 4, (2 * rs) / 16);
            for (int j = p_lo + g; j < p_hi; j += G) orow[j] = mm;
        }
        fence();
        AIM_GSTAMP(0);   // staging, pack, pair setup
        for (;;) {
            AIM_GSTAMP(5);   // loop back-edge
            if (!done) {
                const bool live = (flags & GF_PRESENT) && !(flags & GF_MNULL);
                int16_t *mrow = mrow_at(sm);               // already extended by the lanes that produced it
                if (REDUCE && live && (khi - klo + 1) >= 10) {   // affine_wfa_reduce_wvs, wfa.c:69-140
                    // the group's lanes split the diagonals. `part` (min distance over my diagonals) came with the row;
                    // one pass finds my lowest and highest diagonal within 50 of the best. The reference's two scans --
                    // first such k in [klo, top_limit) else top_limit; last such k in (bottom_limit, khi] else
                    // bottom_limit -- are min(top_limit, lowest) and max(bottom_limit, highest) over the whole row.
                    const int mind = min(max(plen, tlen), group_min<G>(part));
                    int kfirst = 0x7fffffff, klast = -0x7fffffff;
                    // A live row was computed by the previous score step over the same [klo, khi] with the same lanes: the distances of
                    // my first three diagonals are still in registers (dist0 .. dist2; round 6: the third -- 58 % of cfg3's steps have a pair wider than 2 G); only rows wider than 3 G are read back from LDS.
                    {
                        const int k0 = klo + g, k1 = k0 + G, k2 = k1 + G;
                        const bool c0 = k0 <= khi && dist0 - mind <= 50, c1 = k1 <= khi && dist1 - mind <= 50, c2 = k2 <= khi && dist2 - mind <= 50;
                        kfirst = c0 ? k0 : (c1 ? k1 : (c2 ? k2 : kfirst));
                        klast = c2 ? k2 : (c1 ? k1 : (c0 ? k0 : klast));
                    }
                    for (int k = klo + g + 3 * G; k <= khi; k += G) {
                        const int off = mrow[H(k)];
                        if ((max(plen - (off - k), tlen - off) - mind) <= 50) { kfirst = min(kfirst, k); klast = max(klast, k); }
                    }
                    kfirst = group_min<G>(kfirst);
                    klast = -group_min<G>(-klast);
                    int nklo = klo, nkhi = khi;
                    const int top_limit = min(ak - 1, khi);
                    if (klo < top_limit) nklo = min(top_limit, kfirst);
                    const int bottom_limit = max(ak + 1, nklo);
                    if (khi > bottom_limit) nkhi = max(bottom_limit, klast);
                    if (nklo > nkhi) flags |= GF_MNULL | GF_INULL | GF_DNULL;
                    else { klo = nklo; khi = nkhi; }
                    if (g == 0) {
                        int16_t *me = meta_at(sm);
                        me[0] = (int16_t)klo; me[1] = (int16_t)khi; me[2] = (int16_t)flags;
                    }
                    fence();
                }
                if (BT && g == 0)   // final descriptor of this score (after reduction): one 8-byte store
                    *reinterpret_cast<uint2 *>(htab + score) = make_uint2((uint32_t)(uint16_t)klo | ((uint32_t)(uint16_t)khi << 16), (uint32_t)(uint16_t)flags);
                // affine_wfa_end_reached, wfa.c:210-230
                if ((flags & GF_PRESENT) && !(flags & GF_MNULL) && klo <= ak && khi >= ak && (int)mrow[H(ak)] >= tlen) {
                    done = true;
                    final_score = score * U;
                } else if ((score + 1) * U > MS) {   // wfa.c:368-376: the reference steps through the null wavefronts up to MAX_SCORE and leaves with MAX_SCORE + 1
                    done = true;
                    final_score = MS + 1;
                }
            }
            AIM_GSTAMP(1);   // reduce + descriptors + end test
            if (__ballot(!done) == 0ull) break;
            AIM_GTRIP(dbg_st);
            ++score;
            sm = sm + 1 == c.ring_m ? 0 : sm + 1;
            i_x = back(X); i_oe = back(OE); i_e = back(E);
#if defined(AIM_GROUP_PAD_VALU) || defined(AIM_GROUP_PAD_SALU)
            {   // diagnostic only: N independent ALU ops per score step, to tell issue-bound from stall-bound
#ifdef AIM_GROUP_PAD_VALU
                int padv[8] = {g, g, g, g, g, g, g, g};      // 8 accumulators round-robin: consecutive ops are independent
#pragma unroll
                for (int q = 0; q < AIM_GROUP_PAD_VALU; ++q) asm volatile("v_add_u32 %0, %0, 1" : "+v"(padv[q & 7]));
#endif
#ifdef AIM_GROUP_PAD_SALU
                int pads[8] = {score, score, score, score, score, score, score, score};
#pragma unroll
                for (int q = 0; q < AIM_GROUP_PAD_SALU; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(pads[q & 7]));
#endif
            }
#endif
            if (!done) {
                // ---- affine_wfa_compute_next, wfa.c:268-340 --------------------------------------------------------
                const int s_sub = score - X, s_o = score - OE, s_e = score - E;
                int sub_f = 0, o_f = 0, e_f = 0, sub_lo = 1, sub_hi = -1, o_lo = 1, o_hi = -1, e_lo = 1, e_hi = -1;
                if (s_sub >= 0) { const int16_t *m = meta_at(i_x); sub_lo = m[0]; sub_hi = m[1]; sub_f = m[2]; }
                if (s_o >= 0) { const int16_t *m = meta_at(i_oe); o_lo = m[0]; o_hi = m[1]; o_f = m[2]; }
                if (s_e >= 0) { const int16_t *m = meta_at(i_e); e_lo = m[0]; e_hi = m[1]; e_f = m[2]; }
                const bool m_sub_null = (s_sub < 0) || !(sub_f & GF_PRESENT) || (sub_f & GF_MNULL);
                const bool m_o_null = (s_o < 0) || !(o_f & GF_PRESENT) || (o_f & GF_MNULL);
                const bool i_e_null = (s_e < 0) || !(e_f & GF_PRESENT) || !(e_f & GF_HASI) || (e_f & GF_INULL);
                const bool d_e_null = (s_e < 0) || !(e_f & GF_PRESENT) || !(e_f & GF_HASD) || (e_f & GF_DNULL);
                const bool i_out_null = m_o_null && i_e_null, d_out_null = m_o_null && d_e_null;
                AIM_GSTAMP(2);   // score++, source descriptors
                if (m_sub_null && i_out_null && d_out_null) {
                    flags = 0; klo = 0; khi = -1;
                } else {
                    if (m_sub_null) { sub_lo = 1; sub_hi = -1; }
                    if (m_o_null) { o_lo = 1; o_hi = -1; }
                    if (i_e_null && d_e_null) { e_lo = 1; e_hi = -1; }
                    const int lo = min(min(sub_lo, o_lo), e_lo) - 1;
                    const int hi = max(max(sub_hi, o_hi), e_hi) + 1;
                    flags = GF_PRESENT | (i_out_null ? GF_INULL : GF_HASI) | (d_out_null ? GF_DNULL : GF_HASD);
                    klo = lo; khi = hi;
                    int hi_run = hi;                   // the cells this step computes: [lo, hi_run]
                    if (hi - lo + 1 > c.wlds - 2) {   // narrow window outgrown (never true in linear mode): the general kernel takes the pair
                        bad = 1u;
                        done = true;
                        hi_run = lo - 1;               // nothing more is computed or stored for it (its history pool is sized for admitted widths only)
                    }
                    uint2 *hrow = hpool + (hnarrow ? score * wl : score * score + score);   // BACKTRACE: this score's row of the pool (narrow: cell of k at hrow[H(k)], else hrow[k])
#ifdef AIM_GROUP_COUNT_WIDTHS
                    dbg_wsum += hi - lo + 1; dbg_w32 += (hi - lo + 1) > 32; dbg_w64 += (hi - lo + 1) > 64;
#endif
                    const int16_t *r_ms = mrow_at(i_x), *r_mo = mrow_at(i_oe);   // valid rows even when the score does not exist
                    const int16_t *r_ie = islot(s_e < 0 ? 0 : s_e), *r_de = dslot(s_e < 0 ? 0 : s_e);
                    int16_t *om = mrow_at(sm), *oi = islot(score), *od = dslot(score);
                    part = 0x7fffffff;
                    int trip = 0;
                    // What an ABSENT component reads as is -10 (the reference's un-computed default), what an out-of-range fetch reads as is NULL:
                    // when a component is absent every fetch that feeds it is out of range (its sources are null: ranges 1 .. -1), so the
                    // "out of range" value of those fetches is made -10 once per step instead of one more select per cell and component (round 6).
                    const int nul_i = i_out_null ? -10 : kGrpNull, nul_d = d_out_null ? -10 : kGrpNull, nul_s = m_sub_null ? -10 : kGrpNull;
                    // homes advance with k: one conditional subtraction per step instead of a modulo per address (3 per trip)
                    // (MODW: the unsigned minimum of h and h -+ wl is the one that lies in [0, wl): two instructions, no compare + select)
                    auto wrap_up = [&](int h_) { if constexpr (MODW) return (int)min((uint32_t)h_, (uint32_t)(h_ - wl)); else return h_ & wmask; };
                    auto wrap_dn = [&](int h_) { if constexpr (MODW) return (int)min((uint32_t)h_, (uint32_t)(h_ + wl)); else return h_ & wmask; };
                    int hk = H(lo + g);
                    for (int k = lo + g; k <= hi_run; k += G, ++trip, hk = wrap_up(hk + G)) {   // affine_wfa_compute_offsets, wfa.c:231-266, + affine_wfa_extend
                        // The five source cells are fetched TOGETHER and unconditionally (every row is a valid LDS row of
                        // wcap = 2*MAX_SCORE+3 entries and |k +- 1| <= MAX_SCORE+1, so the addresses are always in bounds);
                        // AFFINE_WAVEFRONT_COND_FETCH's range / null tests then select. Guarded reads compiled to one
                        // exec-masked branch and one LDS round trip EACH: five dependent round trips per cell. (Round 4: I and D are
                        // COMPUTED unconditionally too and only their stores are predicated -- as `if (!i_out_null) { ... }` blocks the
                        // compiler sank two of the five loads into the blocks, i.e. back into a second round trip.)
                        AIM_GTRIP(dbg_kt);
                        const int hkm = wrap_dn(hk - 1), hkp = wrap_up(hk + 1);
                        const int raw_mo_m1 = r_mo[hkm], raw_ie_m1 = r_ie[hkm], raw_mo_p1 = r_mo[hkp], raw_de_p1 = r_de[hkp],
                                  raw_ms = r_ms[hk];
                        const int ins_g = (!m_o_null && o_lo <= k - 1 && k - 1 <= o_hi) ? raw_mo_m1 : kGrpNull;
                        const int ins_i = (!i_e_null && e_lo <= k - 1 && k - 1 <= e_hi) ? raw_ie_m1 : kGrpNull;
                        // (offsets are <= READ_SIZE <= 16 368 or kGrpNull: the reference's int16 store of offset + 1 never wraps, so no cast is spelled out)
                        const int ins = (ins_g == kGrpNull && ins_i == kGrpNull) ? nul_i : max(ins_g, ins_i) + 1;   // i_out_null: both are NULL -> -10
                        if (!i_out_null) oi[hk] = (int16_t)ins;
                        const int del_g = (!m_o_null && o_lo <= k + 1 && k + 1 <= o_hi) ? raw_mo_p1 : nul_d;
                        const int del_d = (!d_e_null && e_lo <= k + 1 && k + 1 <= e_hi) ? raw_de_p1 : nul_d;
                        const int del = max(del_g, del_d);                                                            // d_out_null: both are nul_d = -10
                        if (!d_out_null) od[hk] = (int16_t)del;
                        const int sub = (sub_lo <= k && k <= sub_hi) ? raw_ms + 1 : nul_s;                            // m_sub_null: range 1 .. -1 -> -10
                        // M[s][k] as the reference stores it (int16), then affine_wfa_extend (wfa.c:186-208) on that value: a
                        // diagonal's extension depends on nothing but its own offset, so it is applied before the one store
                        const int ext = extend(k, max(del, max(sub, ins)));
                        om[hk] = (int16_t)ext;
                        if (BT) hrow[hnarrow ? hk : k] = make_uint2((uint32_t)(uint16_t)ext | ((uint32_t)(uint16_t)ins << 16), (uint32_t)(uint16_t)del);   // I / D: -10 when absent (never selected)
                        const int dist = max(plen - (ext - k), tlen - ext);
                        part = min(part, dist);
                        if (REDUCE) { dist0 = trip == 0 ? dist : dist0; dist1 = trip == 1 ? dist : dist1; dist2 = trip == 2 ? dist : dist2; }
                    }
                }
                AIM_GSTAMP(3);   // compute + extend loop
                if (g == 0) {
                    int16_t *me = meta_at(sm);
                    me[0] = (int16_t)klo; me[1] = (int16_t)khi; me[2] = (int16_t)flags;
                }
            }
            fence();
            AIM_GSTAMP(4);   // descriptor store
        }
        AIM_GSTAMP(6);   // exit
        if (active && g == 0) {
            if (BT) {   // hand the pair to wfa_group_tb_kernel (walk = 0: the to-do list's kernel aligns it instead)
                TbHead *hd = reinterpret_cast<TbHead *>(hreg);
                hd->final_score = final_score;
                hd->walk = bad == 0u ? 1 : 0;
            }
            if (bad != 0u) {
                const uint32_t slot = atomicAdd(&todo[LANE_TODO_COUNT], 1u);
                todo[LANE_TODO_LIST + slot] = pair + a.pair_base;
            } else if (!BT) {
                aim_result_t r;
                r.max_operations = plen + tlen;
                r.begin_offset = plen + tlen - 1;
                r.end_offset = plen + tlen;
                r.score = final_score;
                r.status = AIM_PAIR_OK;
                r.idx = rq.idx;
#ifdef AIM_GROUP_COUNT_WIDTHS
                r.max_operations = dbg_wsum; r.begin_offset = dbg_w32; r.end_offset = dbg_w64;
#endif
#ifdef AIM_GROUP_COUNT_TRIPS
                r.max_operations = r.begin_offset = r.end_offset = 0;
#endif
                store_result(a, pair, r);
            }
        }
#ifdef AIM_GROUP_COUNT_TRIPS
        if (!BT) {
            int kt = dbg_kt, et = dbg_et, st = dbg_st;
            for (int o = 32; o; o >>= 1) { kt += __shfl_xor(kt, o); et += __shfl_xor(et, o); st += __shfl_xor(st, o); }
            __builtin_amdgcn_s_waitcnt(0);   // (after the pairs' own result stores)
            if (lane == 0 && unit * PPW < a.n_pairs && a.res != nullptr && !(a.p.flags & AIM_FLAG_RES8)) {
                aim_result_t *r0 = a.res + (size_t)unit * PPW;
                r0->max_operations = kt; r0->begin_offset = et; r0->end_offset = st;
            }
        }
#endif
        have = nhave;
        unit = nunit;
        AIM_GSTAMP(7);   // backtrace + result
    }
#ifdef AIM_GROUP_STAMPS
    if (lane == 0) {
        unsigned long long *dbg = reinterpret_cast<unsigned long long *>(a.scratch + 256) + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 8; ++i) dbg[i] = gsum[i];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// affine_wavefronts_backtrace (wfa_backtracing.c:210-351) as a kernel of its own: ONE PAIR PER LANE over the history regions
// the compute kernel left in HBM. (Round 2 walked inside the compute kernel: one lane of a group walked while the others
// idled, and the walk's registers held the CIGAR variants at 87-100 VGPRs.) Every address of a step depends on (score, k)
// only -- cells sit at closed-form indices (GroupCfg) -- so the three descriptors AND the five candidate offsets come back in ONE
// round trip (round 3 needed the descriptors to address the cells: two dependent trips per step of a walk that waits 93 % of its
// cycles), and the range tests of the reference then select -- a cell outside its row is read at a clamped index and never selected. Output: result_t +
// edit operations patched into the ops row the compute kernel pre-filled with 'M' (default ABI), or aim_cigar_t + runs
// (RUNS: the compact CIGAR; runs are collected backwards in the pair's own run scratch and copied out forwards).
template <bool MODW, typename Sink>
__device__ __forceinline__ int group_tb_walk(const GroupCfg &c, const TbRow *tab, const int16_t *pool, int final_score, int plen, int tlen, int X, int OE, int E, Sink &sink)
{
    enum { BT_M = 0, BT_I = 1, BT_D = 2 };
    const int ak = tlen - plen;
    int status = AIM_PAIR_OK;
    auto valid_loc = [&](int kk_, int off_) {
        const int v_ = off_ - kk_, h_ = off_;
        return v_ > 0 && v_ <= plen && h_ > 0 && h_ <= tlen;
    };
    struct Row { int klo, khi, f; };
    auto row = [&](int s_) {
        const uint2 q = *reinterpret_cast<const uint2 *>(tab + s_);
        Row r;
        r.klo = (int16_t)(q.x & 0xffffu); r.khi = (int16_t)(q.x >> 16); r.f = (int)(q.y & 0xffffu);
        return r;
    };
    auto cell = [&](int s_, int which, int k_) -> int {   // which: 0 = M, 1 = I, 2 = D; index clamped into the row, selected by the caller's range test
        return pool[4 * group_cell_index<MODW>(c, s_, k_) + which];   // cells are {M, I, D, -} int16
    };
    int sc = final_score, k = ak;
    int offset = cell(sc, 0, k);
    bool valid = valid_loc(k, offset);
    int bt = BT_M;
    int v = offset - k, h = offset;
    while (v > 0 && h > 0 && sc > 0) {
        if (!valid) {
            valid = valid_loc(k, offset);
            if (valid) {   // add_trailing_gap, wfa_backtracing.c:48-69
                if (k < ak) for (int i = k; i < ak; ++i) sink.put('I');
                else if (k > ak) for (int i = ak; i < k; ++i) sink.put('D');
            }
        }
        const int s_o = sc - OE, s_e = sc - E, s_x = sc - X;
        const int c_o = max(s_o, 0), c_e = max(s_e, 0), c_x = max(s_x, 0);
        const Row ro = row(c_o), re = row(c_e), rx = row(c_x);   // these eight loads depend on (sc, k) only: one round trip
        const int v_de = cell(c_e, 2, k + 1), v_do = cell(c_o, 0, k + 1);
        const int v_ie = cell(c_e, 1, k - 1), v_io = cell(c_o, 0, k - 1), v_mx = cell(c_x, 0, k);
        int o_lo = ro.klo, o_hi = ro.khi, o_f = ro.f, e_lo = re.klo, e_hi = re.khi, e_f = re.f, x_lo = rx.klo, x_hi = rx.khi, x_f = rx.f;
        if (s_o < 0) { o_lo = 1; o_hi = -1; o_f = 0; }
        if (s_e < 0) { e_lo = 1; e_hi = -1; e_f = 0; }
        if (s_x < 0 || bt != BT_M) { x_lo = 1; x_hi = -1; x_f = 0; }
        int del_ext = kGrpNull, del_open = kGrpNull, ins_ext = kGrpNull, ins_open = kGrpNull, misms = kGrpNull;
        if (bt != BT_I) {
            if ((e_f & GF_PRESENT) && !(e_f & GF_DNULL) && e_lo <= k + 1 && k + 1 <= e_hi) del_ext = v_de;
            if ((o_f & GF_PRESENT) && o_lo <= k + 1 && k + 1 <= o_hi) del_open = v_do;
        }
        if (bt != BT_D) {
            if ((e_f & GF_PRESENT) && (e_f & GF_HASI) && e_lo <= k - 1 && k - 1 <= e_hi) ins_ext = (int16_t)(v_ie + 1);
            if ((o_f & GF_PRESENT) && o_lo <= k - 1 && k - 1 <= o_hi) ins_open = (int16_t)(v_io + 1);
        }
        if (bt == BT_M) {
            if ((x_f & GF_PRESENT) && x_lo <= k && k <= x_hi) misms = (int16_t)(v_mx + 1);
        }
        const int max_all = max(misms, max(max(ins_ext, ins_open), max(del_ext, del_open)));
        if (bt == BT_M) {
            const int num_matches = offset - max_all;
            if (num_matches > 0) sink.matches(num_matches);
            offset = max_all;
            v = offset - k;
            h = offset;
            if (v <= 0 || h <= 0) break;
        }
        char op;
        if (max_all == del_ext) { op = 'D'; sc = s_e; ++k; bt = BT_D; }
        else if (max_all == del_open) { op = 'D'; sc = s_o; ++k; bt = BT_M; }
        else if (max_all == ins_ext) { op = 'I'; sc = s_e; --k; offset = (int16_t)(offset - 1); bt = BT_I; }
        else if (max_all == ins_open) { op = 'I'; sc = s_o; --k; offset = (int16_t)(offset - 1); bt = BT_M; }
        else if (max_all == misms) { op = 'X'; sc = s_x; offset = (int16_t)(offset - 1); }
        else { status = AIM_PAIR_WFA_NO_LINK; break; }
        if (valid) sink.put(op);
        v = offset - k;
        h = offset;
    }
    if (status == AIM_PAIR_OK) {
        if (sc == 0) {
            if (offset > 0) sink.matches(offset);
        } else {
            for (; v > 0; --v) sink.put('D');
            for (; h > 0; --h) sink.put('I');
        }
    }
    return status;
}

template <bool RUNS, bool MODW>
__global__ __launch_bounds__(64) void wfa_group_tb_kernel(KArgs a, GroupCfg c)
{
    const int lane = threadIdx.x;
    const uint32_t pair = blockIdx.x * kWave + lane;
    const bool in_batch = pair < a.n_pairs;
    const int rs = a.p.read_size;
    const int U = c.unit;                                // the history table is indexed in score units (GroupCfg::unit)
    const int X = a.p.mismatch / U, OE = (a.p.gap_o + a.p.gap_e) / U, E = a.p.gap_e / U, MS = a.p.max_score;
    char *hreg = a.scratch + a.scratch_per_wave + (size_t)(in_batch ? pair : 0u) * (size_t)c.hist_pair_bytes;
    const TbHead hd = *reinterpret_cast<const TbHead *>(hreg);
    const bool active = in_batch && hd.walk == 1;        // pairs on the to-do list belong to the general kernel
    const TbRow *tab = reinterpret_cast<const TbRow *>(hreg + sizeof(TbHead));
    const int16_t *pool = reinterpret_cast<const int16_t *>(hreg + c.pool_off);
    aim_request_t rq;
    rq.pattern_len = rq.text_len = 0; rq.padding = 0; rq.idx = 0;
    if (active) rq = load_request(a, pair);
    const int plen = rq.pattern_len, tlen = rq.text_len;
    const int final_score = hd.final_score;
    const bool walk = active && final_score <= MS;      // beyond MAX_SCORE the reference returns without a backtrace (wfa.c:368-376, MRAM variant)
    int status = AIM_PAIR_OK;
    if constexpr (RUNS) {
        RunCollector<1> coll(plen + tlen - 1, reinterpret_cast<uint32_t *>(hreg + c.runs_off), c.runs_cap);
        if (walk) status = group_tb_walk<MODW>(c, tab, pool, final_score / U, plen, tlen, X, OE, E, coll);
        coll.flush();
        if (coll.n == 0) {   // nothing inside [0, end): edit_cigar_print still prints operations[begin_offset] = 'M'
            coll.cur_op = (uint32_t)'M'; coll.cur_len = 1u;
            coll.flush();
        }
        store_cigar(a, pair, active, rq.idx, final_score, status, coll, 0u, lane);
    } else {
        OpsSink sink;
        sink.ops = a.ops + (uint64_t)(active ? pair : 0u) * 2 * rs;
        sink.cap = 2 * rs;
        sink.pos = plen + tlen - 1;                       // edit_cigar_allocate, wfa.c:57-67
        if (walk) {
            status = group_tb_walk<MODW>(c, tab, pool, final_score / U, plen, tlen, X, OE, E, sink);
            if (status == AIM_PAIR_OK) ++sink.pos;
        }
        if (active) {
            aim_result_t r;
            r.max_operations = plen + tlen;
            r.begin_offset = sink.pos;
            r.end_offset = plen + tlen;
            r.score = final_score;
            r.status = status;
            r.idx = rq.idx;
            store_result(a, pair, r);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// hist_pair_bytes: bytes of ONE pair's history region (BACKTRACE; 0 otherwise) -- the caller sizes launches (chunks of pairs)
// so that their regions fit its scratch bound. packed: the batch arrives packed (no staging rows in LDS).
inline bool wfa_group_plan_rows(const aim_params_t &p, uint32_t n_pairs, const Knobs &kn, bool packed, int rows, GroupCfg *c, int *G, uint32_t *grid,
                                size_t *lds, size_t *hist_pair_bytes)   // rows: entries per LDS ring row asked for; < 0 = the default rule
{
    if (p.algo != AIM_ALGO_WFA) return false;
    // int16 offsets with NULL = -16384 (offset + 1 must stay above it) and 24-bit home arithmetic bound the shapes; what really decides is LDS below:
    // the packed image (READ_SIZE / 2 bytes per pair) and, without the reduction, rows of 2 * MAX_SCORE + 3 entries. (Rounds 1-3 stopped at READ_SIZE
    // 2048 / MAX_SCORE 400: WFA-adaptive l = 10 000 e = 1 % then ran one pair per wavefront on wfa_wave_kernel, ~11x off cfg3's per-cell rate.)
    if (p.read_size > 16368 || p.max_score > 4000) return false;
    const int R = p.mismatch > p.gap_o + p.gap_e ? p.mismatch : p.gap_o + p.gap_e;
    int ring_m = 1, ring_e = 1;
    ring_m = R + 1;
    while (ring_e <= p.gap_e) ring_e *= 2;
    if (ring_m > 32 || ring_e > 16) return false;
    {
        auto gcd = [](int a_, int b_) { while (b_) { const int t = a_ % b_; a_ = b_; b_ = t; } return a_; };
        const int u = gcd(gcd(p.mismatch, p.gap_o + p.gap_e), p.gap_e);
        c->unit = (u > 1 && !kn.group_unit1) ? u : 1;
    }
    c->kbias = p.max_score + 1;
    c->wcap = 2 * p.max_score + 3;
    c->ring_m = ring_m;
    c->ring_e = ring_e;
    c->np = (p.read_size + 15) / 16 + 1;
    // Ring rows in LDS. Every diagonal having its own home costs 2*MAX_SCORE+3 entries per row although WFA-adaptive keeps
    // the wavefront narrow (measured, tools/group_widths.py: l = 1000 e = 5 %: mean width 23, 20.6 % of the score steps wider
    // than 32, 0.1 % wider than 64, MAX_SCORE 250 -> 503 homes). From 256 homes up the rows are 128 entries addressed
    // modulo 128 instead: 3.2 KB per pair instead of 10.7 KB at cfg3, i.e. 2-4x the pairs resident per CU -- which is what
    // this latency-bound kernel lacked. A wavefront that outgrows 126 diagonals sends its pair to the general kernel.
    c->wlds = c->wcap;
    c->wmask = 0xffff;
    c->wmagic = 0;
    {
        // (same-box probe, tools/group_policy2.py, score-only kernel ms, row of 128 vs one home per diagonal at the better G:
        // l=1000 e=5% 3.76 vs 6.06; l=400 e=10% 1.57 vs 2.29; l=500 e=5% 1.89 vs 2.12; l=250 e=10% 1.87 vs 2.18; l=1000 e=2% 1.63
        // vs 1.63; no pair of these sets outgrew the row. Rows of 64 cost 0.3-47 % of the pairs a detour, rows of 32 most.)
        int narrow = (p.flags & AIM_FLAG_REDUCE) && c->wcap >= 192 ? 128 : 0;   // without the reduction widths grow with the score
        if (rows >= 0) narrow = rows;
        const bool pow2 = (narrow & (narrow - 1)) == 0;
        if (narrow >= 16 && narrow < c->wcap) {
            if (pow2) { c->wlds = narrow; c->wmask = narrow - 1; }
            else if (narrow % 8 == 0 && (p.flags & AIM_FLAG_REDUCE)) {   // MODW kernels; the caller checks that the plan comes out at G = 16
                const int magic = (65536 + narrow - 1) / narrow;
                bool exact = true;                           // (t * magic) >> 16 == t / narrow for every home index the kernel forms
                for (int t = 0; t <= c->wcap + 2 && exact; ++t) exact = (int)(((uint32_t)t * (uint32_t)magic) >> 16) == t / narrow;
                if (exact) { c->wlds = narrow; c->wmask = 0; c->wmagic = magic; }
            }
        }
    }
    int dw = ((ring_m + 2 * ring_e) * c->wlds * 2 + ring_m * 8 + 3) / 4 + 2 * c->np;
    dw |= 1;
    c->pair_dwords = dw;
    // LDS budget for the wavefront windows of one wavefront's pairs: 12 KiB measured best (occupancy beats lanes per
    // pair: 24 KiB -15..30 %, 48 KiB -45 %); wider budgets only when 16 lanes per pair do not fit otherwise, and only
    // while they still leave 6 workgroups per CU (below)
    int g = 32;
    for (size_t cap_kb : {12, 24, 48}) {
        size_t cap_bytes = cap_kb * 1024;
        // narrow rows (3.2 KB per pair): FOUR pairs per wavefront. (Round 2 measured two better -- with the traceback inside the
        // kernel; with the traceback a kernel of its own the step is VALU-issue-bound and 16 lanes per pair waste fewer of them:
        // l=1000 e=5 % 3.20 -> 2.97 ms score-only, 3.99 -> 3.55 with CIGAR; l=1000 e=2 %, l=500 e=5 %, l=400 / 250 e=10 %: -0 .. -12 %.)
        if (cap_kb == 12 && c->wlds != c->wcap) cap_bytes = 13 * 1024;
        if (kn.group_lds_kb >= 0) cap_bytes = (size_t)kn.group_lds_kb * 1024;
        g = 1;
        while (g <= 32 && (size_t)(kWave / g) * dw * 4 > cap_bytes) g *= 2;
        if (g <= 32) break;
    }
    if (g > 32) {   // a whole wavefront per pair (long reads / large MAX_SCORE)
        if ((size_t)dw * 4 > 48 * 1024) return false;
        g = 64;
    } else {
        // A G <= 16 plan that LDS holds to fewer than 6 workgroups per CU loses to a wavefront per pair at up to 16
        // per CU. Measured (round 2 probe, since removed; tools/group_policy2.py is its successor; score-only, G64/G16 pairs/s, each plan at its real LDS fit): 3 per CU
        // 1.56x (l=1000 e=5%), 4 per CU 1.55x (l=400 e=10%); 6 per CU 1.07x / 0.91x / 0.79x; 11 per CU 0.57x; 16 per CU
        // 0.48x. 5 per CU is not measured.
        const size_t stage = (g >= AIM_GROUP_DIRECT_G || packed) ? 8 : (size_t)2 * ((((size_t)(kWave / g) * p.read_size + 15) / 16) * 16) + 8;   // G >= 32 packs from global memory
        const size_t wg = stage + (size_t)(kWave / g) * dw * 4 + 64;
        if (lds_workgroups_per_cu(wg) < 6 && (size_t)dw * 4 <= 48 * 1024) g = 64;
    }
    if (kn.group_g >= 0) {   // experiments: force the lanes per pair if the plan is feasible at all
        const int fg = kn.group_g;
        if ((fg == 1 || fg == 2 || fg == 4 || fg == 8 || fg == 16 || fg == 32 || fg == 64) && (size_t)(kWave / fg) * dw * 4 <= 48 * 1024) g = fg;
    }
    *G = g;
    c->rows_per_wave = kWave / g;
    const size_t rows_bytes = (g >= AIM_GROUP_DIRECT_G || packed) ? 8 : (size_t)2 * ((((size_t)(kWave / g) * p.read_size + 15) / 16) * 16) + 8;
    *lds = rows_bytes + (size_t)(kWave / g) * dw * 4 + 64;
    if (*lds > 64 * 1024) return false;   // beyond the dynamic-LDS limit of a plain launch (only a forced AIM_GROUP_G gets here)
    const size_t lds_fit = lds_workgroups_per_cu(*lds);
    // Every plan takes what really fits, up to 16 single-wave workgroups per CU. An earlier cap of 8 for G <= 16 predated
    // the fused score step and the correct LDS granule; sweep on one box (ms at 8 -> best per CU): l=100 e=2% 0.671 -> 0.553
    // (11), e=5% 3.85 -> 2.82 (14), e=10% 6.17 -> 4.10 (16), l=250 e=5% 4.23 -> 3.53 (11), l=150 e=2% 1.94 -> 1.28 (16),
    // l=100 e=5% CIGAR 2.29 -> 1.73 (14); G = 64 (cfg3): 11 / 12 / 13 / 14 per CU = 7.81 / 7.10 / 7.42 / 7.06 ms.
    // Above 16 per CU the count is kept a multiple of the CU's 4 SIMDs (an odd wavefront makes one SIMD the straggler: l = 100
    // e = 10 % 18 -> 16 per CU 3.96 -> 3.89 ms, l = 150 e = 2 % 21 -> 20 1.174 -> 1.089 ms); the score-only variants (<= 68 VGPRs, 7
    // wavefronts per SIMD) may use 24 (cfg3 score-only 20 / 22 / 24 / 25 per CU: 3.32 / 3.25 / 3.18 / 3.67 ms), the CIGAR
    // variants are register-bound at 5 per SIMD.
    const size_t cap_per_cu = (p.flags & AIM_FLAG_BACKTRACE) ? AIM_GROUP_MAX_PER_CU : AIM_GROUP_MAX_PER_CU + 4;   // CIGAR variants: 82-96 VGPRs (history addressing) = 5 per SIMD
    uint32_t per_cu = (uint32_t)std::min<size_t>(cap_per_cu, lds_fit);
    if (per_cu > 16) per_cu &= ~3u;
    if (kn.group_per_cu >= 0) per_cu = (uint32_t)std::min<size_t>((size_t)std::max(1, kn.group_per_cu), lds_fit);   // residency sweeps
    const uint32_t n_units = (n_pairs + (kWave / g) - 1) / (kWave / g);
    uint32_t gr = resident_grid(kn, per_cu);
    const uint32_t need = ((n_units + 7u) / 8u) * 8u;
    if (gr > need) gr = need < 8u ? 8u : need;
    *grid = gr;
    if (kn.plan_debug)
        fprintf(stderr, "[aim plan] wfa_group G=%d ring_m=%d ring_e=%d wcap=%d pair_lds=%d B wg_lds=%zu B lds_fit=%zu per_cu=%u grid=%u wlds=%d unit=%d\n", g, ring_m, ring_e,
                c->wcap, dw * 4, *lds, lds_fit, per_cu, gr, c->wlds, c->unit);
    // BACKTRACE: the per-pair history region (GroupCfg). One row of the pool per score (in units): narrow rows -- the image of the LDS
    // row (wlds cells; a pair whose wavefront outgrows it leaves for the to-do list before anything of the offending score is stored);
    // one home per diagonal -- row s holds the 2s+1 diagonals a wavefront of score s can reach at most.
    {
        const uint64_t rows = (uint64_t)p.max_score / (uint64_t)c->unit + 2;
        const uint64_t cells = c->wlds != c->wcap ? rows * (uint64_t)c->wlds : rows * rows;
        if (cells * 8 > (1ull << 30)) return false;
        c->pool_off = (int)(sizeof(TbHead) + (size_t)rows * sizeof(TbRow));
        c->pool_cap = (int)cells;                             // cells of 8 bytes {M, I, D, -}
        c->runs_off = (c->pool_off + c->pool_cap * 8 + 15) & ~15;
        c->runs_cap = 2 * p.max_score + 16;
        c->hist_pair_bytes = (c->runs_off + c->runs_cap * 4 + 255) & ~255;
    }
    *hist_pair_bytes = (p.flags & AIM_FLAG_BACKTRACE) ? (size_t)c->hist_pair_bytes : 0;
    return true;
}

// Rows of the LDS rings. One home per diagonal up to MAX_SCORE 94; beyond that, with the reduction, rows of 128 addressed modulo 128 --
// or of 96 where the launcher's MAX_SCORE says the error rate is 5 % or less (4 * MAX_SCORE <= READ_SIZE): at 16 workgroups per CU instead
// of 12 the latency-bound score step has a third more wavefronts to hide behind. Same box, kernel ms, rows of 128 -> 96, pairs leaving
// for the general kernel in brackets: l=1000 e=5 % 3.09 -> 2.82 (0), with CIGAR 3.63 -> 3.41; l=500 e=5 % 1.54 -> 1.41 (0); l=1000 e=2 %
// 1.21 -> 1.10 (0); but l=400 e=10 % 1.34 -> 1.73 (57 of 32 768) and l=250 e=10 % 1.59 -> 1.82 (38 of 65 536): every pair that outgrows
// its row costs a whole general-kernel pass, so wide wavefronts keep 128. (Rows of 80: 42 pairs of cfg3 leave, 3.50 ms; tools/group_rows.py.)
inline bool wfa_group_plan(const aim_params_t &p, uint32_t n_pairs, const Knobs &kn, bool packed, GroupCfg *c, int *G, uint32_t *grid, size_t *lds,
                           size_t *hist_pair_bytes)
{
    int rows = kn.group_wlds;
    if (rows < 0 && (p.flags & AIM_FLAG_REDUCE) && 2 * p.max_score + 3 >= 192 && 4 * p.max_score <= p.read_size) rows = 96;
    if (rows >= 16 && (rows & (rows - 1)) != 0) {   // no power of two: only the G = 16 kernels address such rows
        Knobs quiet = kn;
        quiet.plan_debug = 0;
        if (!wfa_group_plan_rows(p, n_pairs, quiet, packed, rows, c, G, grid, lds, hist_pair_bytes) || (*G != 16 && !(*G == 8 && kn.group_g == 8)) || c->wmagic == 0) rows = -1;
    }
    return wfa_group_plan_rows(p, n_pairs, kn, packed, rows, c, G, grid, lds, hist_pair_bytes);
}

// Kernels are instantiated in ONE translation unit (tu_*.hip defines AIM_TU_WFA_GROUP); every other includer sees the declaration only.
#ifdef AIM_TU_WFA_GROUP
void wfa_group_tb_launch(const aim_params_t &p, const GroupCfg &c, uint32_t n_pairs, const KArgs &ka, hipStream_t s)
{
    (void)p;
    const uint32_t grid = (n_pairs + kWave - 1) / kWave;
    if (c.wmagic) {
        if (ka.cig) hipLaunchKernelGGL((wfa_group_tb_kernel<true, true>), dim3(grid), dim3(kWave), 0, s, ka, c);
        else hipLaunchKernelGGL((wfa_group_tb_kernel<false, true>), dim3(grid), dim3(kWave), 0, s, ka, c);
    } else {
        if (ka.cig) hipLaunchKernelGGL((wfa_group_tb_kernel<true, false>), dim3(grid), dim3(kWave), 0, s, ka, c);
        else hipLaunchKernelGGL((wfa_group_tb_kernel<false, false>), dim3(grid), dim3(kWave), 0, s, ka, c);
    }
}
#else
void wfa_group_tb_launch(const aim_params_t &p, const GroupCfg &c, uint32_t n_pairs, const KArgs &ka, hipStream_t s);
#endif

// Kernels are instantiated in ONE translation unit (tu_*.hip defines AIM_TU_WFA_GROUP); every other includer sees the declaration only.
#ifdef AIM_TU_WFA_GROUP
void wfa_group_launch(const aim_params_t &p, int G, const GroupCfg &c, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s)
{
    const bool red = p.flags & AIM_FLAG_REDUCE, bt = p.flags & AIM_FLAG_BACKTRACE;
#define AIM_GRP(GG)                                                                                                     \
    do {                                                                                                                \
        if (red && bt) hipLaunchKernelGGL((wfa_group_kernel<GG, true, true>), dim3(grid), dim3(kWave), lds, s, ka, c);  \
        else if (red) hipLaunchKernelGGL((wfa_group_kernel<GG, true, false>), dim3(grid), dim3(kWave), lds, s, ka, c);  \
        else if (bt) hipLaunchKernelGGL((wfa_group_kernel<GG, false, true>), dim3(grid), dim3(kWave), lds, s, ka, c);   \
        else hipLaunchKernelGGL((wfa_group_kernel<GG, false, false>), dim3(grid), dim3(kWave), lds, s, ka, c);          \
    } while (0)
    if (c.wmagic) {   // rows that are no power of two (the planner admits them at G = 16 with the reduction only; G = 8 when forced: VERDICT r04 item 4a's experiment)
        if (G == 8) {
            if (bt) hipLaunchKernelGGL((wfa_group_kernel<8, true, true, true>), dim3(grid), dim3(kWave), lds, s, ka, c);
            else hipLaunchKernelGGL((wfa_group_kernel<8, true, false, true>), dim3(grid), dim3(kWave), lds, s, ka, c);
            return;
        }
        if (bt) hipLaunchKernelGGL((wfa_group_kernel<16, true, true, true>), dim3(grid), dim3(kWave), lds, s, ka, c);
        else hipLaunchKernelGGL((wfa_group_kernel<16, true, false, true>), dim3(grid), dim3(kWave), lds, s, ka, c);
        return;
    }
    switch (G) {
    case 1: AIM_GRP(1); break;
    case 2: AIM_GRP(2); break;
    case 4: AIM_GRP(4); break;
    case 8: AIM_GRP(8); break;
    case 16: AIM_GRP(16); break;
    case 32: AIM_GRP(32); break;
    case 64: AIM_GRP(64); break;
    default: break;
    }
#undef AIM_GRP
}
#else
void wfa_group_launch(const aim_params_t &p, int G, const GroupCfg &c, uint32_t grid, size_t lds, const KArgs &ka, hipStream_t s);
#endif

}  // namespace aim
