// aim_device.hpp -- shared device-side definitions for the gfx950 alignment kernels.
// CDNA4 only: 64-lane wavefronts are assumed everywhere.
#pragma once
#include <algorithm>
#include <cstddef>

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "aim_hip.h"

namespace aim {

constexpr int kWave = 64;

// Kernel argument block (plain data, passed by value).
struct KArgs {
    aim_params_t p;
    uint32_t n_pairs;
    const aim_request_t *req;
    const char *patterns;
    const char *texts;
    aim_result_t *res;
    char *ops;            // [n][2*read_size] or nullptr
    char *scratch;        // per-wave scratch base
    uint64_t scratch_per_wave;  // bytes
    uint32_t pool_cap;    // int16 entries per wave (WFA)
    uint32_t meta_cap;    // WfMeta entries per wave (WFA)
    uint32_t ring_slots;  // WFA wave kernel: LDS offset ring, slots (0 = none) ...
    uint32_t slot_w;      // ... and diagonals per slot
    const uint32_t *todo; // nullptr: units are pairs 0..n_pairs-1; else {count @0, pair ids @16..} written by wfa_lane
    uint32_t dbg_poison_lds;    // debugging aid (AIM_DEBUG_POISON_LDS): 0 = off, else 0x100 | byte every workgroup fills its
    uint32_t dbg_lds_bytes;     // dynamic LDS with at kernel entry (results must not depend on it)
    uint32_t dbg_flags;         // AIM_DEBUG_FLAGS. Bit 2 (results stay right): nw_reg leaves pairs with tail cells to the to-do list. DIAGNOSTIC BUILDS ONLY (-DAIM_DIAG_BUILD, results are
                                // WRONG): 1 = dp_strip / dp_group / nw_reg / swg_reg skip the traceback walk; 4 = dp_group / nw_reg / swg_reg do not store their direction bits (and do not walk)
    // Fused batch I/O (round 3; aim_hip.h "pipelined batches"): kernels that can, consume the packed image of a batch
    // directly and emit the compact CIGAR themselves -- no unpack pass, no ops rows, no run-length pass.
    const uint32_t *packedP;    // [n][ceil(read_size/16)] dwords, 2 bits per base, or nullptr (ASCII rows in patterns / texts)
    const uint32_t *packedT;
    aim_cigar_t *cig;           // compact CIGAR headers [n], or nullptr (results in res, ops rows in ops)
    uint32_t *runs;             // shared run buffer: (length << 8) | op
    uint32_t runs_cap;          // capacity of runs[], in runs
    uint32_t *cursor;           // next free run (bump allocated with atomics)
    uint32_t pair_base;         // the launch covers pairs [pair_base, pair_base + n_pairs) of the batch (chunked launches): pair ids
                                // appended to a to-do list are batch-relative
};

// Request / result access for both wire layouts (aim_hip.h: AIM_FLAG_REQ8 / AIM_FLAG_RES8). The flag tests are wave-uniform.
__device__ __forceinline__ aim_request_t load_request(const KArgs &a, uint32_t pair)
{
    if (a.p.flags & AIM_FLAG_REQ8) {
        const aim_request8_t q = reinterpret_cast<const aim_request8_t *>(a.req)[pair];
        aim_request_t r;
        r.pattern_len = q.pattern_len; r.text_len = q.text_len; r.padding = 0; r.idx = q.idx;
        return r;
    }
    return a.req[pair];
}
__device__ __forceinline__ void store_result(const KArgs &a, uint32_t pair, const aim_result_t &r)
{
    if (a.p.flags & AIM_FLAG_RES8) {
        aim_result8_t q;
        // {idx, score} has no status field: a pair that stopped with a status (only AIM_PAIR_NOMEM can occur without BACKTRACE,
        // and the plans make it impossible -- see make_plan, score-only ring) reports AIM_SCORE_FAILED instead of a partial score
        q.idx = r.idx; q.score = r.status == AIM_PAIR_OK ? r.score : AIM_SCORE_FAILED;
        reinterpret_cast<aim_result8_t *>(a.res)[pair] = q;
    } else {
        a.res[pair] = r;
    }
}

// Debugging aid shared by all kernels: fill the workgroup's dynamic LDS with a chosen byte before anything else runs.
// A kernel whose results depend on what LDS held at entry changes its output under this switch (tools/soak_dp_wave.py).
__device__ __forceinline__ void debug_poison_lds(const KArgs &a, char *smem)
{
    if (a.dbg_poison_lds) {   // wave-uniform, false in production
        const uint32_t w = (a.dbg_poison_lds & 0xffu) * 0x01010101u;
        uint32_t *s4 = reinterpret_cast<uint32_t *>(smem);
        for (uint32_t i = threadIdx.x; i < a.dbg_lds_bytes / 4; i += blockDim.x) s4[i] = w;
        __syncthreads();
    }
}

// Run-time knobs (the AIM_* environment variables). They are experiment / debugging switches, not configuration: every
// one of them leaves results bit-identical. They are read in ONE place (read_knobs, aim_capi.hip) into this struct --
// once per aim_set_configure (frozen in the set together with the plan) and once per stateless entry point -- and the
// planners below only ever see the struct, so plan selection cannot change between a set's configure and its launches.
struct Knobs {
    double scratch_gb = -1.0;     // AIM_SCRATCH_GB      upper bound on plan scratch (default: 3/4 of free device memory)
    bool force_wave = false;      // AIM_FORCE_WAVE=1    WFA: general one-pair-per-wavefront kernel only
    bool no_group = false;        // AIM_NO_GROUP=1      WFA: skip wfa_group_kernel
    bool no_lane_ext = false;     // AIM_NO_LANE_EXT=1   WFA: only the (3,4,1) MAX_SCORE<=5 lane shapes (round-1 behaviour)
    bool no_lane_pk = false;      // AIM_NO_LANE_PK=1    packed batches: always unpack to ASCII rows first (round-2 behaviour)
    bool group_overlap = false;   // AIM_GROUP_OVERLAP=1 wfa_group with CIGAR: chunked launches, traceback kernel of chunk c on a second stream under chunk c+1's compute
    bool wfa_no_ring = false;     // AIM_WFA_NO_RING=1   wfa_wave: no LDS offset ring
    int wfa_slotw = -1;           // AIM_WFA_SLOTW       wfa_wave: diagonals per ring slot
    bool force_dpwave = false;    // AIM_FORCE_DPWAVE=1  NW/SWG: row-scan kernel also for short reads
    bool dpw_legacy = false;      // AIM_DPW_LEGACY=1    NW/SWG long reads: round 2's row-scan dp_wave_kernel instead of the strip pipeline
    int strip_k = -1;             // AIM_STRIP_K         dp_strip: cells per lane (16 or 32)
    int dpw_nw = -1;              // AIM_DPW_NW          dp_wave: wavefronts per pair (implies the row-scan kernel)
    int dpl_no_reg = 0;           // AIM_DPL_NO_REG      dp_lane: 1 = never keep the pattern row in registers (A/B)
    int dpl_seq_lds = -1;         // AIM_DPL_SEQ_LDS     dp_lane: where the pattern row lives -- 0 global memory, 1 LDS image, 2 registers (READ_SIZE <= 124); unset: the measured default
    int dpl_per_cu = -1;          // AIM_DPL_PER_CU      dp_lane: residency sweep
    bool no_nw_reg = false;       // AIM_NO_NW_REG=1     NW short reads: nw_lane_kernel only (rows in LDS), no nw_reg_kernel (row in registers) in front
    bool no_lane = false;         // AIM_NO_LANE=1       WFA: no one-pair-per-lane kernel (wfa_lane / wfa_lane_packed): the group kernel takes their shapes (A/B runs)
    int dbg_flags = 0;            // AIM_DEBUG_FLAGS     diagnostic timing runs (KArgs::dbg_flags)
    bool no_swg_reg = false;      // AIM_NO_SWG_REG=1    SWG short reads: swg_lane_kernel only, no swg_reg_kernel (rows in registers) in front
    bool no_dp_group = false;     // AIM_NO_DP_GROUP=1   NW / SWG medium reads (READ_SIZE 177 .. 1024): nw_lane / swg_lane / dp_strip only, no dp_group_kernel (G lanes per pair) in front
    int dpg_per_cu = -1;          // AIM_DPG_PER_CU=n    dp_group_kernel: wavefronts per CU (default 8; experiments)
    int nw_reg_per_cu = -1;       // AIM_NW_REG_PER_CU   nw_reg: residency sweep
    int group_lds_kb = -1;        // AIM_GROUP_LDS_KB    wfa_group: LDS budget for the windows of one wavefront's pairs
    int group_g = -1;             // AIM_GROUP_G         wfa_group: lanes per pair
    int group_per_cu = -1;        // AIM_GROUP_PER_CU    wfa_group: residency sweep
    int ga_per_cu = 0;            // AIM_GA_PER_CU       genasm: wavefronts per CU (residency sweeps; bounded by LDS)
    int group_unit1 = 0;          // AIM_GROUP_UNIT1     wfa_group: step through every score like the reference (A/B of GroupCfg::unit)
    int group_wlds = -1;          // AIM_GROUP_WLDS      wfa_group: entries per LDS ring row (power of two = narrow window, 0 = one home per diagonal)
    int poison_ops = -1;          // AIM_DEBUG_POISON_OPS      fill the ops rows with this byte before every launch (the kernels write ops[begin_offset, end_offset) only)
    int poison_scratch = -1;      // AIM_DEBUG_POISON_SCRATCH  fill scratch with this byte at configure
    int poison_lds = -1;          // AIM_DEBUG_POISON_LDS      fill dynamic LDS with this byte at kernel entry
    bool plan_debug = false;      // AIM_PLAN_DEBUG=1    print the chosen plan to stderr
    // Not a knob but the one fact about the chip every plan needs: compute units of the device the plan is made for
    // (hipDeviceAttributeMultiprocessorCount, read once per device by chip_cus() in aim_capi.hip: 256 on a whole MI355X, 128 / 64 / 32
    // in DPX / QPX / CPX partitions). AIM_CHIP_CUS overrides it (CPU tests plan for devices this host does not have).
    uint32_t cus = 256;
};

// Persistent grid = what is resident: `per_cu` workgroups on each of the device's compute units, rounded up to the multiple of 8
// xcd_unit() needs.
inline uint32_t resident_grid(const Knobs &kn, uint32_t per_cu)
{
    const uint64_t g = (uint64_t)(kn.cus ? kn.cus : 256u) * per_cu;
    return (uint32_t)std::min<uint64_t>((g + 7u) & ~7ull, 1u << 20);
}

// XCD-aware work distribution: workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8 labels the XCD group), so give each group one contiguous slice
// of the batch: neighbouring pairs share 128-B lines and then share an L2.
// Requires gridDim.x % 8 == 0.  Returns false when this (block, iteration) has no work.
// The 8 is the slice count, not an assumption about the device: on a partition with X = 4 / 2 / 1 XCDs workgroup b runs on
// XCD b % X, and since X divides 8 the slices b % 8 == c and c + X, ... land on the same XCD -- every XCD still owns whole
// contiguous slices. What DOES depend on the device is the resident grid size: resident_grid() above.
__device__ __forceinline__ bool xcd_unit(uint32_t n_units, uint32_t it, uint32_t *unit)
{
    const uint32_t xcd = blockIdx.x & 7u;
    const uint32_t j = blockIdx.x >> 3;
    const uint32_t bpx = gridDim.x >> 3;
    const uint32_t per_xcd = (n_units + 7u) >> 3;
    const uint32_t local = j + it * bpx;
    if (local >= per_xcd) return false;
    const uint32_t u = xcd * per_xcd + local;
    if (u >= n_units) return false;
    *unit = u;
    return true;
}

// Workgroups of `lds_bytes` of LDS that fit one CU. LDS (160 KB per CU on gfx950) is handed out in granules of 1 280 B
// (160 KB / 128) -- measured, not taken from documentation: 12 808-B and 13 336-B workgroups fit 11 per CU and break into
// two rounds at 12 (11 x 14 080 <= 163 840 < 12 x 14 080), 10 724-B workgroups fit 14 (14 x 11 520), DESIGN.md 4.2. A
// byte-granular estimate over-counts by one in exactly the cases where that turns a persistent grid into two rounds.
inline size_t lds_workgroups_per_cu(size_t lds_bytes)
{
    const size_t granule = 1280, total = 160 * 1024;
    const size_t alloc = ((lds_bytes + granule - 1) / granule) * granule;
    return alloc == 0 ? 64 : (total / alloc > 0 ? total / alloc : 1);
}

// Wave-wide minimum, result uniform.  DPP row operations + two row broadcasts (no LDS crossbar): the
// classic gfx9 reduction ladder; lane 63 ends up with the minimum of all 64 lanes.
__device__ __forceinline__ int wave_min_i32(int v)
{
    constexpr int big = 0x7fffffff;
#define AIM_DPP_MIN(ctrl, rmask)                                                               \
    v = min(v, __builtin_amdgcn_update_dpp(big, v, ctrl, rmask, 0xf, false))
    AIM_DPP_MIN(0xB1, 0xf);    // quad_perm [1,0,3,2]
    AIM_DPP_MIN(0x4E, 0xf);    // quad_perm [2,3,0,1]
    AIM_DPP_MIN(0x141, 0xf);   // row_half_mirror
    AIM_DPP_MIN(0x140, 0xf);   // row_mirror  -> every lane holds its row's (16 lanes) minimum
    AIM_DPP_MIN(0x142, 0xa);   // row_bcast:15 into rows 1 and 3
    AIM_DPP_MIN(0x143, 0xc);   // row_bcast:31 into rows 2 and 3
#undef AIM_DPP_MIN
    return __builtin_amdgcn_readlane(v, 63);
}

// 4 bytes starting at byte offset `off` of a dword array (little endian).
template <typename PtrT>
__device__ __forceinline__ uint32_t load4_unaligned(PtrT base, int off, int last_word)
{
    const int w = off >> 2;
    const uint32_t lo = base[w];
    const uint32_t hi = base[(w + 1 <= last_word) ? w + 1 : last_word];
    return __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)(off & 3));
}

}  // namespace aim
