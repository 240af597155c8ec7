// aim_capi.hip -- C-ABI (include/aim_hip.h) over the gfx950 alignment kernels.
//
// Host side of the drop-in boundary: what host.c does with dpu_alloc /
// dpu_push_xfer / dpu_launch (WFA/DPU-WRAM/host/host.c:186-330) is done here
// with one HIP stream per device, HBM buffers planned per configuration, and
// asynchronous copies.  No CPU fallback exists: every compute entry point
// needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <type_traits>
#include <vector>

#include "aim_hip.h"
#include "aim_device.hpp"
#include "wfa_wave.hpp"
#include "wfa_lane.hpp"
#include "wfa_lane_packed.hpp"
#include "wfa_group.hpp"
#include "dp_lane.hpp"
#include "dp_wave.hpp"
#include "dp_strip.hpp"
#include "dp_reg.hpp"
#include "dp_group.hpp"
#include "batch_io.hpp"
#include "genasm_wave.hpp"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail(e_ == hipErrorOutOfMemory ? AIM_ENOMEM : AIM_ENODEV, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                             \
    } while (0)

// ---------------------------------------------------------------------------
// launch planning
// ---------------------------------------------------------------------------
enum KernelId { K_WFA_WAVE = 0, K_WFA_LANE = 1, K_DP_LANE = 2, K_DP_WAVE = 3, K_WFA_GROUP = 4, K_GENASM = 5, K_WFA_LANE_PK = 6, K_DP_STRIP = 7, K_DP_REG = 8, K_DP_GROUP = 9 };

// What a launch is asked to consume / produce besides the default ABI (ASCII rows in, result_t + ops rows out). A plan
// honours a mode bit only when its kernel can (Plan::pk / Plan::emits_runs); otherwise the caller runs the conversion
// kernels of batch_io.hpp around the launch.
enum : uint32_t { MODE_PACKED_IN = 1u, MODE_RUNS_OUT = 2u };

struct Plan {
    KernelId kid;
    uint32_t grid;          // workgroups
    uint32_t block;         // threads per workgroup
    size_t lds;             // dynamic LDS bytes
    uint64_t scratch_per_wg;
    uint32_t pool_cap, meta_cap;
    uint32_t ring_slots, slot_w;
    bool seq_lds;
    size_t scratch_total;
    size_t todo_bytes;      // K_WFA_LANE: to-do region in front of the fallback kernel's scratch
    uint32_t fb_grid;       // K_WFA_LANE / K_WFA_GROUP: grid / LDS of the fallback (general) kernel
    size_t fb_lds;
    bool no_lane;
    size_t hist_bytes;      // K_WFA_GROUP + BACKTRACE: history regions (one per pair of a chunk) between to-do region and fallback scratch
    uint32_t chunk_pairs;   // K_WFA_GROUP + BACKTRACE: pairs per compute + traceback launch (their history regions fit the scratch bound)
    aim::GroupCfg gcfg;     // K_WFA_GROUP
    int group_g;
    int strip_k;            // K_DP_STRIP: cells per lane (K_DP_GROUP: of the fallback launch; 0 = its fallback is a dp_lane.hpp kernel)
    uint32_t fb_block;      // K_DP_GROUP: workgroup size of a dp_strip fallback
    uint64_t fb_scratch_per_wg;   // K_DP_GROUP: ... and its slab size (the fallback's scratch starts at offset 0 like the group kernel's: they never run together)
    bool pack_first;        // K_WFA_LANE_PK on a batch of ASCII rows: pack_rows_kernel first (scratch: to-do | flag bits | packed P | packed T | general kernel)
    size_t pack_bytes;      // ... bytes of the flag bits + both packed arrays
    bool pk;                // the kernel reads the packed rows of the batch itself (no unpack pass)
    bool emits_runs;        // the kernel writes aim_cigar_t + runs itself (no ops rows, no cigar_rle_kernel)
};

// The AIM_* environment variables, read HERE and nowhere else (see aim::Knobs, aim_device.hpp).
int env_int(const char *name, int unset)
{
    const char *e = getenv(name);
    return (e && *e) ? atoi(e) : unset;
}
bool env_flag(const char *name)
{
    const char *e = getenv(name);
    return e && e[0] == '1';
}
aim::Knobs read_knobs()
{
    aim::Knobs k;
    if (const char *e = getenv("AIM_SCRATCH_GB")) k.scratch_gb = std::max(0.25, atof(e));
    k.force_wave = env_flag("AIM_FORCE_WAVE");
    k.no_group = env_flag("AIM_NO_GROUP");
    k.no_lane_ext = env_flag("AIM_NO_LANE_EXT");
    k.no_lane = env_flag("AIM_NO_LANE");
    k.no_lane_pk = env_flag("AIM_NO_LANE_PK");
    k.group_overlap = env_flag("AIM_GROUP_OVERLAP");
    k.wfa_no_ring = env_flag("AIM_WFA_NO_RING");
    k.wfa_slotw = env_int("AIM_WFA_SLOTW", -1);
    k.force_dpwave = env_flag("AIM_FORCE_DPWAVE");
    k.dpw_legacy = env_flag("AIM_DPW_LEGACY");
    k.strip_k = env_int("AIM_STRIP_K", -1);
    k.dpw_nw = env_int("AIM_DPW_NW", -1);
    k.dpl_seq_lds = env_int("AIM_DPL_SEQ_LDS", -1);
    k.dpl_no_reg = env_int("AIM_DPL_NO_REG", 0);
    k.dpl_per_cu = env_int("AIM_DPL_PER_CU", -1);
    k.no_nw_reg = env_flag("AIM_NO_NW_REG");
    k.no_swg_reg = env_flag("AIM_NO_SWG_REG");
    k.no_dp_group = env_flag("AIM_NO_DP_GROUP");
    k.dpg_per_cu = env_int("AIM_DPG_PER_CU", -1);
    k.dbg_flags = env_int("AIM_DEBUG_FLAGS", 0);
#ifndef AIM_DIAG_BUILD   // bits 1 / 4 / 8 make kernels skip work (a traceback walk, the direction-bit stores, the bits themselves) and return WRONG results with status OK:
    k.dbg_flags &= 2;    // they exist in diagnostic builds only (python -m aim_amd.build --variant diag --flags "-DAIM_DIAG_BUILD=1"). Bit 2 changes the route, not the result.
#endif
    k.nw_reg_per_cu = env_int("AIM_NW_REG_PER_CU", -1);
    k.group_lds_kb = env_int("AIM_GROUP_LDS_KB", -1);
    k.group_g = env_int("AIM_GROUP_G", -1);
    k.group_per_cu = env_int("AIM_GROUP_PER_CU", -1);
    k.group_wlds = env_int("AIM_GROUP_WLDS", -1);
    k.group_unit1 = env_int("AIM_GROUP_UNIT1", 0);
    k.ga_per_cu = env_int("AIM_GA_PER_CU", 0);
    k.poison_scratch = env_int("AIM_DEBUG_POISON_SCRATCH", -1);
    k.poison_ops = env_int("AIM_DEBUG_POISON_OPS", -1);
    k.poison_lds = env_int("AIM_DEBUG_POISON_LDS", -1);
    k.plan_debug = getenv("AIM_PLAN_DEBUG") != nullptr;
    k.cus = (uint32_t)std::max(0, env_int("AIM_CHIP_CUS", 0));   // 0: ask the device (chip_cus)
    return k;
}

// Compute units of the CURRENT device, read once per device (256 on a whole MI355X; a DPX / QPX / CPX partition or a CU-reduced part
// reports fewer and every persistent grid shrinks with it). Without a device (planning queries on a CPU-only host) the whole chip.
uint32_t chip_cus()
{
    static uint32_t cached[64] = {0};
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return 256;
    }
    if (!cached[dev]) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) {
            (void)hipGetLastError();
            return 256;
        }
        cached[dev] = (uint32_t)n;
    }
    return cached[dev];
}
// Knobs for a plan on the current device: AIM_CHIP_CUS wins, else the device's own count.
aim::Knobs with_chip(aim::Knobs k)
{
    if (k.cus == 0) k.cus = chip_cus();
    return k;
}

// Upper bound on the HBM scratch one plan may ask for (DP tables, WFA history/pools). AIM_SCRATCH_GB overrides. Plans
// are need-capped, so the bound only matters where the tables are huge: config 4 (l = 10 000, 613 MB per pair) runs 128
// pairs in 5 rounds under a 16 GB bound and in 1 round (4.4x faster) from 80 GB up; a full chip of them (256 pairs, one
// per CU) needs 157 GB, which is why the default is 3/4 of the free memory and not 1/2.
//  * device sets: the bound is taken per device inside aim_set_configure, AFTER the set's fixed buffers are allocated,
//    and frozen in the set together with the plan;
//  * stateless entry points (aim_scratch_bytes, then aim_align_device with the buffer the caller allocated in between):
//    the bound must not move between the two calls, so it is read once per process and device.
uint64_t budget_from_free(size_t free_b) { return std::max<uint64_t>((uint64_t)free_b / 4 * 3, (uint64_t)1 << 28); }

uint64_t stateless_budget_bytes(const aim::Knobs &kn)
{
    if (kn.scratch_gb >= 0) return (uint64_t)(kn.scratch_gb * (double)(1ull << 30));
    static uint64_t cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        (void)hipGetLastError();
        return (uint64_t)16 << 30;                           // no device (CPU-only host): planning queries still answer
    }
    if (!cached[dev]) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b == 0) {
            (void)hipGetLastError();
            return (uint64_t)16 << 30;
        }
        cached[dev] = budget_from_free(free_b);
    }
    return cached[dev];
}

int validate_params(const aim_params_t &p)
{
    if (p.algo != AIM_ALGO_NW && p.algo != AIM_ALGO_SWG && p.algo != AIM_ALGO_WFA && p.algo != AIM_ALGO_GENASM)
        return fail(AIM_EINVAL, "unknown algorithm %d", p.algo);
    if (p.read_size <= 0 || (p.read_size & 7)) return fail(AIM_EINVAL, "read_size must be a positive multiple of 8 (got %d)", p.read_size);
    if (p.max_score < 0) return fail(AIM_EINVAL, "max_score must be >= 0");
    if ((p.flags & AIM_FLAG_REQ8) && p.read_size >= 32760)
        return fail(AIM_EINVAL, "AIM_FLAG_REQ8 carries int16 lengths: read_size must be < 32760");
    if (p.algo == AIM_ALGO_GENASM) {   // no penalties, no score cap; lengths are int32
        if ((p.flags & AIM_FLAG_RES8) && (p.flags & AIM_FLAG_BACKTRACE))
            return fail(AIM_EINVAL, "AIM_FLAG_RES8 (idx, score results) cannot be combined with AIM_FLAG_BACKTRACE");
        if (p.read_size > (1 << 24)) return fail(AIM_EINVAL, "read_size must be <= 2^24");
        return AIM_OK;
    }
    // same admission rule as the launchers (run-wfa-pim-wram.py:41-43): m <= 0 and x, g, a > 0
    if (p.algo == AIM_ALGO_NW) {
        if (p.mismatch <= 0 || p.gap_i <= 0 || p.gap_d <= 0) return fail(AIM_EINVAL, "NW penalties must be x, g > 0");
    } else {
        if (p.match > 0 || p.mismatch <= 0 || p.gap_o <= 0 || p.gap_e <= 0)
            return fail(AIM_EINVAL, "Wrong affine gap penalties must be  m <= 0 and g, a, x > 0");
    }
    if ((p.flags & AIM_FLAG_RES8) && (p.flags & AIM_FLAG_BACKTRACE))
        return fail(AIM_EINVAL, "AIM_FLAG_RES8 (idx, score results) cannot be combined with AIM_FLAG_BACKTRACE");
    // the reference's lengths, WFA offsets and NW / SWG cells are int16 (WFA/DPU-WRAM/common/common.h:98-100, 174-175): what it admits
    // is < 32 767; READ_SIZE is a multiple of 8
    if (p.read_size >= 32760)
        return fail(AIM_EINVAL, "%s are int16: read_size must be < 32760", p.algo == AIM_ALGO_WFA ? "WFA offsets (common.h:98-100)" : "NW/SWG cells");
    return AIM_OK;
}

inline bool p_is_nw(const aim_params_t *p) { return p->algo == AIM_ALGO_NW; }

int make_plan_inner(const aim_params_t &p, uint32_t n_pairs, const aim::Knobs &kn, uint64_t budget, uint32_t mode, Plan *pl)
{
    int rc = AIM_OK;
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    if (p.algo == AIM_ALGO_GENASM) {
        pl->kid = K_GENASM;
        aim::genasm_plan(p, kn, n_pairs, &pl->grid, &pl->block, &pl->lds);
        pl->scratch_per_wg = aim::kGaSlabBytes;
        pl->scratch_total = (size_t)pl->grid * aim::kGaSlabBytes;   // slow-path columns, one slab per wavefront
        return AIM_OK;
    }
    if (p.algo == AIM_ALGO_WFA && (mode & MODE_PACKED_IN) && !kn.force_wave && !kn.no_lane_pk && !kn.no_lane && !pl->no_lane &&
        aim::wfa_lane_packed_supported(p, !kn.no_lane_ext)) {
        // packed rows in; {idx, score}, compact CIGAR or result_t + ops rows out: one kernel per batch, no scratch (wfa_lane_packed.hpp)
        pl->kid = K_WFA_LANE_PK;
        aim::wfa_lane_packed_plan(p, n_pairs, kn, &pl->grid, &pl->block, &pl->lds);
        pl->scratch_total = 256;
        pl->pk = true;
        pl->emits_runs = bt && (mode & MODE_RUNS_OUT);
        return AIM_OK;
    }
    if (p.algo == AIM_ALGO_WFA) {
        const bool lane_ok = !kn.force_wave && !kn.no_lane && !pl->no_lane && aim::wfa_lane_supported(p, !kn.no_lane_ext);
        aim::GroupCfg gc;
        int gg = 0;
        uint32_t ggrid = 0, gchunk = n_pairs;
        size_t glds = 0, ghist = 0, ghist_pair = 0;
        const bool gpk = (mode & MODE_PACKED_IN) && !kn.no_lane_pk;   // the group kernel reads packed rows itself
        bool group_ok = !lane_ok && !kn.force_wave && !pl->no_lane && !kn.no_group &&
                        aim::wfa_group_plan(p, n_pairs, kn, gpk, &gc, &gg, &ggrid, &glds, &ghist_pair);
        if (group_ok && ghist_pair) {
            // BACKTRACE: every pair of a launch keeps its history region until the traceback kernel has walked it. Launches are
            // chunks of the batch whose regions fit half of the scratch bound (one chunk whenever possible).
            // One launch when the regions fit. (AIM_GROUP_OVERLAP=1: chunks of two grid rounds, two alternating buffers of regions, the
            // traceback kernel of chunk c on a second stream while chunk c + 1 is computed. Measured and NOT adopted: cfg3 3.57 ->
            // 3.79 ms, l = 100 e = 5 % 1.62 -> 1.75 ms -- several shorter compute launches lose more to ramp-up and partial rounds than
            // the overlapped 10 % traceback gives back. Kept as a tested code path: it is also what a batch does whose regions do
            // not fit the scratch bound.)
            const uint64_t fit = (budget / 4) / ghist_pair;          // per buffer
            if (fit < 4096 && fit < n_pairs) group_ok = false;   // too few pairs in flight to fill the chip: the general kernel's pools are smaller
            else {
                const uint64_t two_rounds = 2ull * ggrid * (uint64_t)(64 / gg);
                uint64_t chunk = std::min<uint64_t>(fit, std::max<uint64_t>(two_rounds, 4096));
                if (n_pairs <= 2 * fit && (chunk >= n_pairs || !kn.group_overlap)) chunk = n_pairs;   // one launch, one buffer of regions (half the bound)
                else if (!kn.group_overlap) chunk = fit;                                                  // as few launches as fit, over the two buffers
                gchunk = (uint32_t)chunk;
                if (gchunk < n_pairs) gchunk &= ~63u;
                const uint32_t nchunks = (n_pairs + gchunk - 1) / gchunk;
                ghist = ((((size_t)gchunk * ghist_pair) + 255) & ~(size_t)255) * (nchunks > 1 ? 2 : 1);
            }
        }
        if (lane_ok) {
            // one pair per lane, everything in registers; no scratch, no second kernel (pairs with bytes outside A/C/G/T take
            // the kernel's own raw-byte path)
            pl->kid = K_WFA_LANE;
            aim::wfa_lane_plan(p, n_pairs, kn, &pl->grid, &pl->block, &pl->lds);
            pl->scratch_total = 256;   // unused; aim_scratch_bytes() keeps 0 for "invalid configuration"
#if AIM_LANE_STAMPS
            pl->scratch_total += (size_t)pl->grid * 64;   // diagnostic builds park their s_memtime sums behind the first 256 bytes
#endif
            return AIM_OK;
        }
        if (!lane_ok && !kn.force_wave && !kn.no_lane_pk && !kn.no_lane && !pl->no_lane && aim::wfa_lane_packed_supported(p, !kn.no_lane_ext)) {
            // ASCII rows of a shape only the packed lane kernel takes (READ_SIZE other than 80 / 112: l = 150 and friends; CIGAR at
            // MAX_SCORE 6..10): pack on the device (batch_io.hpp), run the packed kernel, let the general kernel re-align the
            // non-ACGT pairs (to-do list)
            Plan fb;
            memset(&fb, 0, sizeof fb);
            fb.no_lane = true;
            aim::Knobs kq = kn;
            kq.no_group = true;
            rc = make_plan_inner(p, std::min<uint32_t>(n_pairs, 1024u * 64u), kq, budget, 0u, &fb);
            if (rc) return rc;
            *pl = fb;
            pl->fb_grid = fb.grid;
            pl->fb_lds = fb.lds;
            pl->kid = K_WFA_LANE_PK;
            aim::wfa_lane_packed_plan(p, n_pairs, kn, &pl->grid, &pl->block, &pl->lds);
            pl->pack_first = true;
            pl->todo_bytes = aim::wfa_lane_todo_bytes(n_pairs);
            const size_t npw = aim::packed_row_dwords(p.read_size);
            pl->pack_bytes = ((((size_t)n_pairs + 31) / 32 * 4 + 255) & ~(size_t)255) + 2 * ((((size_t)n_pairs * npw * 4) + 64 + 255) & ~(size_t)255);
            pl->scratch_total = pl->todo_bytes + pl->pack_bytes + fb.scratch_total;
            return AIM_OK;
        }
        if (group_ok) {
            // fast path + the general kernel in to-do mode behind it (its plan goes into *pl first)
            Plan fb;
            memset(&fb, 0, sizeof fb);
            fb.no_lane = true;
            rc = make_plan_inner(p, std::min<uint32_t>(n_pairs, 1024u * 64u), kn, budget, 0u, &fb);
            if (rc) return rc;
            *pl = fb;
            pl->fb_grid = fb.grid;
            pl->fb_lds = fb.lds;
            pl->kid = K_WFA_GROUP;
            pl->gcfg = gc;
            pl->group_g = gg;
            pl->grid = ggrid;
            pl->block = 64;
            pl->lds = glds;
            pl->todo_bytes = aim::wfa_lane_todo_bytes(n_pairs);
            pl->hist_bytes = ghist;
            pl->chunk_pairs = gchunk;
            pl->pk = gpk;
            pl->emits_runs = bt && (mode & MODE_RUNS_OUT);
            pl->scratch_total = pl->todo_bytes + pl->hist_bytes + fb.scratch_total;
            return AIM_OK;
        }
        pl->kid = K_WFA_WAVE;
        pl->block = 64;
        const uint64_t ms = (uint64_t)p.max_score;
        const uint64_t full = 3 * (ms + 2) * (ms + 2) + 64;
        uint64_t cap;
        if (bt) {
            cap = full;
        } else {   // score-only: the pool is a ring that must hold the live window (scores s-R .. s) plus the one being built
            const uint64_t R = (uint64_t)std::max(p.mismatch, p.gap_o + p.gap_e);
            cap = std::min(full, (R + 2) * 3 * (2 * ms + 3));
        }
        const uint64_t cap_min = bt ? 0 : cap;   // below this a score-only ring would overwrite wavefronts still in use
        pl->meta_cap = (uint32_t)(ms + 2);
        // LDS ring for the live window of wavefronts: max(x, o+e)+1 slots of slot_w diagonals (M, I, D)
        {
            const uint32_t R = (uint32_t)std::max(p.mismatch, p.gap_o + p.gap_e);
            uint32_t w = 16;
            while (w < 2 * (uint32_t)ms + 3 && w < 128) w *= 2;   // 128: keeps 16 workgroups resident per CU at l = 1000 (measured +9 % over 256)
            if (kn.wfa_slotw >= 0) w = (uint32_t)std::max(16, kn.wfa_slotw) & ~15u;
            while (w > 16 && (uint64_t)(R + 1) * 3 * w * 2 > 24 * 1024) w /= 2;
            const bool ring_ok = (uint64_t)(R + 1) * 3 * w * 2 <= 24 * 1024 && !kn.wfa_no_ring;
            pl->ring_slots = ring_ok ? R + 1 : 0;
            pl->slot_w = ring_ok ? w : 0;
        }
        const size_t seq_bytes = 2 * ((size_t)p.read_size + 8);
        pl->seq_lds = seq_bytes <= 40 * 1024;
        const size_t ring_bytes = ((size_t)pl->ring_slots * 3 * pl->slot_w * sizeof(int16_t) + 15) & ~(size_t)15;
        pl->lds = aim::kMetaRing * sizeof(aim::WfMeta) + ring_bytes + (pl->seq_lds ? seq_bytes : 0);
        // persistent single-wave workgroups: exactly what is resident (4 waves/SIMD by VGPRs, 160 KiB LDS per CU);
        // a larger grid runs in uneven rounds
        const uint32_t wg_per_cu = (uint32_t)std::min<size_t>(16, aim::lds_workgroups_per_cu(pl->lds));
        uint32_t grid = aim::resident_grid(kn, wg_per_cu);
        const uint32_t need = ((n_pairs + 7u) / 8u) * 8u;
        if (grid > need) grid = std::max(8u, need);
        uint64_t per = (uint64_t)pl->meta_cap * sizeof(aim::WfMeta) + cap * sizeof(int16_t);
        per = (per + 255) & ~255ull;
        while (grid > 2 * kn.cus && grid > 16 && per * grid > budget) grid = ((grid / 2) + 7u) & ~7u;
        if (per * grid > budget) {
            const uint64_t meta_b = (uint64_t)pl->meta_cap * sizeof(aim::WfMeta);
            if (bt) {
                // With BACKTRACE the pool is a bump arena like the DPU's (allocate_new_score, wfa.c:143-183): a smaller one
                // is legal and a pair that outgrows it reports AIM_PAIR_NOMEM (dpu_allocator_wram.c:19-23 "out of memory").
                uint64_t avail = budget / grid;
                if (avail < meta_b + 4096) return fail(AIM_ENOMEM, "scratch budget too small for max_score %d", p.max_score);
                cap = (avail - meta_b - 256) / sizeof(int16_t);
                per = (meta_b + cap * sizeof(int16_t) + 255) & ~255ull;
            } else {
                // Score-only: the pool is a ring and must keep its full live window (a shrunken ring silently overwrites
                // wavefronts score-x / score-o-e / score-e still read). Fewer workgroups instead, down to one per XCD.
                while (grid > 8 && per * grid > budget) grid -= 8;
                if (per * grid > budget || cap < cap_min)
                    return fail(AIM_ENOMEM, "scratch budget too small for the live window of max_score %d", p.max_score);
            }
        }
        pl->grid = grid;
        pl->pool_cap = (uint32_t)std::min<uint64_t>(cap, 0x7fffffffu);
        pl->scratch_per_wg = per;
        pl->scratch_total = (size_t)(per * grid);
        return AIM_OK;
    }
    // NW / SWG medium reads: G lanes per pair (dp_group.hpp); empty sequences and plen > 2 tlen reach nw_lane / swg_lane (READ_SIZE <= 320) or
    // dp_strip_kernel (to-do mode) through the to-do list. [the pairs' direction-bit slabs OR, afterwards, the fallback kernel's scratch | to-do region]
    if (aim::dp_group_supported(p, kn)) {
        // (a budget that holds neither this plan nor its fallback is not an error: the plans below -- dp_strip / dp_lane alone -- are tried next)
        Plan fb;
        memset(&fb, 0, sizeof fb);
        bool ok = true;
        if (p.read_size <= 320) {
            aim::Knobs kq = kn;
            kq.dpl_seq_lds = 0;   // the to-do pass: every lane loads its own pair's rows from global memory (the listed pairs are not consecutive)
            kq.dpl_no_reg = 0;
            ok = aim::dp_lane_plan(p, n_pairs, budget / 2, kq, &fb.grid, &fb.block, &fb.lds, &fb.scratch_per_wg, &fb.scratch_total, &fb.seq_lds);
            fb.strip_k = 0;
        } else
            ok = aim::dp_strip_plan(p, n_pairs, budget / 2, kn, &fb.grid, &fb.block, &fb.lds, &fb.scratch_per_wg, &fb.scratch_total, &fb.strip_k, &fb.pool_cap);
        if (ok) {
            uint32_t grid = 0;
            size_t lds = 0;
            uint64_t per = 0;
            aim::dp_group_plan(p, n_pairs, kn, &grid, &lds, &per);
            while (grid > 8 && per * grid > budget / 2) grid -= 8;
            ok = per * grid <= budget / 2;
            if (ok) {
                pl->kid = K_DP_GROUP;
                pl->block = 64;
                pl->grid = grid;
                pl->lds = lds;
                pl->scratch_per_wg = per;
                pl->fb_scratch_per_wg = fb.scratch_per_wg;
                pl->fb_grid = fb.grid;
                pl->fb_block = fb.block;
                pl->fb_lds = fb.lds;
                pl->strip_k = fb.strip_k;
                pl->pool_cap = fb.pool_cap;
                pl->todo_bytes = aim::wfa_lane_todo_bytes(n_pairs);
                pl->scratch_total = ((std::max<size_t>(fb.scratch_total, (size_t)(pl->scratch_per_wg * pl->grid)) + 255) & ~(size_t)255) + pl->todo_bytes;
                return AIM_OK;
            }
        }
    }
    // NW / SWG long reads: one pair per workgroup of 1-12 wavefronts, row-scan, canonical table in per-workgroup HBM scratch
    // (SWG with int8 cells -- the launchers' MAX_SCORE < 127 -- wraps by design and is nobody's but the literal kernels': swg_lane_kernel, one pair per LANE with its M and I rows
    //  in LDS, takes it as far as 64 lanes' rows fit a CU's LDS (READ_SIZE <= 1 199; round 6: until then dp_wave_kernel's literal path, one lane per WORKGROUP, from READ_SIZE 321 on:
    //  6 - 13 GCUPS, profiles/r06/cliff_scan.txt))
    bool cell8_lane = p.algo == AIM_ALGO_SWG && aim::swg_cell_bytes(p) == 1 && !kn.force_dpwave && !kn.dpw_legacy && kn.dpw_nw < 0 && kn.strip_k <= 0 &&
                      (size_t)2 * (p.read_size + 1) * 64 <= 150 * 1024;
    if (cell8_lane && p.read_size > 320) {   // (with CIGAR a lane's table is READ_SIZE^2 x 4 bytes x 64 lanes per workgroup: a budget that does not hold eight of them keeps the one-table-per-workgroup path)
        Plan t;
        memset(&t, 0, sizeof t);
        cell8_lane = aim::dp_lane_plan(p, n_pairs, budget, kn, &t.grid, &t.block, &t.lds, &t.scratch_per_wg, &t.scratch_total, &t.seq_lds);
    }
    if ((p.read_size > 320 && !cell8_lane) || kn.force_dpwave) {
        const bool cell8 = p.algo == AIM_ALGO_SWG && aim::swg_cell_bytes(p) == 1;
        if (!kn.dpw_legacy && kn.dpw_nw < 0 && aim::dp_strip_supported(p, cell8, kn)) {
            // column-strip pipeline (dp_strip.hpp): previous row in registers, packed int16 arithmetic, mailboxes instead of barriers
            pl->kid = K_DP_STRIP;
            return aim::dp_strip_plan(p, n_pairs, budget, kn, &pl->grid, &pl->block, &pl->lds, &pl->scratch_per_wg, &pl->scratch_total, &pl->strip_k, &pl->pool_cap)
                       ? AIM_OK
                       : fail(AIM_ENOMEM, "scratch budget (AIM_SCRATCH_GB) or LDS too small for read_size %d", p.read_size);
        }
        pl->kid = K_DP_WAVE;
        return aim::dp_wave_plan(p, n_pairs, budget, kn, cell8, &pl->grid, &pl->block, &pl->lds, &pl->scratch_per_wg, &pl->scratch_total)
                   ? AIM_OK
                   : fail(AIM_ENOMEM, "scratch budget (AIM_SCRATCH_GB) or LDS too small for read_size %d", p.read_size);
    }
    // NW short reads: the row in registers (dp_reg.hpp); pairs it cannot take (tail cells: plen >= tlen + 2, short outliers) reach
    // nw_lane_kernel through a to-do list. [table slabs of max(grid) wavefronts | to-do region]
    if (p.algo == AIM_ALGO_NW && !kn.no_nw_reg && !kn.force_dpwave && aim::nw_reg_supported(p)) {
        Plan fb;
        memset(&fb, 0, sizeof fb);
        aim::Knobs kq = kn;
        kq.dpl_seq_lds = p.read_size <= 124 ? 2 : 0;   // the to-do pass: every lane loads its own pair's pattern row into registers (the listed pairs are not consecutive: no staged image); READ_SIZE 128: from global memory -- never an LDS image (the plan's LDS and the launch agree)
        kq.dpl_no_reg = 0;
        if (!aim::dp_lane_plan(p, n_pairs, budget / 2, kq, &fb.grid, &fb.block, &fb.lds, &fb.scratch_per_wg, &fb.scratch_total, &fb.seq_lds))
            return fail(AIM_ENOMEM, "scratch budget too small for read_size %d", p.read_size);
        const int npk = aim::nw_reg_npk(p.read_size, bt);
        const uint64_t slab = bt ? (uint64_t)aim::nw_reg_slab_bytes(npk, p.read_size) : 256;
        uint32_t per_cu = npk <= 42 ? (bt ? 8u : 8u) : 8u;   // (178 - 255 VGPRs: two wavefronts per SIMD)
        if (kn.nw_reg_per_cu > 0) per_cu = (uint32_t)kn.nw_reg_per_cu;
        pl->lds = aim::nw_reg_lds_bytes(p);
        per_cu = (uint32_t)std::min<size_t>(per_cu, aim::lds_workgroups_per_cu(pl->lds));
        const uint32_t n_groups = (n_pairs + 63u) / 64u;
        uint32_t g = aim::resident_grid(kn, per_cu);
        const uint32_t need = ((n_groups + 7u) / 8u) * 8u;
        if (g > need) g = need < 8u ? 8u : need;
        const uint64_t per = std::max<uint64_t>((slab + 255) & ~255ull, fb.scratch_per_wg);
        while (g > 8 && per * g > budget / 2) g -= 8;
        if (per * std::max(g, fb.grid) > budget) return fail(AIM_ENOMEM, "scratch budget too small for read_size %d", p.read_size);
        pl->kid = K_DP_REG;
        pl->grid = g;
        pl->block = 64;
        pl->scratch_per_wg = per;
        pl->fb_grid = fb.grid;
        pl->fb_lds = fb.lds;
        pl->todo_bytes = aim::wfa_lane_todo_bytes(n_pairs);
        pl->scratch_total = (size_t)(per * std::max(g, fb.grid)) + pl->todo_bytes;
        return AIM_OK;
    }
    // SWG short reads: the M and I rows in registers (dp_reg.hpp, round 5); tail pairs, outliers and pairs whose cells wrap reach swg_lane_kernel
    // through the same to-do list. [table slabs of max(grid) wavefronts | to-do region]
    if (p.algo == AIM_ALGO_SWG && !kn.no_swg_reg && !kn.force_dpwave && aim::swg_reg_supported(p)) {
        Plan fb;
        memset(&fb, 0, sizeof fb);
        aim::Knobs kq = kn;
        kq.dpl_seq_lds = p.read_size <= 124 ? 2 : 0;
        kq.dpl_no_reg = 0;
        if (!aim::dp_lane_plan(p, n_pairs, budget / 2, kq, &fb.grid, &fb.block, &fb.lds, &fb.scratch_per_wg, &fb.scratch_total, &fb.seq_lds))
            return fail(AIM_ENOMEM, "scratch budget too small for read_size %d", p.read_size);
        const int npk = aim::swg_reg_npk(p.read_size);
        const uint64_t slab = bt ? (uint64_t)aim::swg_reg_slab_bytes(npk, p.read_size) : 256;
        uint32_t per_cu = 8u;   // two wavefronts per SIMD
        if (kn.nw_reg_per_cu > 0) per_cu = (uint32_t)kn.nw_reg_per_cu;
        pl->lds = aim::swg_reg_lds_bytes(p, npk);
        per_cu = (uint32_t)std::min<size_t>(per_cu, aim::lds_workgroups_per_cu(pl->lds));
        const uint32_t n_groups = (n_pairs + 63u) / 64u;
        uint32_t g = aim::resident_grid(kn, per_cu);
        const uint32_t need = ((n_groups + 7u) / 8u) * 8u;
        if (g > need) g = need < 8u ? 8u : need;
        const uint64_t per = std::max<uint64_t>((slab + 255) & ~255ull, fb.scratch_per_wg);
        while (g > 8 && per * g > budget / 2) g -= 8;
        if (per * std::max(g, fb.grid) > budget) return fail(AIM_ENOMEM, "scratch budget too small for read_size %d", p.read_size);
        pl->kid = K_DP_REG;
        pl->grid = g;
        pl->block = 64;
        pl->scratch_per_wg = per;
        pl->fb_grid = fb.grid;
        pl->fb_lds = fb.lds;
        pl->todo_bytes = aim::wfa_lane_todo_bytes(n_pairs);
        pl->scratch_total = (size_t)(per * std::max(g, fb.grid)) + pl->todo_bytes;
        return AIM_OK;
    }
    // NW / SWG short reads: one pair per lane, flat DP table in per-wave HBM scratch
    pl->kid = K_DP_LANE;
    return aim::dp_lane_plan(p, n_pairs, budget, kn, &pl->grid, &pl->block, &pl->lds, &pl->scratch_per_wg, &pl->scratch_total,
                             &pl->seq_lds)
               ? AIM_OK
               : fail(AIM_ENOMEM, "scratch budget too small for read_size %d", p.read_size);
}

const char *kernel_name(const Plan &pl, const aim_params_t &p)
{
    switch (pl.kid) {
    case K_WFA_WAVE: return "wfa_wave_kernel";
    case K_WFA_LANE: return "wfa_lane_kernel";
    case K_WFA_LANE_PK: return "wfa_lane_packed_kernel";
    case K_WFA_GROUP: return "wfa_group_kernel";
    case K_DP_LANE: return p_is_nw(&p) ? "nw_lane_kernel" : "swg_lane_kernel";
    case K_DP_REG: return p_is_nw(&p) ? "nw_reg_kernel" : "swg_reg_kernel";
    case K_DP_WAVE: return "dp_wave_kernel";
    case K_DP_STRIP: return "dp_strip_kernel";
    case K_DP_GROUP: return "dp_group_kernel";
    case K_GENASM: return "genasm_wave_kernel";
    }
    return "";
}

// One line that identifies a plan completely: kernel, lanes / wavefronts per pair, grid, block, LDS, scratch.
int describe_plan(const Plan &pl, const aim_params_t &p, uint32_t n_pairs, uint64_t budget, char *out, size_t cap)
{
    char extra[160] = "";
    if (pl.kid == K_WFA_GROUP) snprintf(extra, sizeof extra, " G=%d hist=%zu chunk=%u fb_grid=%u packed_in=%d runs_out=%d", pl.group_g, pl.hist_bytes, pl.chunk_pairs, pl.fb_grid, (int)pl.pk, (int)pl.emits_runs);
    else if (pl.kid == K_WFA_LANE_PK) snprintf(extra, sizeof extra, " pack_first=%d fb_grid=%u", (int)pl.pack_first, pl.fb_grid);
    else if (pl.kid == K_DP_WAVE) snprintf(extra, sizeof extra, " wavefronts_per_pair=%u", pl.block / 64);
    else if (pl.kid == K_DP_STRIP) snprintf(extra, sizeof extra, " wavefronts_per_pair=%u cells_per_lane=%d", pl.block / 64, pl.strip_k);
    else if (pl.kid == K_WFA_WAVE) snprintf(extra, sizeof extra, " pool_cap=%u ring=%ux%u seq_lds=%d", pl.pool_cap, pl.ring_slots, pl.slot_w, (int)pl.seq_lds);
    else if (pl.kid == K_DP_LANE) snprintf(extra, sizeof extra, " seq_lds=%d", (int)pl.seq_lds);
    else if (pl.kid == K_DP_REG) snprintf(extra, sizeof extra, " fb_grid=%u fb_lds=%zu", pl.fb_grid, pl.fb_lds);
    else if (pl.kid == K_DP_GROUP) snprintf(extra, sizeof extra, " lanes_per_pair=%d fb_grid=%u fb_block=%u fb_lds=%zu", aim::dp_group_lanes(p.read_size, (p.flags & AIM_FLAG_BACKTRACE) != 0, p.algo == AIM_ALGO_SWG), pl.fb_grid, pl.fb_block, pl.fb_lds);
    return snprintf(out, cap, "%s n=%u grid=%u block=%u lds=%zu scratch=%zu budget=%llu%s", kernel_name(pl, p), n_pairs, pl.grid,
                    pl.block, pl.lds, pl.scratch_total, (unsigned long long)budget, extra);
}

int make_plan(const aim_params_t &p, uint32_t n_pairs, const aim::Knobs &kn, uint64_t budget, Plan *pl, uint32_t mode = 0u)
{
    int rc = validate_params(p);
    if (rc) return rc;
    memset(pl, 0, sizeof *pl);
    rc = make_plan_inner(p, n_pairs, kn.cus ? kn : with_chip(kn), budget, mode, pl);
    if (rc == AIM_OK && kn.plan_debug) {
        char line[384];
        describe_plan(*pl, p, n_pairs, budget, line, sizeof line);
        fprintf(stderr, "[aim plan] %s\n", line);
    }
    return rc;
}

void launch_wfa_wave(bool bt, bool red, const Plan &pl, const aim::KArgs &ka, hipStream_t s)
{
    aim::wfa_wave_launch(bt, red, pl.seq_lds, pl.grid, pl.lds, ka, s);
}

// A second stream per device for work that only has to be ordered against ONE kernel of a launch, not against the whole stream:
// wfa_group's traceback kernel of chunk c runs there while chunk c + 1 is computed on the caller's stream (the traceback is a
// latency-bound walk -- 93 % of its wavefront cycles are waits -- so it costs the compute kernel next to nothing). Events are
// re-recorded per launch; a hipStreamWaitEvent captures the record that precedes it, so re-use across launches is safe.
// OWNERSHIP: every slot of a device set has its own (created on first use, destroyed with the slot), so two host threads driving two
// sets -- or two slots -- on one device never record / wait on each other's events. The stateless aim_align_device has no slot: it
// uses one per-device instance and holds that device's mutex across its whole chunk loop.
struct AuxStream {
    hipStream_t stream = nullptr;
    hipEvent_t computed[2] = {nullptr, nullptr};   // compute kernel of the chunk that uses history buffer b has finished
    hipEvent_t walked[2] = {nullptr, nullptr};     // traceback of that chunk has finished (buffer b is free again)
};
int aux_stream_create(AuxStream &a)
{
    if (a.stream) return AIM_OK;
    HIP_TRY(hipStreamCreateWithFlags(&a.stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipEventCreateWithFlags(&a.computed[i], hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&a.walked[i], hipEventDisableTiming));
    }
    return AIM_OK;
}
void aux_stream_destroy(AuxStream &a)
{
    for (int i = 0; i < 2; ++i) {
        if (a.computed[i]) (void)hipEventDestroy(a.computed[i]);
        if (a.walked[i]) (void)hipEventDestroy(a.walked[i]);
    }
    if (a.stream) (void)hipStreamDestroy(a.stream);
    a = AuxStream();
}
struct SharedAux { std::mutex mu; AuxStream aux; };
SharedAux *shared_aux_for_current_device()
{
    static SharedAux table[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return nullptr; }
    return &table[dev];
}

// The fused batch I/O of a launch (Plan::pk / Plan::emits_runs): packed rows in, compact CIGAR out.
struct FusedIo {
    const uint32_t *packedP = nullptr, *packedT = nullptr;
    aim_cigar_t *cig = nullptr;
    uint32_t *runs = nullptr, *cursor = nullptr;
    uint32_t runs_cap = 0, run_slot = 0;
};

// Enqueue one alignment launch that follows plan `pl` (made for >= n_pairs pairs under the caller's knobs and budget).
int launch(const Plan &pl, const aim::Knobs &kn, const aim_params_t &p, uint32_t n_pairs, const void *d_req,
           const char *d_pat, const char *d_txt, void *d_res, char *d_ops, void *d_scratch, size_t scratch_bytes,
           hipStream_t stream, const FusedIo *fio = nullptr, AuxStream *slot_aux = nullptr)
{
    if (n_pairs == 0) return AIM_OK;
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    const bool red = p.flags & AIM_FLAG_REDUCE;
    if (pl.pk && !pl.pack_first && (!fio || !fio->packedP || !fio->packedT)) return fail(AIM_EINVAL, "plan reads packed rows but none were given");
    if (pl.emits_runs && (!fio || !fio->cig || !fio->runs || !fio->cursor)) return fail(AIM_EINVAL, "plan emits the compact CIGAR but no buffers were given");
    const bool reads_ascii = !(pl.kid == K_WFA_LANE_PK && !pl.pack_first);   // (wfa_group reads them for its to-do pairs even on packed batches)
    const bool writes_res = !(pl.kid == K_WFA_LANE_PK && pl.emits_runs);
    if (!d_req || (reads_ascii && (!d_pat || !d_txt)) || (writes_res && !d_res)) return fail(AIM_EINVAL, "null device buffer");
    if (bt && writes_res && !d_ops) return fail(AIM_EINVAL, "AIM_FLAG_BACKTRACE needs an ops buffer");
    if (scratch_bytes < pl.scratch_total || (!d_scratch && pl.scratch_total))
        return fail(AIM_EINVAL, "scratch too small: need %zu bytes, got %zu", pl.scratch_total, scratch_bytes);
    aim::KArgs ka;
    ka.p = p;
    ka.n_pairs = n_pairs;
    ka.req = static_cast<const aim_request_t *>(d_req);
    ka.patterns = d_pat;
    ka.texts = d_txt;
    ka.res = static_cast<aim_result_t *>(d_res);
    ka.ops = d_ops;
    ka.scratch = (char *)d_scratch;
    ka.scratch_per_wave = pl.scratch_per_wg;
    ka.pool_cap = pl.pool_cap;
    ka.meta_cap = pl.meta_cap;
    ka.ring_slots = pl.ring_slots;
    ka.slot_w = pl.slot_w;
    ka.todo = nullptr;
    ka.dbg_poison_lds = kn.poison_lds >= 0 ? (0x100u | (uint32_t)(kn.poison_lds & 0xff)) : 0u;
    ka.dbg_flags = (uint32_t)kn.dbg_flags;
    ka.dbg_lds_bytes = (uint32_t)pl.lds;
    ka.packedP = fio ? fio->packedP : nullptr;
    ka.packedT = fio ? fio->packedT : nullptr;
    ka.cig = fio ? fio->cig : nullptr;
    ka.runs = fio ? fio->runs : nullptr;
    ka.runs_cap = fio ? fio->runs_cap : 0u;
    ka.cursor = fio ? fio->cursor : nullptr;
    ka.pair_base = 0;
    if (kn.poison_ops >= 0 && d_ops && bt)   // debugging aid: results must not depend on what the ops rows held before (only ops[begin_offset, end_offset) is written)
        HIP_TRY(hipMemsetAsync(d_ops, kn.poison_ops & 0xff, (size_t)n_pairs * 2 * p.read_size, stream));
    switch (pl.kid) {
    case K_WFA_WAVE:
        launch_wfa_wave(bt, red, pl, ka, stream);
        break;
    case K_WFA_LANE:
        ka.scratch_per_wave = 256;
        aim::wfa_lane_launch(p, pl.grid, pl.block, pl.lds, ka, stream);
        break;
    case K_WFA_LANE_PK:
        if (pl.pack_first) {
            // [to-do | flag bits | packed P | packed T | general kernel scratch]
            const size_t npw = aim::packed_row_dwords(p.read_size);
            const size_t flag_b = (((size_t)n_pairs + 31) / 32 * 4 + 255) & ~(size_t)255, arr_b = (((size_t)n_pairs * npw * 4) + 64 + 255) & ~(size_t)255;
            char *base = (char *)d_scratch;
            uint32_t *flags_d = reinterpret_cast<uint32_t *>(base + pl.todo_bytes);
            uint32_t *pkP = reinterpret_cast<uint32_t *>(base + pl.todo_bytes + flag_b), *pkT = reinterpret_cast<uint32_t *>(base + pl.todo_bytes + flag_b + arr_b);
            HIP_TRY(hipMemsetAsync(base, 0, 64, stream));
            HIP_TRY(hipMemsetAsync(flags_d, 0, flag_b, stream));
            const uint64_t threads = (uint64_t)n_pairs * npw;
            hipLaunchKernelGGL(aim::pack_rows_kernel, dim3((unsigned)((threads + 255) / 256), 2), dim3(256), 0, stream, ka, pkP, pkT, flags_d,
                               reinterpret_cast<uint32_t *>(base));
            HIP_TRY(hipGetLastError());
            aim::KArgs kp = ka;
            kp.packedP = pkP;
            kp.packedT = pkT;
            aim::wfa_lane_packed_launch(p, pl.grid, pl.lds, kp, 0u, stream);
            HIP_TRY(hipGetLastError());
            aim::KArgs kb = ka;
            kb.todo = reinterpret_cast<const uint32_t *>(base);
            kb.scratch = base + pl.todo_bytes + pl.pack_bytes;
            kb.scratch_per_wave = pl.scratch_per_wg;
            Plan fb = pl;
            fb.grid = pl.fb_grid;
            fb.lds = pl.fb_lds;
            kb.dbg_lds_bytes = (uint32_t)fb.lds;
            launch_wfa_wave(bt, red, fb, kb, stream);
            break;
        }
        aim::wfa_lane_packed_launch(p, pl.grid, pl.lds, ka, fio->run_slot, stream);
        break;
    case K_WFA_GROUP: {
        // [to-do region | history regions | general kernel scratch]: count zeroed per launch; fast kernel (+ traceback kernel
        // with BACKTRACE) chunk by chunk; then the general kernel drains the to-do list
        HIP_TRY(hipMemsetAsync(d_scratch, 0, 64, stream));
        ka.scratch_per_wave = pl.todo_bytes;   // (diagnostic builds park their stamps behind the to-do region)
        const uint32_t chunk = (bt && pl.chunk_pairs) ? pl.chunk_pairs : n_pairs;
        const size_t rqb = (p.flags & AIM_FLAG_REQ8) ? sizeof(aim_request8_t) : sizeof(aim_request_t);
        const size_t rsb = (p.flags & AIM_FLAG_RES8) ? sizeof(aim_result8_t) : sizeof(aim_result_t);
        const size_t npw = aim::packed_row_dwords(p.read_size);
        const uint32_t nchunks = (n_pairs + chunk - 1) / chunk;
        const bool overlap = bt && nchunks > 1;
        const size_t buf_bytes = overlap ? pl.hist_bytes / 2 : 0;     // two alternating buffers of history regions
        AuxStream *aux = slot_aux;
        std::unique_lock<std::mutex> shared_lock;   // stateless callers: the device's shared instance, locked until the last wait is enqueued
        if (overlap && !aux) {
            SharedAux *sh = shared_aux_for_current_device();
            if (!sh) return fail(AIM_ENODEV, "no current HIP device");
            shared_lock = std::unique_lock<std::mutex>(sh->mu);
            aux = &sh->aux;
        }
        if (overlap) {
            int arc = aux_stream_create(*aux);
            if (arc) return arc;
        }
        uint32_t ci = 0;
        for (uint32_t first = 0; first < n_pairs; first += chunk, ++ci) {
            aim::KArgs kc = ka;
            const int b = (int)(ci & 1u);
            kc.n_pairs = std::min(chunk, n_pairs - first);
            kc.pair_base = first;
            kc.scratch_per_wave = pl.todo_bytes + (overlap ? (size_t)b * buf_bytes : 0);   // where this chunk's history regions start
            kc.req = reinterpret_cast<const aim_request_t *>(static_cast<const char *>(d_req) + (size_t)first * rqb);
            if (d_pat) { kc.patterns = d_pat + (size_t)first * p.read_size; kc.texts = d_txt + (size_t)first * p.read_size; }
            if (d_res) kc.res = reinterpret_cast<aim_result_t *>(static_cast<char *>(d_res) + (size_t)first * rsb);
            if (d_ops) kc.ops = d_ops + (size_t)first * 2 * p.read_size;
            if (ka.packedP) { kc.packedP = ka.packedP + (size_t)first * npw; kc.packedT = ka.packedT + (size_t)first * npw; }
            if (ka.cig) kc.cig = ka.cig + first;
            // a chunk smaller than the plan's grid needs fewer workgroups (the grid stays a multiple of 8)
            uint32_t grid = pl.grid;
            {
                const uint32_t ppw = 64u / (uint32_t)pl.group_g;
                const uint32_t need = ((((kc.n_pairs + ppw - 1) / ppw) + 7u) / 8u) * 8u;
                if (grid > need) grid = need < 8u ? 8u : need;
            }
            if (overlap && ci >= 2) HIP_TRY(hipStreamWaitEvent(stream, aux->walked[b], 0));   // buffer b: its previous chunk has been walked
            aim::wfa_group_launch(p, pl.group_g, pl.gcfg, grid, pl.lds, kc, stream);
            HIP_TRY(hipGetLastError());
            if (overlap) {
                HIP_TRY(hipEventRecord(aux->computed[b], stream));
                HIP_TRY(hipStreamWaitEvent(aux->stream, aux->computed[b], 0));
                aim::wfa_group_tb_launch(p, pl.gcfg, kc.n_pairs, kc, aux->stream);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipEventRecord(aux->walked[b], aux->stream));
            } else if (bt) {
                aim::wfa_group_tb_launch(p, pl.gcfg, kc.n_pairs, kc, stream);
                HIP_TRY(hipGetLastError());
            }
        }
        if (overlap) {   // the caller's stream continues only after both buffers' tracebacks
            HIP_TRY(hipStreamWaitEvent(stream, aux->walked[0], 0));
            HIP_TRY(hipStreamWaitEvent(stream, aux->walked[1], 0));
        }
        aim::KArgs kb = ka;
        kb.packedP = kb.packedT = nullptr;
        kb.cig = nullptr;
        kb.todo = reinterpret_cast<const uint32_t *>(d_scratch);
        kb.scratch = (char *)d_scratch + pl.todo_bytes + pl.hist_bytes;
        kb.scratch_per_wave = pl.scratch_per_wg;
        if (pl.pk) {   // the general kernel reads ASCII rows: expand the to-do pairs' packed rows in place first
            hipLaunchKernelGGL(aim::unpack_todo_rows_kernel, dim3(256), dim3(256), 0, stream, ka, kb.todo, ka.packedP, ka.packedT, const_cast<char *>(d_pat), const_cast<char *>(d_txt));
            HIP_TRY(hipGetLastError());
        }
        Plan fb = pl;
        fb.grid = pl.fb_grid;
        fb.lds = pl.fb_lds;
        kb.dbg_lds_bytes = (uint32_t)fb.lds;
        launch_wfa_wave(bt, red, fb, kb, stream);
        if (pl.emits_runs) {   // the general kernel wrote result_t + ops rows for the to-do pairs: their compact CIGAR
            hipLaunchKernelGGL(aim::cigar_rle_todo_kernel, dim3(256), dim3(64), 0, stream, kb, kb.todo, ka.cig, ka.runs, ka.runs_cap, ka.cursor);
            HIP_TRY(hipGetLastError());
        }
        break;
    }
    case K_DP_LANE:
        aim::dp_lane_launch(p, kn, pl.grid, pl.lds, pl.seq_lds, ka, stream);
        break;
    case K_DP_REG: {
        // [table slabs | to-do region]: count zeroed per launch; the register kernel; then nw_lane_kernel over the pairs it left
        uint32_t *todo_d = reinterpret_cast<uint32_t *>((char *)d_scratch + (pl.scratch_total - pl.todo_bytes));
        HIP_TRY(hipMemsetAsync(todo_d, 0, 64, stream));
        ka.todo = todo_d;
        if (p.algo == AIM_ALGO_SWG) aim::swg_reg_launch(p, pl.grid, pl.lds, ka, stream);
        else aim::nw_reg_launch(p, pl.grid, pl.lds, ka, stream);
        HIP_TRY(hipGetLastError());
        aim::Knobs kq = kn;
        kq.dpl_seq_lds = p.read_size <= 124 ? 2 : 0;
        kq.dpl_no_reg = 0;
        ka.dbg_lds_bytes = (uint32_t)pl.fb_lds;
        aim::dp_lane_launch(p, kq, pl.fb_grid, pl.fb_lds, false, ka, stream);
        break;
    }
    case K_DP_GROUP: {
        // [fallback kernel's scratch | to-do region]: count zeroed per launch; the group kernel; then the literal-capable kernel over the pairs it left
        uint32_t *todo_d = reinterpret_cast<uint32_t *>((char *)d_scratch + (pl.scratch_total - pl.todo_bytes));
        HIP_TRY(hipMemsetAsync(todo_d, 0, 64, stream));
        ka.todo = todo_d;
        aim::dp_group_launch(p, pl.grid, pl.lds, ka, stream);
        HIP_TRY(hipGetLastError());
        ka.dbg_lds_bytes = (uint32_t)pl.fb_lds;
        ka.scratch_per_wave = pl.fb_scratch_per_wg;
        if (pl.strip_k == 0) {
            aim::Knobs kq = kn;
            kq.dpl_seq_lds = 0;
            kq.dpl_no_reg = 0;
            aim::dp_lane_launch(p, kq, pl.fb_grid, pl.fb_lds, false, ka, stream);
        } else {
            ka.pool_cap = pl.pool_cap;
            aim::dp_strip_launch(p, pl.strip_k, pl.fb_grid, pl.fb_block, pl.fb_lds, ka, stream);
        }
        break;
    }
    case K_DP_WAVE:
        aim::dp_wave_launch(p, p.algo == AIM_ALGO_SWG && aim::swg_cell_bytes(p) == 1, pl.grid, pl.block, pl.lds, ka, stream);
        break;
    case K_DP_STRIP:
        // [slabs | lock words | pool tables of the literal path]: the locks are free at every launch
        ka.pool_cap = pl.pool_cap;
        aim::dp_strip_launch(p, pl.strip_k, pl.grid, pl.block, pl.lds, ka, stream);
        break;
    case K_GENASM:
        aim::genasm_launch(p, kn, pl.grid, pl.lds, ka, stream);
        break;
    }
    HIP_TRY(hipGetLastError());
    return AIM_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// device set
// ---------------------------------------------------------------------------
// One buffer set + stream: what a batch needs from H2D to D2H. Slot 0 also serves aim_set_push / launch / pull.
struct aim_slot {
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // h2d b/e, kernel b/e, d2h b/e
    void *d_req = nullptr;   // aim_request_t[] or aim_request8_t[] (AIM_FLAG_REQ8)
    char *d_pat = nullptr, *d_txt = nullptr, *d_ops = nullptr;
    void *d_res = nullptr;   // aim_result_t[] or aim_result8_t[] (AIM_FLAG_RES8)
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    uint32_t *d_packP = nullptr, *d_packT = nullptr;                 // packed input rows
    uint32_t *d_rawidx = nullptr;                                    // side list of raw pairs
    char *d_rawP = nullptr, *d_rawT = nullptr;
    aim_cigar_t *d_cig = nullptr;                                    // compact CIGAR
    uint32_t *d_runs = nullptr, *d_cursor = nullptr, *h_cursor = nullptr;
    void *d_rawreq = nullptr, *d_rawres = nullptr;                   // raw side pass (fused packed batches): the side list as a batch of its own
    char *d_rawops = nullptr;
    aim_cigar_t *d_rawcig = nullptr;
    uint32_t n_pairs = 0;
    uint32_t runs_sent = 0;  // runs of the batch in flight whose D2H copy aim_set_submit already enqueued (slotted run buffer)
    bool pushed = false, launched = false, submitted = false;
    AuxStream aux;           // this slot's second stream + events (chunked wfa_group launches with CIGAR); created on first use
    Plan plan_last;          // the plan the last launch followed (the configure-time plan re-made for the pushed pair count)
    aim_batch_io_t io;       // the batch in flight (aim_set_submit .. aim_set_wait)
};

struct aim_device_ctx {
    int dev = -1;
    std::vector<aim_slot> slots;
    Plan plan;               // made at configure time for max_pairs under the set's knobs and this device's budget
    uint64_t budget = 0;     // scratch bound of one slot of this device, frozen at configure time
    float h2d_ms = 0.f, kernel_ms = 0.f, d2h_ms = 0.f;   // aim_set_submit / aim_set_wait: phase times of this device's batches, summed
};

struct aim_set {
    std::vector<aim_device_ctx> devs;
    aim_params_t params;
    uint32_t max_pairs = 0, max_raw = 0, max_runs = 0;
    bool configured = false;
    aim::Knobs knobs;        // AIM_* switches as read by the last aim_set_configure; launches never re-read the environment
    float h2d_ms = 0.f, kernel_ms = 0.f, d2h_ms = 0.f;
};

namespace {
inline size_t req_size(const aim_params_t &p) { return (p.flags & AIM_FLAG_REQ8) ? sizeof(aim_request8_t) : sizeof(aim_request_t); }
inline size_t res_size(const aim_params_t &p) { return (p.flags & AIM_FLAG_RES8) ? sizeof(aim_result8_t) : sizeof(aim_result_t); }

void free_slot(aim_slot &s)
{
    void *bufs[] = {s.d_req, s.d_pat, s.d_txt, s.d_ops, s.d_res, s.d_scratch, s.d_packP, s.d_packT, s.d_rawidx, s.d_rawP, s.d_rawT,
                    s.d_cig, s.d_runs, s.d_cursor, s.d_rawreq, s.d_rawres, s.d_rawops, s.d_rawcig};
    for (void *b : bufs)
        if (b) (void)hipFree(b);
    if (s.h_cursor) (void)hipHostFree(s.h_cursor);
    for (auto &e : s.ev)
        if (e) (void)hipEventDestroy(e);
    if (s.stream) (void)hipStreamDestroy(s.stream);
    aux_stream_destroy(s.aux);
    s = aim_slot();
}

void free_device_buffers(aim_device_ctx &d)
{
    if (d.dev < 0) return;
    (void)hipSetDevice(d.dev);
    for (auto &s : d.slots) free_slot(s);
    d.slots.clear();
}

// OR of f(i) over i < n != 0, on up to 8 threads for large n (batch-sized host scans sit on the caller's critical path)
template <typename F>
bool any_nonzero(size_t n, F f)
{
    auto part = [&](size_t lo, size_t hi) { uint32_t acc = 0; for (size_t i = lo; i < hi; ++i) acc |= f(i); return acc; };
    const unsigned nt = n >= (1u << 19) ? 8u : 1u;
    if (nt == 1) return part(0, n) != 0;
    uint32_t acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<std::thread> th;
    for (unsigned t = 1; t < nt; ++t) th.emplace_back([&, t] { acc[t] = part(n * t / nt, n * (t + 1) / nt); });
    acc[0] = part(0, n / nt);
    for (auto &x : th) x.join();
    uint32_t all = 0;
    for (unsigned t = 0; t < nt; ++t) all |= acc[t];
    return all != 0;
}

// Every length of a batch against READ_SIZE (host.c:119-123) before anything is enqueued. A 4 M-pair batch is 32-64 MB of
// requests: scanned by one thread this was 2.5 ms of every aim_set_submit -- a fifth of the host CLI's loop at 4e8 pairs/s -- so
// large batches are scanned branch-free by a few threads and only a failing scan is repeated to name the pair.
int check_lengths(const aim_params_t &p, uint32_t n_pairs, const void *requests)
{
    const int rs = p.read_size;
    const bool req8 = p.flags & AIM_FLAG_REQ8;
    auto bad_in = [&](size_t lo, size_t hi) -> uint32_t {
        uint32_t bad = 0;
        if (req8) {
            const aim_request8_t *r = static_cast<const aim_request8_t *>(requests);
            for (size_t i = lo; i < hi; ++i) bad |= (uint32_t)((r[i].pattern_len < 0) | (r[i].text_len < 0) | (r[i].pattern_len > rs) | (r[i].text_len > rs));
        } else {
            const aim_request_t *r = static_cast<const aim_request_t *>(requests);
            for (size_t i = lo; i < hi; ++i) bad |= (uint32_t)((r[i].pattern_len < 0) | (r[i].text_len < 0) | (r[i].pattern_len > rs) | (r[i].text_len > rs));
        }
        return bad;
    };
    uint32_t bad = 0;
    const unsigned nt = n_pairs >= (1u << 19) ? 8u : 1u;
    if (nt == 1) bad = bad_in(0, n_pairs);
    else {
        uint32_t part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        std::vector<std::thread> th;
        for (unsigned t = 1; t < nt; ++t) th.emplace_back([&, t] { part[t] = bad_in((size_t)n_pairs * t / nt, (size_t)n_pairs * (t + 1) / nt); });
        part[0] = bad_in(0, (size_t)n_pairs / nt);
        for (auto &x : th) x.join();
        for (unsigned t = 0; t < nt; ++t) bad |= part[t];
    }
    if (!bad) return AIM_OK;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        const int pl = req8 ? static_cast<const aim_request8_t *>(requests)[i].pattern_len
                            : static_cast<const aim_request_t *>(requests)[i].pattern_len;
        const int tl = req8 ? static_cast<const aim_request8_t *>(requests)[i].text_len
                            : static_cast<const aim_request_t *>(requests)[i].text_len;
        if (pl < 0 || tl < 0 || pl > rs || tl > rs)
            return fail(AIM_EINVAL, "READ LENGTH less than length of the input reads (pair %u)", i);  // host.c:119-123
    }
    return AIM_OK;
}

int status_error(uint32_t idx, int status)
{
    return fail(AIM_EALIGN, "pair idx %u stopped with status %d (%s)", idx, status,
                status == AIM_PAIR_WFA_NO_LINK ? "Backtrace error: No link found during backtrace"
                : status == AIM_PAIR_SWG_NO_OP ? "SWG backtrace. No backtrace operation found"
                                               : "out of memory");
}

// the configure-time plan re-made for this batch's pair count (smaller grids for smaller batches) under the SAME frozen
// knobs and budget; should that ever need more scratch than configure allocated, the configure-time plan itself is
// followed (every kernel tolerates a grid larger than its work)
Plan plan_for_batch(aim_set *set, aim_device_ctx &d, aim_slot &s, uint32_t n_pairs, uint32_t mode)
{
    Plan pl;
    aim::Knobs quiet = set->knobs;
    quiet.plan_debug = false;   // the configure-time plan was printed; per-launch re-plans are not
    int rc = n_pairs ? make_plan(set->params, n_pairs, quiet, d.budget, &pl, mode) : AIM_OK;
    if (rc || !n_pairs || pl.scratch_total > s.scratch_bytes) pl = d.plan;
    return pl;
}

int launch_on_slot(aim_set *set, aim_device_ctx &d, aim_slot &s, uint32_t mode = 0u, FusedIo *fio = nullptr)
{
    const Plan pl = plan_for_batch(set, d, s, s.n_pairs, mode);
    s.plan_last = pl;
    if (fio && pl.emits_runs) {   // slotted run buffer: pair p owns runs[3p, 3p + 3), the cursor starts behind the slots (wfa_lane_packed.hpp)
        fio->run_slot = (pl.kid == K_WFA_LANE_PK && (uint64_t)s.n_pairs * aim::kRunSlot <= fio->runs_cap) ? aim::kRunSlot : 0u;
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)fio->cursor, (int)(s.n_pairs * fio->run_slot), 1, s.stream));
    }
    return launch(pl, set->knobs, set->params, s.n_pairs, s.d_req, s.d_pat, s.d_txt, s.d_res, s.d_ops, s.d_scratch,
                  s.scratch_bytes, s.stream, fio, &s.aux);
}
}  // namespace

extern "C" {

int aim_abi_version(void) { return AIM_ABI_VERSION; }
const char *aim_last_error(void) { return g_err; }

int aim_device_count(int *count)
{
    if (!count) return fail(AIM_EINVAL, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        *count = 0;
        return fail(AIM_ENODEV, "no HIP device: %s", e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    }
    *count = n;
    return AIM_OK;
}

int aim_set_alloc(uint32_t nr_devices, const int *device_ids, aim_set_t **out)
{
    if (!out || nr_devices == 0) return fail(AIM_EINVAL, "bad arguments");
    int have = 0;
    int rc = aim_device_count(&have);
    if (rc) return rc;
    aim_set *s = new aim_set();
    s->devs.resize(nr_devices);
    for (uint32_t i = 0; i < nr_devices; ++i) {
        const int id = device_ids ? device_ids[i] : (int)i;
        if (id < 0 || id >= have) {
            aim_set_free(s);
            return fail(AIM_ENODEV, "device %d not present (%d devices)", id, have);
        }
        s->devs[i].dev = id;
        hipError_t e = hipSetDevice(id);
        if (e != hipSuccess) {
            aim_set_free(s);
            return fail(AIM_ENODEV, "device %d setup failed: %s", id, hipGetErrorString(e));
        }
    }
    *out = s;
    return AIM_OK;
}

int aim_set_nr_devices(const aim_set_t *set, uint32_t *nr)
{
    if (!set || !nr) return fail(AIM_EINVAL, "bad arguments");
    *nr = (uint32_t)set->devs.size();
    return AIM_OK;
}

int aim_set_configure_slots(aim_set_t *set, const aim_params_t *params, uint32_t max_pairs, uint32_t slots, uint32_t max_raw,
                            uint32_t max_runs)
{
    if (!set || !params || max_pairs == 0 || slots == 0 || slots > 4) return fail(AIM_EINVAL, "bad arguments");
    int rc = validate_params(*params);
    if (rc) return rc;
    if (max_runs && !(params->flags & AIM_FLAG_BACKTRACE)) return fail(AIM_EINVAL, "compact CIGAR output needs AIM_FLAG_BACKTRACE");
    const aim::Knobs kn = read_knobs();
    // From here on the set is unconfigured until every device succeeded: a failure half-way must not leave the old
    // max_pairs / params describing buffers that were already freed or re-sized.
    set->configured = false;
    set->max_pairs = 0;
    for (auto &d : set->devs) free_device_buffers(d);
    const size_t rs = (size_t)params->read_size;
    const size_t rowdw = aim::packed_row_dwords(params->read_size);
    auto configure_slot = [&](aim_device_ctx &d, aim_slot &s) -> int {
        HIP_TRY(hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking));
        for (auto &e : s.ev) HIP_TRY(hipEventCreate(&e));
        // the MRAM plan of host.c:215-241, in HBM (+64 B tail slack on the sequence arrays)
        HIP_TRY(hipMalloc(&s.d_req, (size_t)max_pairs * req_size(*params)));
        HIP_TRY(hipMalloc(&s.d_res, (size_t)max_pairs * res_size(*params)));
        HIP_TRY(hipMalloc((void **)&s.d_pat, (size_t)max_pairs * rs + 64));
        HIP_TRY(hipMalloc((void **)&s.d_txt, (size_t)max_pairs * rs + 64));
        if (params->flags & AIM_FLAG_BACKTRACE) HIP_TRY(hipMalloc((void **)&s.d_ops, (size_t)max_pairs * 2 * rs + 64));
        if (max_raw) {
            HIP_TRY(hipMalloc((void **)&s.d_packP, (size_t)max_pairs * rowdw * 4 + 64));
            HIP_TRY(hipMalloc((void **)&s.d_packT, (size_t)max_pairs * rowdw * 4 + 64));
            HIP_TRY(hipMalloc((void **)&s.d_rawidx, (size_t)max_raw * 4));
            HIP_TRY(hipMalloc((void **)&s.d_rawP, (size_t)max_raw * rs + 64));   // (+64 B: the raw side pass hands them to the ASCII kernels as patterns / texts, aim_hip.h tail slack)
            HIP_TRY(hipMalloc((void **)&s.d_rawT, (size_t)max_raw * rs + 64));
            HIP_TRY(hipMalloc(&s.d_rawreq, (size_t)max_raw * req_size(*params)));
            HIP_TRY(hipMalloc(&s.d_rawres, (size_t)max_raw * res_size(*params)));
            if (params->flags & AIM_FLAG_BACKTRACE) HIP_TRY(hipMalloc((void **)&s.d_rawops, (size_t)max_raw * 2 * rs + 64));
            if (max_runs) HIP_TRY(hipMalloc((void **)&s.d_rawcig, (size_t)max_raw * sizeof(aim_cigar_t)));
        }
        if (max_runs) {
            HIP_TRY(hipMalloc((void **)&s.d_cig, (size_t)max_pairs * sizeof(aim_cigar_t)));
            HIP_TRY(hipMalloc((void **)&s.d_runs, (size_t)max_runs * 4));
            HIP_TRY(hipMalloc((void **)&s.d_cursor, 64));
            HIP_TRY(hipHostMalloc((void **)&s.h_cursor, 64, hipHostMallocDefault));
        }
        return AIM_OK;
    };
    auto configure_device = [&](aim_device_ctx &d) -> int {
        HIP_TRY(hipSetDevice(d.dev));
        d.slots.resize(slots);
        for (auto &s : d.slots) {
            int src = configure_slot(d, s);
            if (src) return src;
        }
        // scratch bound of one slot of THIS device, after the fixed buffers exist; halved and re-planned if the allocation
        // still fails
        if (kn.scratch_gb >= 0) {
            d.budget = (uint64_t)(kn.scratch_gb * (double)(1ull << 30)) / slots;
        } else {
            size_t free_b = 0, total_b = 0;
            HIP_TRY(hipMemGetInfo(&free_b, &total_b));
            d.budget = budget_from_free(free_b / slots);
        }
        for (int attempt = 0;; ++attempt) {
            int prc = make_plan(*params, max_pairs, kn, d.budget, &d.plan);
            if (prc) return prc;
            hipError_t e = hipSuccess;
            for (auto &s : d.slots) {
                s.scratch_bytes = d.plan.scratch_total;
                if (d.plan.scratch_total && (e = hipMalloc(&s.d_scratch, d.plan.scratch_total)) != hipSuccess) break;
            }
            if (e == hipSuccess) break;
            (void)hipGetLastError();
            for (auto &s : d.slots) {
                if (s.d_scratch) (void)hipFree(s.d_scratch);
                s.d_scratch = nullptr;
            }
            if (e != hipErrorOutOfMemory || attempt >= 6 || d.budget <= ((uint64_t)1 << 28))
                return fail(e == hipErrorOutOfMemory ? AIM_ENOMEM : AIM_ENODEV, "hipMalloc(scratch, %zu bytes) failed: %s",
                            d.plan.scratch_total, hipGetErrorString(e));
            d.budget /= 2;
        }
        for (auto &s : d.slots) {
            // Debugging aid: AIM_DEBUG_POISON_SCRATCH=<0..255> fills the scratch with that byte. Results must not depend on
            // it (every scratch byte a launch reads must have been written by that launch). On
            // the slot's own stream and completed here: launches run on non-blocking streams that do not synchronise with
            // the null stream.
            if (d.plan.scratch_total && kn.poison_scratch >= 0) {
                HIP_TRY(hipMemsetAsync(s.d_scratch, kn.poison_scratch & 0xff, d.plan.scratch_total, s.stream));
                HIP_TRY(hipStreamSynchronize(s.stream));
            }
            s.n_pairs = 0;
            s.pushed = s.launched = s.submitted = false;
            s.plan_last = d.plan;
        }
        return AIM_OK;
    };
    for (auto &d : set->devs) {
        rc = configure_device(d);
        if (rc) {
            char keep[sizeof g_err];
            memcpy(keep, g_err, sizeof keep);
            for (auto &x : set->devs) free_device_buffers(x);   // nothing half-configured survives
            memcpy(g_err, keep, sizeof keep);
            return rc;
        }
    }
    set->params = *params;
    set->max_pairs = max_pairs;
    set->max_raw = max_raw;
    set->max_runs = max_runs;
    set->knobs = kn;
    set->configured = true;
    return AIM_OK;
}

int aim_set_configure(aim_set_t *set, const aim_params_t *params, uint32_t max_pairs)
{
    return aim_set_configure_slots(set, params, max_pairs, 1, 0, 0);
}

int aim_set_push(aim_set_t *set, uint32_t device, uint32_t n_pairs, const void *requests, const char *patterns,
                 const char *texts)
{
    if (!set || device >= set->devs.size()) return fail(AIM_EINVAL, "bad device index");
    if (!set->configured) return fail(AIM_ESTATE, "aim_set_configure has not been called");
    if (n_pairs > set->max_pairs) return fail(AIM_EINVAL, "n_pairs %u exceeds configured capacity %u", n_pairs, set->max_pairs);
    if (n_pairs && (!requests || !patterns || !texts)) return fail(AIM_EINVAL, "null host buffer");
    aim_device_ctx &d = set->devs[device];
    aim_slot &s = d.slots[0];
    if (s.submitted) return fail(AIM_ESTATE, "slot 0 of device %d holds a submitted batch: aim_set_wait it first", d.dev);
    const size_t rs = (size_t)set->params.read_size;
    int rc = check_lengths(set->params, n_pairs, requests);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(d.dev));
    HIP_TRY(hipEventRecord(s.ev[0], s.stream));
    if (n_pairs) {
        HIP_TRY(hipMemcpyAsync(s.d_req, requests, (size_t)n_pairs * req_size(set->params), hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.d_pat, patterns, (size_t)n_pairs * rs, hipMemcpyHostToDevice, s.stream));
        HIP_TRY(hipMemcpyAsync(s.d_txt, texts, (size_t)n_pairs * rs, hipMemcpyHostToDevice, s.stream));
    }
    HIP_TRY(hipEventRecord(s.ev[1], s.stream));
    s.n_pairs = n_pairs;
    s.pushed = true;
    s.launched = false;
    return AIM_OK;
}

int aim_set_launch(aim_set_t *set)
{
    if (!set || !set->configured) return fail(AIM_ESTATE, "set is not configured");
    for (auto &d : set->devs) {
        aim_slot &s = d.slots[0];
        if (!s.pushed) return fail(AIM_ESTATE, "device %d has no pushed batch", d.dev);
        HIP_TRY(hipSetDevice(d.dev));
        HIP_TRY(hipEventRecord(s.ev[2], s.stream));
        int rc = launch_on_slot(set, d, s);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(s.ev[3], s.stream));
    }
    float worst_h2d = 0.f, worst_k = 0.f;
    for (auto &d : set->devs) {   // DPU_SYNCHRONOUS: wait for every device
        aim_slot &s = d.slots[0];
        HIP_TRY(hipSetDevice(d.dev));
        HIP_TRY(hipStreamSynchronize(s.stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, s.ev[0], s.ev[1]));
        worst_h2d = std::max(worst_h2d, ms);
        HIP_TRY(hipEventElapsedTime(&ms, s.ev[2], s.ev[3]));
        worst_k = std::max(worst_k, ms);
        s.launched = true;
    }
    set->h2d_ms += worst_h2d;
    set->kernel_ms += worst_k;
    return AIM_OK;
}

int aim_set_pull(aim_set_t *set, uint32_t device, void *results, char *ops)
{
    if (!set || device >= set->devs.size()) return fail(AIM_EINVAL, "bad device index");
    if (!set->configured) return fail(AIM_ESTATE, "aim_set_configure has not been called");
    aim_device_ctx &d = set->devs[device];
    aim_slot &s = d.slots[0];
    if (!s.launched) return fail(AIM_ESTATE, "device %d has not been launched", d.dev);
    const bool bt = set->params.flags & AIM_FLAG_BACKTRACE;
    if (s.n_pairs && (!results || (bt && !ops))) return fail(AIM_EINVAL, "null host buffer");
    const size_t rs = (size_t)set->params.read_size;
    HIP_TRY(hipSetDevice(d.dev));
    HIP_TRY(hipEventRecord(s.ev[4], s.stream));
    if (s.n_pairs) {
        HIP_TRY(hipMemcpyAsync(results, s.d_res, (size_t)s.n_pairs * res_size(set->params), hipMemcpyDeviceToHost, s.stream));
        if (bt) HIP_TRY(hipMemcpyAsync(ops, s.d_ops, (size_t)s.n_pairs * 2 * rs, hipMemcpyDeviceToHost, s.stream));
    }
    HIP_TRY(hipEventRecord(s.ev[5], s.stream));
    HIP_TRY(hipStreamSynchronize(s.stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s.ev[4], s.ev[5]));
    set->d2h_ms += ms;
    if (set->params.flags & AIM_FLAG_RES8) return AIM_OK;   // score-only: no status other than AIM_PAIR_OK exists
    const aim_result_t *r = static_cast<const aim_result_t *>(results);
    for (uint32_t i = 0; i < s.n_pairs; ++i)
        if (r[i].status != AIM_PAIR_OK) return status_error(r[i].idx, r[i].status);
    return AIM_OK;
}

int aim_set_submit(aim_set_t *set, uint32_t device, uint32_t slot, const aim_batch_io_t *io)
{
    if (!set || device >= set->devs.size() || !io) return fail(AIM_EINVAL, "bad arguments");
    if (!set->configured) return fail(AIM_ESTATE, "aim_set_configure has not been called");
    aim_device_ctx &d = set->devs[device];
    if (slot >= d.slots.size()) return fail(AIM_EINVAL, "slot %u not configured (%zu slots)", slot, d.slots.size());
    aim_slot &s = d.slots[slot];
    if (s.submitted) return fail(AIM_ESTATE, "slot %u of device %d holds a batch: aim_set_wait it first", slot, d.dev);
    const aim_params_t &p = set->params;
    const uint32_t n = io->n_pairs;
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    const bool packed = io->packed_patterns || io->packed_texts;
    if (n > set->max_pairs) return fail(AIM_EINVAL, "n_pairs %u exceeds configured capacity %u", n, set->max_pairs);
    if (n && !io->requests) return fail(AIM_EINVAL, "null requests");
    if (n && packed && (!io->packed_patterns || !io->packed_texts || !s.d_packP))
        return fail(AIM_EINVAL, "packed batch needs both packed arrays and a set configured with max_raw_pairs > 0");
    if (n && !packed && (!io->patterns || !io->texts)) return fail(AIM_EINVAL, "null sequence rows");
    if (packed && io->n_raw > set->max_raw) return fail(AIM_EINVAL, "n_raw %u exceeds configured capacity %u", io->n_raw, set->max_raw);
    if (packed && io->n_raw && (!io->raw_pairs || !io->raw_patterns || !io->raw_texts)) return fail(AIM_EINVAL, "null raw side list");
    if (io->cigars && (!bt || !s.d_cig || !io->runs)) return fail(AIM_EINVAL, "compact CIGAR needs AIM_FLAG_BACKTRACE, max_runs > 0 and a run buffer");
    if (n && !io->results && !io->cigars) return fail(AIM_EINVAL, "no output buffer");
    if (io->ops && !bt) return fail(AIM_EINVAL, "ops requested without AIM_FLAG_BACKTRACE");
    int rc = check_lengths(p, n, io->requests);
    if (rc) return rc;
    if (packed)
        for (uint32_t j = 0; j < io->n_raw; ++j)
            if (io->raw_pairs[j] >= n) return fail(AIM_EINVAL, "raw_pairs[%u] = %u is outside the batch", j, io->raw_pairs[j]);
    const size_t rs = (size_t)p.read_size;
    const size_t rowb = (size_t)aim::packed_row_dwords(p.read_size) * 4;
    HIP_TRY(hipSetDevice(d.dev));
    s.io = *io;
    s.n_pairs = n;
    s.runs_sent = 0;
    s.pushed = s.launched = false;
    // Everything below only enqueues work on the slot's stream. Should an enqueue fail half-way, the copies already queued
    // still reference the caller's buffers: the stream is drained before the error is returned, so that a failed submit
    // never leaves the slot (or the caller's memory) in flight.
    auto enqueue = [&]() -> int {
        HIP_TRY(hipEventRecord(s.ev[0], s.stream));
        if (n) {
            HIP_TRY(hipMemcpyAsync(s.d_req, io->requests, (size_t)n * req_size(p), hipMemcpyHostToDevice, s.stream));
            if (packed) {
                HIP_TRY(hipMemcpyAsync(s.d_packP, io->packed_patterns, (size_t)n * rowb, hipMemcpyHostToDevice, s.stream));
                HIP_TRY(hipMemcpyAsync(s.d_packT, io->packed_texts, (size_t)n * rowb, hipMemcpyHostToDevice, s.stream));
                if (io->n_raw) {
                    HIP_TRY(hipMemcpyAsync(s.d_rawidx, io->raw_pairs, (size_t)io->n_raw * 4, hipMemcpyHostToDevice, s.stream));
                    HIP_TRY(hipMemcpyAsync(s.d_rawP, io->raw_patterns, (size_t)io->n_raw * rs, hipMemcpyHostToDevice, s.stream));
                    HIP_TRY(hipMemcpyAsync(s.d_rawT, io->raw_texts, (size_t)io->n_raw * rs, hipMemcpyHostToDevice, s.stream));
                }
            } else {
                HIP_TRY(hipMemcpyAsync(s.d_pat, io->patterns, (size_t)n * rs, hipMemcpyHostToDevice, s.stream));
                HIP_TRY(hipMemcpyAsync(s.d_txt, io->texts, (size_t)n * rs, hipMemcpyHostToDevice, s.stream));
            }
        }
        HIP_TRY(hipEventRecord(s.ev[1], s.stream));
        HIP_TRY(hipEventRecord(s.ev[2], s.stream));
        if (n) {
            aim::KArgs ka;
            memset(&ka, 0, sizeof ka);
            ka.p = p;
            ka.n_pairs = n;
            ka.req = static_cast<const aim_request_t *>(s.d_req);
            ka.res = static_cast<aim_result_t *>(s.d_res);
            ka.ops = s.d_ops;
            // Does the alignment kernel of this configuration take the batch as it arrived and deliver what was asked for?
            // (wfa_lane_packed_kernel: packed rows in, {idx, score} or compact CIGAR out.) Then this batch is ONE kernel;
            // otherwise the conversion kernels of batch_io.hpp run around the default-ABI kernel.
            const uint32_t mode = (packed ? MODE_PACKED_IN : 0u) | ((io->cigars && !io->results && !io->ops) ? MODE_RUNS_OUT : 0u);
            const Plan pl = plan_for_batch(set, d, s, n, mode);
            const uint32_t runs_cap = std::min(io->runs_cap, set->max_runs);
            if (packed && !pl.pk) {   // expand into the reference's char[n][READ_SIZE] layout (batch_io.hpp), then run as usual
                const uint64_t threads = (uint64_t)n * (rs / 8);
                hipLaunchKernelGGL(aim::unpack_rows_kernel, dim3((unsigned)((threads + 255) / 256), 2), dim3(256), 0, s.stream, ka, s.d_packP,
                                   s.d_packT, s.d_pat, s.d_txt);
                if (io->n_raw) {
                    const uint64_t rt = (uint64_t)io->n_raw * (rs / 8);
                    hipLaunchKernelGGL(aim::scatter_raw_rows_kernel, dim3((unsigned)((rt + 255) / 256), 2), dim3(256), 0, s.stream, p.read_size,
                                       io->n_raw, s.d_rawidx, s.d_rawP, s.d_rawT, s.d_pat, s.d_txt);
                }
                HIP_TRY(hipGetLastError());
            }
            FusedIo fio;
            if (pl.pk) { fio.packedP = s.d_packP; fio.packedT = s.d_packT; }
            if (pl.emits_runs) { fio.cig = s.d_cig; fio.runs = s.d_runs; fio.cursor = s.d_cursor; fio.runs_cap = runs_cap; }
            rc = launch_on_slot(set, d, s, mode, &fio);
            if (rc) return rc;
            s.runs_sent = pl.emits_runs ? n * fio.run_slot : 0u;   // 0 when the run buffer is bump-allocated: nothing is known before the kernel ran
            if (io->cigars && !pl.emits_runs) {
                HIP_TRY(hipMemsetAsync(s.d_cursor, 0, 4, s.stream));
                hipLaunchKernelGGL(aim::cigar_rle_kernel, dim3((n + 63) / 64), dim3(64), 0, s.stream, ka, s.d_cig, s.d_runs, runs_cap, s.d_cursor);
                HIP_TRY(hipGetLastError());
            }
            if (pl.pk && io->n_raw) {
                // Raw side pass: the pairs of the side list hold a byte outside A/C/G/T (the reference compares raw bytes,
                // host.c:126-127); what the packed kernel computed for them is void. They are aligned as a batch of their own by
                // the ASCII kernels -- requests gathered, the side list's rows are its patterns / texts -- and their results
                // replace the void ones (with the compact CIGAR: their runs are appended to the run buffer).
                const uint32_t nr = io->n_raw;
                const uint32_t rq_dw = (uint32_t)req_size(p) / 4, rs_dw = (uint32_t)res_size(p) / 4;
                auto blocks = [](uint64_t t) { return dim3((unsigned)((t + 255) / 256)); };
                hipLaunchKernelGGL(aim::gather_elems_kernel, blocks((uint64_t)nr * rq_dw), dim3(256), 0, s.stream, (const uint32_t *)s.d_req,
                                   s.d_rawidx, nr, rq_dw, (uint32_t *)s.d_rawreq);
                HIP_TRY(hipGetLastError());
                const Plan rp = plan_for_batch(set, d, s, nr, 0u);
                rc = launch(rp, set->knobs, p, nr, s.d_rawreq, s.d_rawP, s.d_rawT, s.d_rawres, s.d_rawops, s.d_scratch, s.scratch_bytes, s.stream, nullptr, &s.aux);
                if (rc) return rc;
                // (both outputs may be asked for at once -- aim_cigar_t + runs AND result_t [+ ops rows]: each gets its own scatter)
                if (io->cigars) {
                    aim::KArgs kr = ka;
                    kr.n_pairs = nr;
                    kr.res = static_cast<aim_result_t *>(s.d_rawres);
                    kr.ops = s.d_rawops;
                    hipLaunchKernelGGL(aim::cigar_rle_kernel, dim3((nr + 63) / 64), dim3(64), 0, s.stream, kr, s.d_rawcig, s.d_runs, runs_cap, s.d_cursor);
                    hipLaunchKernelGGL(aim::scatter_elems_kernel, blocks((uint64_t)nr * 4), dim3(256), 0, s.stream, (const uint32_t *)s.d_rawcig,
                                       s.d_rawidx, nr, 4u, (uint32_t *)s.d_cig);
                }
                if (!io->cigars || io->results || io->ops) {
                    hipLaunchKernelGGL(aim::scatter_elems_kernel, blocks((uint64_t)nr * rs_dw), dim3(256), 0, s.stream, (const uint32_t *)s.d_rawres,
                                       s.d_rawidx, nr, rs_dw, (uint32_t *)s.d_res);
                    if (bt) {   // default output (result_t + ops rows): the side list's ops rows go home too
                        const uint32_t op_dw = (uint32_t)(2 * rs / 4);
                        hipLaunchKernelGGL(aim::scatter_elems_kernel, blocks((uint64_t)nr * op_dw), dim3(256), 0, s.stream, (const uint32_t *)s.d_rawops,
                                           s.d_rawidx, nr, op_dw, (uint32_t *)s.d_ops);
                    }
                }
                HIP_TRY(hipGetLastError());
            }
        }
        HIP_TRY(hipEventRecord(s.ev[3], s.stream));
        HIP_TRY(hipEventRecord(s.ev[4], s.stream));
        if (n) {
            if (io->cigars) {
                HIP_TRY(hipMemcpyAsync(s.h_cursor, s.d_cursor, 4, hipMemcpyDeviceToHost, s.stream));
                HIP_TRY(hipMemcpyAsync(io->cigars, s.d_cig, (size_t)n * sizeof(aim_cigar_t), hipMemcpyDeviceToHost, s.stream));
                // slotted run buffer (wfa_lane_packed.hpp): the first 3 n runs are the pairs' own slots, known now -- their copy
                // overlaps the next batch instead of waiting in aim_set_wait for the cursor; only runs behind the slots follow there
                if (s.runs_sent) HIP_TRY(hipMemcpyAsync(io->runs, s.d_runs, (size_t)s.runs_sent * 4, hipMemcpyDeviceToHost, s.stream));
            }
            if (io->results) HIP_TRY(hipMemcpyAsync(io->results, s.d_res, (size_t)n * res_size(p), hipMemcpyDeviceToHost, s.stream));
            if (io->ops) HIP_TRY(hipMemcpyAsync(io->ops, s.d_ops, (size_t)n * 2 * rs, hipMemcpyDeviceToHost, s.stream));
        }
        HIP_TRY(hipEventRecord(s.ev[5], s.stream));
        return AIM_OK;
    };
    rc = enqueue();
    if (rc) {
        char keep[sizeof g_err];
        memcpy(keep, g_err, sizeof keep);
        (void)hipStreamSynchronize(s.stream);
        (void)hipGetLastError();
        memcpy(g_err, keep, sizeof keep);
        return rc;
    }
    s.submitted = true;
    return AIM_OK;
}

int aim_set_wait(aim_set_t *set, uint32_t device, uint32_t slot, uint32_t *n_runs)
{
    if (!set || device >= set->devs.size()) return fail(AIM_EINVAL, "bad arguments");
    if (!set->configured) return fail(AIM_ESTATE, "aim_set_configure has not been called");
    aim_device_ctx &d = set->devs[device];
    if (slot >= d.slots.size()) return fail(AIM_EINVAL, "slot %u not configured", slot);
    aim_slot &s = d.slots[slot];
    if (!s.submitted) return fail(AIM_ESTATE, "slot %u of device %d holds no batch", slot, d.dev);
    HIP_TRY(hipSetDevice(d.dev));
    if (getenv("AIM_WAIT_POLL")) {   // A/B (host lanes sharing a device): poll the stream instead of blocking inside the runtime
        hipError_t q;
        while ((q = hipStreamQuery(s.stream)) == hipErrorNotReady) std::this_thread::sleep_for(std::chrono::microseconds(50));
        if (q != hipSuccess) return fail(AIM_ENODEV, "hipStreamQuery failed: %s", hipGetErrorString(q));
    } else
    HIP_TRY(hipStreamSynchronize(s.stream));
    s.submitted = false;
    const aim_batch_io_t &io = s.io;
    uint32_t runs = 0;
    if (io.cigars && s.n_pairs) {   // the run count is only known now: fetch exactly that many
        runs = std::min(std::min(*s.h_cursor, io.runs_cap), set->max_runs);
        if (runs > s.runs_sent) {
            HIP_TRY(hipMemcpyAsync(io.runs + s.runs_sent, s.d_runs + s.runs_sent, (size_t)(runs - s.runs_sent) * 4, hipMemcpyDeviceToHost, s.stream));
            HIP_TRY(hipStreamSynchronize(s.stream));
        }
        s.runs_sent = 0;
    }
    if (n_runs) *n_runs = runs;
    // per DEVICE (devices run concurrently: aim_set_timers reports the slowest device, like aim_set_launch does)
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, s.ev[0], s.ev[1]));
    d.h2d_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, s.ev[2], s.ev[3]));
    d.kernel_ms += ms;
    HIP_TRY(hipEventElapsedTime(&ms, s.ev[4], s.ev[5]));
    d.d2h_ms += ms;
    // Every pair's status (the reference stops at the first DPU fault). 4 M compact CIGARs are 64 MB: one thread scanning them took as
    // long as the batch's transfers (the e2e rate with CIGAR was bound by THIS loop, not by PCIe), so the scan is an OR over a few
    // threads and only a batch that holds a fault is walked to name it.
    if (io.cigars) {
        const aim_cigar_t *cg = io.cigars;
        if (any_nonzero(s.n_pairs, [cg](size_t i) { return (uint32_t)cg[i].status; }))
            for (uint32_t i = 0; i < s.n_pairs; ++i) {
                if (io.cigars[i].status & AIM_CIGAR_OVERFLOW) return fail(AIM_ENOMEM, "run buffer too small (pair idx %u)", io.cigars[i].idx);
                if (io.cigars[i].status != AIM_PAIR_OK) return status_error(io.cigars[i].idx, io.cigars[i].status);
            }
    } else if (io.results && !(set->params.flags & AIM_FLAG_RES8)) {
        const aim_result_t *r = static_cast<const aim_result_t *>(io.results);
        if (any_nonzero(s.n_pairs, [r](size_t i) { return (uint32_t)r[i].status; }))
            for (uint32_t i = 0; i < s.n_pairs; ++i)
                if (r[i].status != AIM_PAIR_OK) return status_error(r[i].idx, r[i].status);
    }
    return AIM_OK;
}

int aim_set_timers(const aim_set_t *set, float *h2d_ms, float *kernel_ms, float *d2h_ms)
{
    if (!set) return fail(AIM_EINVAL, "set is NULL");
    float dh = 0.f, dk = 0.f, dd = 0.f;   // submit / wait path: the slowest device
    for (const auto &d : set->devs) { dh = std::max(dh, d.h2d_ms); dk = std::max(dk, d.kernel_ms); dd = std::max(dd, d.d2h_ms); }
    if (h2d_ms) *h2d_ms = set->h2d_ms + dh;
    if (kernel_ms) *kernel_ms = set->kernel_ms + dk;
    if (d2h_ms) *d2h_ms = set->d2h_ms + dd;
    return AIM_OK;
}

int aim_set_fallback_pairs(aim_set_t *set, uint32_t device, uint32_t *n_fallback)
{
    if (!set || device >= set->devs.size() || !n_fallback) return fail(AIM_EINVAL, "bad arguments");
    if (!set->configured) return fail(AIM_ESTATE, "aim_set_configure has not been called");
    aim_device_ctx &d = set->devs[device];
    aim_slot &s = d.slots[0];
    if (!s.launched) return fail(AIM_ESTATE, "device %d has not been launched", d.dev);
    *n_fallback = 0;
    const Plan &pl = s.plan_last;   // the plan the launch actually followed, not a re-plan
    if ((pl.kid != K_WFA_GROUP && pl.kid != K_DP_REG && pl.kid != K_DP_GROUP && !pl.pack_first) || s.n_pairs == 0) return AIM_OK;   // wfa_lane_kernel has no fallback: it aligns every pair itself
    HIP_TRY(hipSetDevice(d.dev));
    const char *count_at = static_cast<const char *>(s.d_scratch) + ((pl.kid == K_DP_REG || pl.kid == K_DP_GROUP) ? pl.scratch_total - pl.todo_bytes : 0);   // the to-do count
    HIP_TRY(hipMemcpy(n_fallback, count_at, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return AIM_OK;
}

int aim_set_plan_describe(const aim_set_t *set, uint32_t device, char *out, size_t cap)
{
    if (!set || device >= set->devs.size() || !out || cap == 0) return fail(AIM_EINVAL, "bad arguments");
    if (!set->configured) return fail(AIM_ESTATE, "aim_set_configure has not been called");
    const aim_device_ctx &d = set->devs[device];
    const aim_slot &s = d.slots[0];
    describe_plan(s.plan_last, set->params, s.launched ? s.n_pairs : set->max_pairs, d.budget, out, cap);
    return AIM_OK;
}

int aim_set_free(aim_set_t *set)
{
    if (!set) return AIM_OK;
    for (auto &d : set->devs) free_device_buffers(d);
    delete set;
    return AIM_OK;
}

int aim_host_alloc(void **ptr, size_t bytes)
{
    if (!ptr) return fail(AIM_EINVAL, "ptr is NULL");
    HIP_TRY(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return AIM_OK;
}

int aim_host_free(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return AIM_OK;
}

size_t aim_scratch_bytes(const aim_params_t *params, uint32_t n_pairs)
{
    Plan pl;
    const aim::Knobs kn = read_knobs();
    if (!params || make_plan(*params, n_pairs, kn, stateless_budget_bytes(kn), &pl)) return 0;
    return pl.scratch_total;
}

int aim_align_device(const aim_params_t *params, uint32_t n_pairs, const void *d_requests,
                     const char *d_patterns, const char *d_texts, void *d_results, char *d_ops,
                     void *d_scratch, size_t scratch_bytes, void *hip_stream)
{
    if (!params) return fail(AIM_EINVAL, "params is NULL");
    int n = 0;
    int rc = aim_device_count(&n);
    if (rc) return rc;
    const aim::Knobs kn = read_knobs();
    Plan pl;
    rc = make_plan(*params, n_pairs, kn, stateless_budget_bytes(kn), &pl);
    if (rc) return rc;
    return launch(pl, kn, *params, n_pairs, d_requests, d_patterns, d_texts, d_results, d_ops, d_scratch, scratch_bytes,
                  (hipStream_t)hip_stream);
}

int aim_plan_describe(const aim_params_t *params, uint32_t n_pairs, char *out, size_t cap)
{
    if (!params || !out || cap == 0) return fail(AIM_EINVAL, "bad arguments");
    const aim::Knobs kn = read_knobs();
    const uint64_t budget = stateless_budget_bytes(kn);
    Plan pl;
    int rc = make_plan(*params, n_pairs, kn, budget, &pl);
    if (rc) return rc;
    describe_plan(pl, *params, n_pairs, budget, out, cap);
    return AIM_OK;
}

const char *aim_kernel_name(const aim_params_t *params)
{
    Plan pl;
    aim::Knobs kn = read_knobs();
    kn.plan_debug = false;
    if (!params || make_plan(*params, 1u << 20, kn, stateless_budget_bytes(kn), &pl)) return "";
    return kernel_name(pl, *params);
}

// ---------------------------------------------------------------------------
// host-side helpers
// ---------------------------------------------------------------------------
int aim_launcher_sizes(int32_t algo, int32_t read_length, double error, int32_t mismatch, int32_t gap_o,
                       int32_t gap_e, int32_t gap, int32_t *max_score, int32_t *read_size)
{
    if (!max_score || !read_size || read_length <= 0) return fail(AIM_EINVAL, "bad arguments");
    // Python float arithmetic of the launchers == IEEE double here.
    volatile double wrong = (double)read_length * error;
    volatile double a = wrong * (double)mismatch;
    volatile double b = (algo == AIM_ALGO_NW) ? wrong * (double)gap : wrong * (double)(gap_o + gap_e);
    *max_score = (int32_t)std::ceil(a > b ? a : b);
    volatile double t = (double)read_length + wrong;
    t = t + 7.0;
    t = t / 8.0;
    *read_size = (int32_t)std::ceil(t) * 8;
    return AIM_OK;
}

int aim_cigar_format(const char *ops, int32_t begin_offset, int32_t end_offset, char *out, int32_t cap)
{
    if (!ops || !out || cap < 4 || begin_offset < 0) return fail(AIM_EINVAL, "bad arguments");
    int n = 0;
    char last_op = ops[begin_offset];
    int run = 1;
    for (int i = begin_offset + 1; i < end_offset; ++i) {
        if (ops[i] == last_op) {
            ++run;
        } else {
            n += snprintf(out + n, (size_t)(cap - n), "%d%c", run, last_op);
            if (n >= cap - 1) return fail(AIM_EINVAL, "cigar buffer too small");
            last_op = ops[i];
            run = 1;
        }
    }
    n += snprintf(out + n, (size_t)(cap - n), "%d%c\n", run, last_op);
    if (n >= cap) return fail(AIM_EINVAL, "cigar buffer too small");
    return n;
}

int aim_pack_sequence(const char *seq, int32_t len, int32_t read_size, uint32_t *row)
{
    if (!seq || !row || len < 0 || len > read_size) return fail(AIM_EINVAL, "bad arguments");
    const uint32_t dw = aim::packed_row_dwords(read_size);
    uint32_t bad = 0;
    int i = 0;
    for (uint32_t w = 0; w < dw; ++w) {
        uint32_t v = 0;
        const int lim = std::min(len, (int)(w + 1) * 16);
        for (int sh = 0; i < lim; ++i, sh += 2) {
            const unsigned char c = (unsigned char)seq[i];
            const uint32_t code = (c >> 1) & 3u;
            bad |= (uint32_t)(c ^ (unsigned char)"ACTG"[code]);
            v |= code << sh;
        }
        row[w] = v;
    }
    return bad == 0;
}

int aim_pack_batch(const aim_params_t *params, uint32_t n_pairs, const void *requests, const char *patterns, const char *texts,
                   uint32_t *packed_patterns, uint32_t *packed_texts, uint32_t *raw_pairs, char *raw_patterns, char *raw_texts,
                   uint32_t max_raw, uint32_t *n_raw, int threads)
{
    if (!params || !n_raw || (n_pairs && (!requests || !patterns || !texts || !packed_patterns || !packed_texts)))
        return fail(AIM_EINVAL, "bad arguments");
    const int rs = params->read_size;
    if (rs <= 0 || (rs & 7)) return fail(AIM_EINVAL, "read_size must be a positive multiple of 8");
    int rc = check_lengths(*params, n_pairs, requests);
    if (rc) return rc;
    const uint32_t dw = aim::packed_row_dwords(rs);
    const bool req8 = params->flags & AIM_FLAG_REQ8;
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    std::vector<uint8_t> is_raw(n_pairs, 0);
    auto work = [&](int t) {
        const size_t lo = (size_t)n_pairs * t / threads, hi = (size_t)n_pairs * (t + 1) / threads;
        for (size_t i = lo; i < hi; ++i) {
            const int pl = req8 ? static_cast<const aim_request8_t *>(requests)[i].pattern_len : static_cast<const aim_request_t *>(requests)[i].pattern_len;
            const int tl = req8 ? static_cast<const aim_request8_t *>(requests)[i].text_len : static_cast<const aim_request_t *>(requests)[i].text_len;
            const int okp = aim_pack_sequence(patterns + i * rs, pl, rs, packed_patterns + i * dw);
            const int okt = aim_pack_sequence(texts + i * rs, tl, rs, packed_texts + i * dw);
            is_raw[i] = !(okp == 1 && okt == 1);
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    uint32_t n = 0;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        if (!is_raw[i]) continue;
        if (n < max_raw && raw_pairs && raw_patterns && raw_texts) {
            raw_pairs[n] = i;
            memcpy(raw_patterns + (size_t)n * rs, patterns + (size_t)i * rs, (size_t)rs);
            memcpy(raw_texts + (size_t)n * rs, texts + (size_t)i * rs, (size_t)rs);
        }
        ++n;
    }
    *n_raw = n;
    if (n > max_raw) return fail(AIM_ENOMEM, "%u pairs must travel raw, side list holds %u", n, max_raw);
    return AIM_OK;
}

int aim_cigar_format_runs(const uint32_t *runs, uint32_t n_runs, char *out, int32_t cap)
{
    if (!runs || !out || cap < 4 || n_runs == 0) return fail(AIM_EINVAL, "bad arguments");
    int n = 0;
    uint32_t run = runs[0] >> 8;
    char last_op = (char)(runs[0] & 0xff);
    for (uint32_t i = 1; i < n_runs; ++i) {
        const char op = (char)(runs[i] & 0xff);
        if (op == last_op) {
            run += runs[i] >> 8;
        } else {
            n += snprintf(out + n, (size_t)(cap - n), "%u%c", run, last_op);
            if (n >= cap - 1) return fail(AIM_EINVAL, "cigar buffer too small");
            last_op = op;
            run = runs[i] >> 8;
        }
    }
    n += snprintf(out + n, (size_t)(cap - n), "%u%c\n", run, last_op);
    if (n >= cap) return fail(AIM_EINVAL, "cigar buffer too small");
    return n;
}

static inline uint64_t sm64(uint64_t *s)
{
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

int aim_gen_pairs(uint64_t seed, uint64_t first_idx, uint32_t n_pairs, int32_t len, double error,
                  int32_t read_size, aim_request_t *requests, char *patterns, char *texts)
{
    if (!requests || !patterns || !texts || len <= 0 || read_size <= 0) return fail(AIM_EINVAL, "bad arguments");
    const int nedits = (int)std::ceil((double)len * error);
    if (len + nedits > read_size) return fail(AIM_EINVAL, "read_size %d too small for len %d + %d edits", read_size, len, nedits);
    static const char kBase[4] = {'A', 'C', 'G', 'T'};
    auto one = [&](uint32_t i) {
        uint64_t st = seed * 0xD1342543DE82EF95ull + (first_idx + i) * 0x2545F4914F6CDD1Dull + 0x632BE59BD9B4E019ull;
        char *p = patterns + (size_t)i * read_size;
        char *t = texts + (size_t)i * read_size;
        memset(p, 0, (size_t)read_size);
        memset(t, 0, (size_t)read_size);
        for (int j = 0; j < len; ++j) p[j] = kBase[sm64(&st) >> 62];
        memcpy(t, p, (size_t)len);
        int cur = len;
        for (int e = 0; e < nedits; ++e) {
            const uint64_t r = sm64(&st);
            const int kind = (int)((r >> 40) % 3);
            const char b = kBase[(r >> 32) & 3];
            if (kind == 0 && cur > 0) {          // substitute (may re-draw the same base)
                t[(r & 0xffffffffu) % (uint32_t)cur] = b;
            } else if (kind == 1 && cur > 0) {   // delete
                const int pos = (int)((r & 0xffffffffu) % (uint32_t)cur);
                memmove(t + pos, t + pos + 1, (size_t)(cur - pos - 1));
                t[--cur] = 0;
            } else {                             // insert
                const int pos = (int)((r & 0xffffffffu) % (uint32_t)(cur + 1));
                memmove(t + pos + 1, t + pos, (size_t)(cur - pos));
                t[pos] = b;
                ++cur;
            }
        }
        requests[i].pattern_len = len;
        requests[i].text_len = cur;
        requests[i].padding = 0;
        requests[i].idx = (uint32_t)(first_idx + i);
    };
    // pair i depends only on (seed, first_idx + i): long reads (each edit moves up to len bytes) are generated by all cores
    const uint64_t work = (uint64_t)n_pairs * (uint64_t)len * (uint64_t)(nedits + 1);
    unsigned nt = work > (1ull << 28) ? std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u) : 1u;
    if (nt > n_pairs) nt = n_pairs ? n_pairs : 1;
    if (nt <= 1) {
        for (uint32_t i = 0; i < n_pairs; ++i) one(i);
    } else {
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; ++t)
            th.emplace_back([&, t] { for (uint32_t i = t; i < n_pairs; i += nt) one(i); });
        for (auto &x : th) x.join();
    }
    return AIM_OK;
}

}  // extern "C"
