// aim_capi.hip -- C-ABI (include/aim_hip.h) over the gfx950 alignment kernels.
//
// Host side of the drop-in boundary: what host.c does with dpu_alloc /
// dpu_push_xfer / dpu_launch (WFA/DPU-WRAM/host/host.c:186-330) is done here
// with one HIP stream per device, HBM buffers planned per configuration, and
// asynchronous copies.  No CPU fallback exists: every compute entry point
// needs a HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

#include "aim_hip.h"
#include "aim_device.hpp"
#include "wfa_wave.hpp"
#include "wfa_lane.hpp"
#include "wfa_group.hpp"
#include "dp_lane.hpp"
#include "dp_wave.hpp"

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
enum KernelId { K_WFA_WAVE = 0, K_WFA_LANE = 1, K_DP_LANE = 2, K_DP_WAVE = 3, K_WFA_GROUP = 4 };

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
    size_t hist_bytes;      // K_WFA_GROUP + BACKTRACE: per-pair history slabs between to-do region and fallback scratch
    aim::GroupCfg gcfg;     // K_WFA_GROUP
    int group_g;
};

// Upper bound on the HBM scratch one plan may ask for (DP tables, WFA history/pools). AIM_SCRATCH_GB overrides. The
// default is three quarters of the device's free memory, read ONCE per process: every entry point re-plans (aim_scratch_bytes,
// then aim_align_device with the buffer the caller allocated in between), so the bound must not move between calls.
// Plans are need-capped, so this only matters where the tables are huge: config 4 (l = 10 000, 613 MB per pair) runs
// 128 pairs in 5 rounds under a 16 GB bound and in 1 round (4.4x faster) from 80 GB up; a full chip of them (256 pairs,
// one per CU) needs 157 GB, which is why the fraction is 3/4 and not 1/2.
uint64_t scratch_budget_bytes()
{
    if (const char *e = getenv("AIM_SCRATCH_GB")) {
        double gb = atof(e);
        if (gb < 0.25) gb = 0.25;
        return (uint64_t)(gb * (double)(1ull << 30));
    }
    static const uint64_t cached = [] {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b == 0) {
            (void)hipGetLastError();
            return (uint64_t)16 << 30;                       // no device (CPU-only host): planning queries still answer
        }
        return std::max<uint64_t>((uint64_t)free_b / 4 * 3, (uint64_t)1 << 28);
    }();
    return cached;
}

int validate_params(const aim_params_t &p)
{
    if (p.algo != AIM_ALGO_NW && p.algo != AIM_ALGO_SWG && p.algo != AIM_ALGO_WFA)
        return fail(AIM_EINVAL, "unknown algorithm %d", p.algo);
    if (p.read_size <= 0 || (p.read_size & 7)) return fail(AIM_EINVAL, "read_size must be a positive multiple of 8 (got %d)", p.read_size);
    if (p.max_score < 0) return fail(AIM_EINVAL, "max_score must be >= 0");
    // same admission rule as the launchers (run-wfa-pim-wram.py:41-43): m <= 0 and x, g, a > 0
    if (p.algo == AIM_ALGO_NW) {
        if (p.mismatch <= 0 || p.gap_i <= 0 || p.gap_d <= 0) return fail(AIM_EINVAL, "NW penalties must be x, g > 0");
    } else {
        if (p.match > 0 || p.mismatch <= 0 || p.gap_o <= 0 || p.gap_e <= 0)
            return fail(AIM_EINVAL, "Wrong affine gap penalties must be  m <= 0 and g, a, x > 0");
    }
    if (p.algo == AIM_ALGO_WFA && p.read_size >= 16376)
        return fail(AIM_EINVAL, "WFA offsets are int16 (common.h:98-100): read_size must be < 16376");
    if (p.algo != AIM_ALGO_WFA && p.read_size >= 32760)
        return fail(AIM_EINVAL, "NW/SWG cells are int16: read_size must be < 32760");
    return AIM_OK;
}

inline bool p_is_nw(const aim_params_t *p) { return p->algo == AIM_ALGO_NW; }

bool force_wave_kernel()
{
    const char *e = getenv("AIM_FORCE_WAVE");
    return e && e[0] == '1';
}

int make_plan_inner(const aim_params_t &p, uint32_t n_pairs, Plan *pl)
{
    int rc = AIM_OK;
    const uint64_t budget = scratch_budget_bytes();
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    if (p.algo == AIM_ALGO_WFA) {
        const bool lane_ok = !force_wave_kernel() && !pl->no_lane && aim::wfa_lane_supported(p);
        aim::GroupCfg gc;
        int gg = 0;
        uint32_t ggrid = 0;
        size_t glds = 0, ghist = 0;
        const bool no_group = getenv("AIM_NO_GROUP") && getenv("AIM_NO_GROUP")[0] == '1';
        bool group_ok = !lane_ok && !force_wave_kernel() && !pl->no_lane && !no_group &&
                        aim::wfa_group_plan(p, n_pairs, &gc, &gg, &ggrid, &glds, &ghist) && ghist <= budget / 2;
        if (lane_ok || group_ok) {
            // fast path + the general kernel in to-do mode behind it (its plan goes into *pl first)
            Plan fb;
            memset(&fb, 0, sizeof fb);
            fb.no_lane = true;
            rc = make_plan_inner(p, std::min<uint32_t>(n_pairs, 1024u * 64u), &fb);
            if (rc) return rc;
            *pl = fb;
            pl->fb_grid = fb.grid;
            pl->fb_lds = fb.lds;
            if (lane_ok) {
                pl->kid = K_WFA_LANE;
                aim::wfa_lane_plan(p, n_pairs, &pl->grid, &pl->block, &pl->lds);
            } else {
                pl->kid = K_WFA_GROUP;
                pl->gcfg = gc;
                pl->group_g = gg;
                pl->grid = ggrid;
                pl->block = 64;
                pl->lds = glds;
            }
            pl->todo_bytes = aim::wfa_lane_todo_bytes(n_pairs);
            pl->hist_bytes = lane_ok ? 0 : ghist;
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
        } else {
            const uint64_t R = (uint64_t)std::max(p.mismatch, p.gap_o + p.gap_e);
            cap = std::min(full, (R + 2) * 3 * (2 * ms + 3));
        }
        pl->meta_cap = (uint32_t)(ms + 2);
        // LDS ring for the live window of wavefronts: max(x, o+e)+1 slots of slot_w diagonals (M, I, D)
        {
            const uint32_t R = (uint32_t)std::max(p.mismatch, p.gap_o + p.gap_e);
            uint32_t w = 16;
            while (w < 2 * (uint32_t)ms + 3 && w < 128) w *= 2;   // 128: keeps 16 workgroups resident per CU at l = 1000 (measured +9 % over 256)
            if (const char *e = getenv("AIM_WFA_SLOTW")) w = (uint32_t)std::max(16, atoi(e)) & ~15u;
            while (w > 16 && (uint64_t)(R + 1) * 3 * w * 2 > 24 * 1024) w /= 2;
            const bool ring_ok = (uint64_t)(R + 1) * 3 * w * 2 <= 24 * 1024 && !(getenv("AIM_WFA_NO_RING") && getenv("AIM_WFA_NO_RING")[0] == '1');
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
        uint32_t grid = 256 * wg_per_cu;
        const uint32_t need = ((n_pairs + 7u) / 8u) * 8u;
        if (grid > need) grid = std::max(8u, need);
        uint64_t per = (uint64_t)pl->meta_cap * sizeof(aim::WfMeta) + cap * sizeof(int16_t);
        per = (per + 255) & ~255ull;
        while (grid > 512 && per * grid > budget) grid = ((grid / 2) + 7u) & ~7u;
        if (per * grid > budget) {   // shrink the pool; overflow then reports AIM_PAIR_NOMEM like the DPU arena
            const uint64_t meta_b = (uint64_t)pl->meta_cap * sizeof(aim::WfMeta);
            uint64_t avail = budget / grid;
            if (avail < meta_b + 4096) return fail(AIM_ENOMEM, "scratch budget too small for max_score %d", p.max_score);
            cap = (avail - meta_b - 256) / sizeof(int16_t);
            per = (meta_b + cap * sizeof(int16_t) + 255) & ~255ull;
        }
        pl->grid = grid;
        pl->pool_cap = (uint32_t)std::min<uint64_t>(cap, 0x7fffffffu);
        pl->scratch_per_wg = per;
        pl->scratch_total = (size_t)(per * grid);
        return AIM_OK;
    }
    // NW / SWG long reads: one pair per workgroup of 1-12 wavefronts, row-scan, canonical table in per-workgroup HBM scratch
    const bool force_dpw = getenv("AIM_FORCE_DPWAVE") && getenv("AIM_FORCE_DPWAVE")[0] == '1';
    if (p.read_size > 320 || force_dpw) {
        pl->kid = K_DP_WAVE;
        const bool cell8 = p.algo == AIM_ALGO_SWG && aim::swg_cell_bytes(p) == 1;
        return aim::dp_wave_plan(p, n_pairs, budget, cell8, &pl->grid, &pl->block, &pl->lds, &pl->scratch_per_wg, &pl->scratch_total)
                   ? AIM_OK
                   : fail(AIM_ENOMEM, "scratch budget (AIM_SCRATCH_GB) or LDS too small for read_size %d", p.read_size);
    }
    // NW / SWG short reads: one pair per lane, flat DP table in per-wave HBM scratch
    pl->kid = K_DP_LANE;
    return aim::dp_lane_plan(p, n_pairs, budget, &pl->grid, &pl->block, &pl->lds, &pl->scratch_per_wg, &pl->scratch_total,
                             &pl->seq_lds)
               ? AIM_OK
               : fail(AIM_ENOMEM, "scratch budget too small for read_size %d", p.read_size);
}

int make_plan(const aim_params_t &p, uint32_t n_pairs, Plan *pl)
{
    int rc = validate_params(p);
    if (rc) return rc;
    memset(pl, 0, sizeof *pl);
    return make_plan_inner(p, n_pairs, pl);
}

template <bool BT, bool RED>
void launch_wfa_wave(const Plan &pl, const aim::KArgs &ka, hipStream_t s)
{
    if (pl.seq_lds)
        hipLaunchKernelGGL((aim::wfa_wave_kernel<BT, RED, true>), dim3(pl.grid), dim3(64), pl.lds, s, ka);
    else
        hipLaunchKernelGGL((aim::wfa_wave_kernel<BT, RED, false>), dim3(pl.grid), dim3(64), pl.lds, s, ka);
}

int launch(const aim_params_t &p, uint32_t n_pairs, const aim_request_t *d_req, const char *d_pat,
           const char *d_txt, aim_result_t *d_res, char *d_ops, void *d_scratch, size_t scratch_bytes,
           hipStream_t stream)
{
    Plan pl;
    int rc = make_plan(p, n_pairs, &pl);
    if (rc) return rc;
    if (n_pairs == 0) return AIM_OK;
    const bool bt = p.flags & AIM_FLAG_BACKTRACE;
    const bool red = p.flags & AIM_FLAG_REDUCE;
    if (!d_req || !d_pat || !d_txt || !d_res) return fail(AIM_EINVAL, "null device buffer");
    if (bt && !d_ops) return fail(AIM_EINVAL, "AIM_FLAG_BACKTRACE needs an ops buffer");
    if (scratch_bytes < pl.scratch_total || (!d_scratch && pl.scratch_total))
        return fail(AIM_EINVAL, "scratch too small: need %zu bytes, got %zu", pl.scratch_total, scratch_bytes);
    aim::KArgs ka;
    ka.p = p;
    ka.n_pairs = n_pairs;
    ka.req = d_req;
    ka.patterns = d_pat;
    ka.texts = d_txt;
    ka.res = d_res;
    ka.ops = d_ops;
    ka.scratch = (char *)d_scratch;
    ka.scratch_per_wave = pl.scratch_per_wg;
    ka.pool_cap = pl.pool_cap;
    ka.meta_cap = pl.meta_cap;
    ka.ring_slots = pl.ring_slots;
    ka.slot_w = pl.slot_w;
    ka.todo = nullptr;
    switch (pl.kid) {
    case K_WFA_WAVE:
        if (bt && red) launch_wfa_wave<true, true>(pl, ka, stream);
        else if (bt) launch_wfa_wave<true, false>(pl, ka, stream);
        else if (red) launch_wfa_wave<false, true>(pl, ka, stream);
        else launch_wfa_wave<false, false>(pl, ka, stream);
        break;
    case K_WFA_LANE:
    case K_WFA_GROUP: {
        // [to-do region | general kernel scratch]: count zeroed per launch, fast kernel, then the drain
        HIP_TRY(hipMemsetAsync(d_scratch, 0, 64, stream));
        ka.scratch_per_wave = pl.todo_bytes;   // (diagnostic builds park their stamps behind the to-do region)
        if (pl.kid == K_WFA_LANE) aim::wfa_lane_launch(p, pl.grid, pl.block, pl.lds, ka, stream);
        else aim::wfa_group_launch(p, pl.group_g, pl.gcfg, pl.grid, pl.lds, ka, stream);
        HIP_TRY(hipGetLastError());
        aim::KArgs kb = ka;
        kb.todo = reinterpret_cast<const uint32_t *>(d_scratch);
        kb.scratch = (char *)d_scratch + pl.todo_bytes + pl.hist_bytes;
        kb.scratch_per_wave = pl.scratch_per_wg;
        Plan fb = pl;
        fb.grid = pl.fb_grid;
        fb.lds = pl.fb_lds;
        if (bt && red) launch_wfa_wave<true, true>(fb, kb, stream);
        else if (bt) launch_wfa_wave<true, false>(fb, kb, stream);
        else if (red) launch_wfa_wave<false, true>(fb, kb, stream);
        else launch_wfa_wave<false, false>(fb, kb, stream);
        break;
    }
    case K_DP_LANE:
        aim::dp_lane_launch(p, pl.grid, pl.lds, pl.seq_lds, ka, stream);
        break;
    case K_DP_WAVE:
        aim::dp_wave_launch(p, p.algo == AIM_ALGO_SWG && aim::swg_cell_bytes(p) == 1, pl.grid, pl.block, pl.lds, ka, stream);
        break;
    }
    HIP_TRY(hipGetLastError());
    return AIM_OK;
}

}  // namespace

// ---------------------------------------------------------------------------
// device set
// ---------------------------------------------------------------------------
struct aim_device_ctx {
    int dev = -1;
    hipStream_t stream = nullptr;
    hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // h2d b/e, kernel b/e, d2h b/e
    aim_request_t *d_req = nullptr;
    char *d_pat = nullptr, *d_txt = nullptr, *d_ops = nullptr;
    aim_result_t *d_res = nullptr;
    void *d_scratch = nullptr;
    size_t scratch_bytes = 0;
    uint32_t n_pairs = 0;
    bool pushed = false, launched = false;
};

struct aim_set {
    std::vector<aim_device_ctx> devs;
    aim_params_t params;
    uint32_t max_pairs = 0;
    bool configured = false;
    float h2d_ms = 0.f, kernel_ms = 0.f, d2h_ms = 0.f;
};

namespace {
void free_device_buffers(aim_device_ctx &d)
{
    if (d.dev < 0) return;
    (void)hipSetDevice(d.dev);
    if (d.d_req) (void)hipFree(d.d_req);
    if (d.d_pat) (void)hipFree(d.d_pat);
    if (d.d_txt) (void)hipFree(d.d_txt);
    if (d.d_ops) (void)hipFree(d.d_ops);
    if (d.d_res) (void)hipFree(d.d_res);
    if (d.d_scratch) (void)hipFree(d.d_scratch);
    d.d_req = nullptr; d.d_pat = d.d_txt = d.d_ops = nullptr; d.d_res = nullptr; d.d_scratch = nullptr;
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
        aim_device_ctx &d = s->devs[i];
        d.dev = id;
        hipError_t e = hipSetDevice(id);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking);
        for (int k = 0; k < 6 && e == hipSuccess; ++k) e = hipEventCreate(&d.ev[k]);
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

int aim_set_configure(aim_set_t *set, const aim_params_t *params, uint32_t max_pairs)
{
    if (!set || !params || max_pairs == 0) return fail(AIM_EINVAL, "bad arguments");
    Plan pl;
    int rc = make_plan(*params, max_pairs, &pl);
    if (rc) return rc;
    const size_t rs = (size_t)params->read_size;
    for (auto &d : set->devs) {
        free_device_buffers(d);
        HIP_TRY(hipSetDevice(d.dev));
        // the MRAM plan of host.c:215-241, in HBM (+64 B tail slack on the sequence arrays)
        HIP_TRY(hipMalloc((void **)&d.d_req, (size_t)max_pairs * sizeof(aim_request_t)));
        HIP_TRY(hipMalloc((void **)&d.d_res, (size_t)max_pairs * sizeof(aim_result_t)));
        HIP_TRY(hipMalloc((void **)&d.d_pat, (size_t)max_pairs * rs + 64));
        HIP_TRY(hipMalloc((void **)&d.d_txt, (size_t)max_pairs * rs + 64));
        if (params->flags & AIM_FLAG_BACKTRACE) HIP_TRY(hipMalloc((void **)&d.d_ops, (size_t)max_pairs * 2 * rs + 64));
        d.scratch_bytes = pl.scratch_total;
        if (pl.scratch_total) HIP_TRY(hipMalloc(&d.d_scratch, pl.scratch_total));
        // Debugging aid: AIM_DEBUG_POISON_SCRATCH=<0..255> fills the scratch with that byte. Results must not depend on it
        // (every scratch byte a launch reads must have been written by that launch); see tools/poison_probe.py.
        if (pl.scratch_total) {
            if (const char *e = getenv("AIM_DEBUG_POISON_SCRATCH")) {
                // on the set's own stream and completed here: the launches run on a non-blocking stream that does not
                // synchronise with the null stream (a null-stream hipMemset raced with the first launch and produced
                // two false alarms before this was understood)
                HIP_TRY(hipMemsetAsync(d.d_scratch, atoi(e) & 0xff, pl.scratch_total, d.stream));
                HIP_TRY(hipStreamSynchronize(d.stream));
            }
        }
        d.n_pairs = 0;
        d.pushed = d.launched = false;
    }
    set->params = *params;
    set->max_pairs = max_pairs;
    set->configured = true;
    return AIM_OK;
}

int aim_set_push(aim_set_t *set, uint32_t device, uint32_t n_pairs, const aim_request_t *requests,
                 const char *patterns, const char *texts)
{
    if (!set || device >= set->devs.size()) return fail(AIM_EINVAL, "bad device index");
    if (!set->configured) return fail(AIM_ESTATE, "aim_set_configure has not been called");
    if (n_pairs > set->max_pairs) return fail(AIM_EINVAL, "n_pairs %u exceeds configured capacity %u", n_pairs, set->max_pairs);
    if (n_pairs && (!requests || !patterns || !texts)) return fail(AIM_EINVAL, "null host buffer");
    aim_device_ctx &d = set->devs[device];
    const size_t rs = (size_t)set->params.read_size;
    for (uint32_t i = 0; i < n_pairs; ++i) {
        if (requests[i].pattern_len < 0 || requests[i].text_len < 0 || requests[i].pattern_len > (int)rs ||
            requests[i].text_len > (int)rs)
            return fail(AIM_EINVAL, "READ LENGTH less than length of the input reads (pair %u)", i);  // host.c:119-123
    }
    HIP_TRY(hipSetDevice(d.dev));
    HIP_TRY(hipEventRecord(d.ev[0], d.stream));
    if (n_pairs) {
        HIP_TRY(hipMemcpyAsync(d.d_req, requests, (size_t)n_pairs * sizeof(aim_request_t), hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(d.d_pat, patterns, (size_t)n_pairs * rs, hipMemcpyHostToDevice, d.stream));
        HIP_TRY(hipMemcpyAsync(d.d_txt, texts, (size_t)n_pairs * rs, hipMemcpyHostToDevice, d.stream));
    }
    HIP_TRY(hipEventRecord(d.ev[1], d.stream));
    d.n_pairs = n_pairs;
    d.pushed = true;
    d.launched = false;
    return AIM_OK;
}

int aim_set_launch(aim_set_t *set)
{
    if (!set || !set->configured) return fail(AIM_ESTATE, "set is not configured");
    for (auto &d : set->devs) {
        if (!d.pushed) return fail(AIM_ESTATE, "device %d has no pushed batch", d.dev);
        HIP_TRY(hipSetDevice(d.dev));
        HIP_TRY(hipEventRecord(d.ev[2], d.stream));
        int rc = launch(set->params, d.n_pairs, d.d_req, d.d_pat, d.d_txt, d.d_res, d.d_ops, d.d_scratch,
                        d.scratch_bytes, d.stream);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(d.ev[3], d.stream));
    }
    float worst_h2d = 0.f, worst_k = 0.f;
    for (auto &d : set->devs) {   // DPU_SYNCHRONOUS: wait for every device
        HIP_TRY(hipSetDevice(d.dev));
        HIP_TRY(hipStreamSynchronize(d.stream));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, d.ev[0], d.ev[1]));
        worst_h2d = std::max(worst_h2d, ms);
        HIP_TRY(hipEventElapsedTime(&ms, d.ev[2], d.ev[3]));
        worst_k = std::max(worst_k, ms);
        d.launched = true;
    }
    set->h2d_ms += worst_h2d;
    set->kernel_ms += worst_k;
    return AIM_OK;
}

int aim_set_pull(aim_set_t *set, uint32_t device, aim_result_t *results, char *ops)
{
    if (!set || device >= set->devs.size()) return fail(AIM_EINVAL, "bad device index");
    aim_device_ctx &d = set->devs[device];
    if (!d.launched) return fail(AIM_ESTATE, "device %d has not been launched", d.dev);
    const bool bt = set->params.flags & AIM_FLAG_BACKTRACE;
    if (d.n_pairs && (!results || (bt && !ops))) return fail(AIM_EINVAL, "null host buffer");
    const size_t rs = (size_t)set->params.read_size;
    HIP_TRY(hipSetDevice(d.dev));
    HIP_TRY(hipEventRecord(d.ev[4], d.stream));
    if (d.n_pairs) {
        HIP_TRY(hipMemcpyAsync(results, d.d_res, (size_t)d.n_pairs * sizeof(aim_result_t), hipMemcpyDeviceToHost, d.stream));
        if (bt) HIP_TRY(hipMemcpyAsync(ops, d.d_ops, (size_t)d.n_pairs * 2 * rs, hipMemcpyDeviceToHost, d.stream));
    }
    HIP_TRY(hipEventRecord(d.ev[5], d.stream));
    HIP_TRY(hipStreamSynchronize(d.stream));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, d.ev[4], d.ev[5]));
    set->d2h_ms += ms;
    for (uint32_t i = 0; i < d.n_pairs; ++i)
        if (results[i].status != AIM_PAIR_OK)
            return fail(AIM_EALIGN, "pair idx %u stopped with status %d (%s)", results[i].idx, results[i].status,
                        results[i].status == AIM_PAIR_WFA_NO_LINK ? "Backtrace error: No link found during backtrace"
                        : results[i].status == AIM_PAIR_SWG_NO_OP ? "SWG backtrace. No backtrace operation found"
                                                                   : "out of memory");
    return AIM_OK;
}

int aim_set_timers(const aim_set_t *set, float *h2d_ms, float *kernel_ms, float *d2h_ms)
{
    if (!set) return fail(AIM_EINVAL, "set is NULL");
    if (h2d_ms) *h2d_ms = set->h2d_ms;
    if (kernel_ms) *kernel_ms = set->kernel_ms;
    if (d2h_ms) *d2h_ms = set->d2h_ms;
    return AIM_OK;
}

int aim_set_fallback_pairs(aim_set_t *set, uint32_t device, uint32_t *n_fallback)
{
    if (!set || device >= set->devs.size() || !n_fallback) return fail(AIM_EINVAL, "bad arguments");
    aim_device_ctx &d = set->devs[device];
    if (!d.launched) return fail(AIM_ESTATE, "device %d has not been launched", d.dev);
    *n_fallback = 0;
    Plan pl;
    int rc = make_plan(set->params, set->max_pairs, &pl);
    if (rc) return rc;
    if ((pl.kid != K_WFA_LANE && pl.kid != K_WFA_GROUP) || d.n_pairs == 0) return AIM_OK;
    HIP_TRY(hipSetDevice(d.dev));
    HIP_TRY(hipMemcpy(n_fallback, d.d_scratch, sizeof(uint32_t), hipMemcpyDeviceToHost));
    return AIM_OK;
}

int aim_set_free(aim_set_t *set)
{
    if (!set) return AIM_OK;
    for (auto &d : set->devs) {
        if (d.dev < 0) continue;
        (void)hipSetDevice(d.dev);
        free_device_buffers(d);
        for (auto &e : d.ev)
            if (e) (void)hipEventDestroy(e);
        if (d.stream) (void)hipStreamDestroy(d.stream);
    }
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
    if (!params || make_plan(*params, n_pairs, &pl)) return 0;
    return pl.scratch_total;
}

int aim_align_device(const aim_params_t *params, uint32_t n_pairs, const aim_request_t *d_requests,
                     const char *d_patterns, const char *d_texts, aim_result_t *d_results, char *d_ops,
                     void *d_scratch, size_t scratch_bytes, void *hip_stream)
{
    if (!params) return fail(AIM_EINVAL, "params is NULL");
    int n = 0;
    int rc = aim_device_count(&n);
    if (rc) return rc;
    return launch(*params, n_pairs, d_requests, d_patterns, d_texts, d_results, d_ops, d_scratch, scratch_bytes,
                  (hipStream_t)hip_stream);
}

const char *aim_kernel_name(const aim_params_t *params)
{
    Plan pl;
    if (!params || make_plan(*params, 1u << 20, &pl)) return "";
    switch (pl.kid) {
    case K_WFA_WAVE: return "wfa_wave_kernel";
    case K_WFA_LANE: return "wfa_lane_kernel";
    case K_WFA_GROUP: return "wfa_group_kernel";
    case K_DP_LANE: return p_is_nw(params) ? "nw_lane_kernel" : "swg_lane_kernel";
    case K_DP_WAVE: return "dp_wave_kernel";
    }
    return "";
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
    for (uint32_t i = 0; i < n_pairs; ++i) {
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
    }
    return AIM_OK;
}

}  // extern "C"
