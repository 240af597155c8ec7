// aim_device.hpp -- shared device-side definitions for the gfx950 alignment kernels.
// CDNA4 only: 64-lane wavefronts are assumed everywhere.
#pragma once

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
    const uint32_t *todo; // nullptr: units are pairs 0..n_pairs-1; else {count @0, pair ids @16..} written by wfa_lane
};

// XCD-aware work distribution: workgroups are dealt round-robin over the 8 XCDs
// (blockIdx % 8 labels the XCD group), so give each group one contiguous slice
// of the batch: neighbouring pairs share 128-B lines and then share an L2.
// Requires gridDim.x % 8 == 0.  Returns false when this (block, iteration) has no work.
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

__device__ __forceinline__ int wave_min_i32(int v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_xor(v, off, kWave);
        v = o < v ? o : v;
    }
    return v;
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
