// batch_io.hpp -- the wire formats either side of the alignment kernels:
//
//   * unpack_rows_kernel / scatter_raw_rows_kernel: a PACKED batch (2 bits per base + a side list of pairs that hold a
//     byte outside A/C/G/T, which travel as raw rows) is expanded in HBM into the reference's own layout --
//     char[n][READ_SIZE] ASCII rows (WFA/DPU-WRAM/host/host.c:126-127, 258-268) -- so every alignment kernel runs
//     unchanged and bit-exact.  What crosses PCIe is READ_SIZE/4 bytes per sequence instead of READ_SIZE.
//   * cigar_rle_kernel: the run-length encoding edit_cigar_print does on the host (host.c:69-89) done on the device
//     over ops[begin_offset, end_offset), so that ~3 runs per pair cross PCIe instead of 2*READ_SIZE op bytes.
//
// Both are pure data movement: HBM-bound, coalesced, no LDS.
#pragma once

#include "aim_device.hpp"

namespace aim {

// ---- packed batches --------------------------------------------------------------------------------------------------
// code = (ascii >> 1) & 3: A 0, C 1, T 2, G 3 (the mapping the kernels use internally); base i of a sequence sits at bits
// [2*(i%16), 2*(i%16)+1] of dword i/16 of its row; a row is ceil(READ_SIZE/16) dwords.
__host__ __device__ inline uint32_t packed_row_dwords(int read_size) { return (uint32_t)(read_size + 15) / 16u; }

// One thread expands 8 bases (16 packed bits -> 8 ASCII bytes); bytes at or beyond the sequence length are zero, like
// the parser's rows.  grid.y: 0 = patterns, 1 = texts.
__global__ __launch_bounds__(256) void unpack_rows_kernel(KArgs a, const uint32_t *packedP, const uint32_t *packedT, char *outP, char *outT)
{
    const int rs = a.p.read_size;
    const uint32_t per_row = (uint32_t)rs / 8u;                    // 8-byte pieces per ASCII row
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)a.n_pairs * per_row) return;
    const uint32_t pair = (uint32_t)(t / per_row), piece = (uint32_t)(t - (uint64_t)pair * per_row);
    const bool is_text = blockIdx.y != 0;
    const aim_request_t rq = load_request(a, pair);
    const int len = is_text ? rq.text_len : rq.pattern_len;
    const uint16_t *src = reinterpret_cast<const uint16_t *>((is_text ? packedT : packedP) + (uint64_t)pair * packed_row_dwords(rs));
    const uint32_t bits = src[piece];
    // spread the eight 2-bit codes into bytes, then look "ACTG"[code] up with v_perm
    const uint32_t lo = (bits & 3u) | ((bits & 0xCu) << 6) | ((bits & 0x30u) << 12) | ((bits & 0xC0u) << 18);
    const uint32_t hb = bits >> 8;
    const uint32_t hi = (hb & 3u) | ((hb & 0xCu) << 6) | ((hb & 0x30u) << 12) | ((hb & 0xC0u) << 18);
    uint32_t w0 = __builtin_amdgcn_perm(0u, 0x47544341u, lo);
    uint32_t w1 = __builtin_amdgcn_perm(0u, 0x47544341u, hi);
    const int rem = len - (int)piece * 8;                          // valid bytes of this piece
    if (rem < 8) {
        const uint64_t keep = rem <= 0 ? 0ull : ((1ull << (8 * rem)) - 1ull);
        w0 &= (uint32_t)keep;
        w1 &= (uint32_t)(keep >> 32);
    }
    uint2 *dst = reinterpret_cast<uint2 *>((is_text ? outT : outP) + (uint64_t)pair * rs) + piece;
    *dst = make_uint2(w0, w1);
}

// Side list: raw_pairs[j] is the batch index of the j-th pair that could not be packed; its ASCII rows are copied verbatim.
__global__ __launch_bounds__(256) void scatter_raw_rows_kernel(int read_size, uint32_t n_raw, const uint32_t *raw_pairs, const char *rawP,
                                                               const char *rawT, char *outP, char *outT)
{
    const uint32_t per_row = (uint32_t)read_size / 8u;
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n_raw * per_row) return;
    const uint32_t j = (uint32_t)(t / per_row), piece = (uint32_t)(t - (uint64_t)j * per_row);
    const uint32_t pair = raw_pairs[j];
    const bool is_text = blockIdx.y != 0;
    const uint2 v = reinterpret_cast<const uint2 *>((is_text ? rawT : rawP) + (uint64_t)j * read_size)[piece];
    reinterpret_cast<uint2 *>((is_text ? outT : outP) + (uint64_t)pair * read_size)[piece] = v;
}

// The inverse of unpack_rows_kernel: ASCII rows -> packed rows on the device, for configurations only the packed lane kernel
// covers (READ_SIZE other than 80 / 112) when the batch arrived as ASCII rows (the default ABI). One thread packs 16 bases
// (one output dword) and validates them by decoding back; a pair with a byte outside A/C/G/T inside either sequence is
// appended ONCE (flag bit per pair) to the to-do list {count @0, pair ids @16..} that the general kernel drains afterwards
// over the ASCII rows -- the same hand-over wfa_group_kernel uses. grid.y: 0 = patterns, 1 = texts.
typedef uint32_t pk_in_u32x4 __attribute__((ext_vector_type(4), aligned(4)));
__global__ __launch_bounds__(256) void pack_rows_kernel(KArgs a, uint32_t *packedP, uint32_t *packedT, uint32_t *flag_bits, uint32_t *todo)
{
    const int rs = a.p.read_size;
    const uint32_t npw = packed_row_dwords(rs);
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)a.n_pairs * npw) return;
    const uint32_t pair = (uint32_t)(t / npw), j = (uint32_t)(t - (uint64_t)pair * npw);
    const bool is_text = blockIdx.y != 0;
    const aim_request_t rq = load_request(a, pair);
    const int len = is_text ? rq.text_len : rq.pattern_len;
    const char *row = (is_text ? a.texts : a.patterns) + (uint64_t)pair * rs + 16u * j;   // (the arrays carry >= 16 B of tail slack)
    const pk_in_u32x4 v = *reinterpret_cast<const pk_in_u32x4 *>(row);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t out = 0, bad = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int rem = len - (int)(16u * j + 4u * i);
        const uint32_t mask = rem >= 4 ? ~0u : (rem <= 0 ? 0u : ((1u << (8 * rem)) - 1u));
        const uint32_t av = w[i] & mask;
        const uint32_t c = (av >> 1) & 0x03030303u & mask;
        const uint32_t rec = __builtin_amdgcn_perm(0u, 0x47544341u, c) & mask;
        bad |= rec ^ av;
        out |= __builtin_amdgcn_udot4(c, 0x40100401u, 0u, false) << (8 * i);
    }
    (is_text ? packedT : packedP)[t] = out;
    if (bad) {
        const uint32_t bit = 1u << (pair & 31u);
        const uint32_t old = atomicOr(&flag_bits[pair >> 5], bit);
        if (!(old & bit)) {
            const uint32_t slot = atomicAdd(&todo[0], 1u);
            todo[16 + slot] = pair;
        }
    }
}

// Raw side pass of a packed batch whose alignment kernel read the packed rows itself (aim_capi.hip): the pairs of the side
// list are aligned by the ASCII kernels as a small batch of their own -- requests gathered, results scattered back.
// Elements are `dw` dwords (requests 2 / 4, results 2 / 6, aim_cigar_t 4).
__global__ __launch_bounds__(256) void gather_elems_kernel(const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t dw, uint32_t *dst)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n * dw) return;
    const uint32_t j = (uint32_t)(t / dw), w = (uint32_t)(t - (uint64_t)j * dw);
    dst[t] = src[(uint64_t)idx[j] * dw + w];
}
__global__ __launch_bounds__(256) void scatter_elems_kernel(const uint32_t *src, const uint32_t *idx, uint32_t n, uint32_t dw, uint32_t *dst)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (uint64_t)n * dw) return;
    const uint32_t j = (uint32_t)(t / dw), w = (uint32_t)(t - (uint64_t)j * dw);
    dst[(uint64_t)idx[j] * dw + w] = src[t];
}

// ---- compact CIGAR -------------------------------------------------------------------------------------------------
// One pair per lane.  Pass 1 counts the runs of ops[begin, end) word-wise (a run starts where a byte differs from its
// predecessor), the wavefront reserves its runs with ONE atomic add on the batch cursor, pass 2 writes them:
// run = (length << 8) | op character.  Placement in the run buffer depends on scheduling, content does not: each header
// carries its own offset.
__device__ __forceinline__ uint32_t nonzero_byte_mask(uint32_t x)   // 0x80 in every non-zero byte
{
    return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}

// one wavefront of 64 pairs: `pair` is this lane's pair (any value when !active); all 64 lanes must call
__device__ __forceinline__ void cigar_rle_wave(const KArgs &a, uint32_t pair, bool active, int lane, aim_cigar_t *hdr, uint32_t *runs, uint32_t runs_cap,
                                               uint32_t *cursor)
{
    const int rs = a.p.read_size;
    aim_result_t r;
    r.begin_offset = 0; r.end_offset = 0; r.score = 0; r.status = AIM_PAIR_OK; r.idx = 0; r.max_operations = 0;
    if (active) r = a.res[pair];
    // edit_cigar_print (host.c:69-89) always prints ops[begin_offset] as a first run, then extends / splits up to end_offset
    int b = r.begin_offset < 0 ? 0 : r.begin_offset;
    int e = r.end_offset > 2 * rs ? 2 * rs : r.end_offset;
    if (e <= b) e = b + 1;
    if (b >= 2 * rs) { b = 2 * rs - 1; e = 2 * rs; }
    const bool walk = active && r.status == AIM_PAIR_OK;
    const uint32_t *row = reinterpret_cast<const uint32_t *>(a.ops + (uint64_t)pair * 2 * rs);
    // boundary mask of word w: bit 8j+7 set <=> byte 4w+j differs from byte 4w+j-1, restricted to positions in (b, e)
    auto boundaries = [&](int w, uint32_t cur, uint32_t prev) -> uint32_t {
        uint32_t m = nonzero_byte_mask(cur ^ __builtin_amdgcn_alignbyte(cur, prev, 3u));   // cur ^ (bytes shifted up by one, prev's top byte in)
        const int lo = b + 1 - 4 * w, hi = e - 4 * w;                                     // keep byte j iff lo <= j < hi
        if (lo > 0) m &= lo >= 4 ? 0u : (0xffffffffu << (8 * lo));
        if (hi < 4) m &= hi <= 0 ? 0u : (0xffffffffu >> (8 * (4 - hi)));
        return m;
    };
    uint32_t n_runs = 0;
    if (walk) {
        n_runs = 1;
        uint32_t prev = 0;
        for (int w = b >> 2; w <= (e - 1) >> 2; ++w) {
            const uint32_t cur = row[w];
            n_runs += (uint32_t)__builtin_popcount(boundaries(w, cur, prev));
            prev = cur;
        }
    }
    // wave-level exclusive prefix sum of n_runs (DPP row scans + row broadcasts), one atomic for the wavefront
    uint32_t incl = n_runs;
#define AIM_RLE_SCAN(ctrl, rmask) incl += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)incl, ctrl, rmask, 0xf, false)
    AIM_RLE_SCAN(0x111, 0xf);   // row_shr:1
    AIM_RLE_SCAN(0x112, 0xf);   // row_shr:2
    AIM_RLE_SCAN(0x114, 0xf);   // row_shr:4
    AIM_RLE_SCAN(0x118, 0xf);   // row_shr:8
    AIM_RLE_SCAN(0x142, 0xa);   // row_bcast:15
    AIM_RLE_SCAN(0x143, 0xc);   // row_bcast:31
#undef AIM_RLE_SCAN
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, kWave - 1);
    uint32_t base = 0;
    if (lane == 0 && total) base = atomicAdd(cursor, total);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    const uint32_t off = base + incl - n_runs;
    // aim_cigar_t carries n_runs in 16 bits and a run its length in 24: a pair beyond either (GenASM long reads with > 65 535
    // runs, l >~ 330 kb at e = 10 %) reports AIM_CIGAR_OVERFLOW -- the caller then gathers ops rows -- instead of a truncated count
    const bool fits = off + n_runs <= runs_cap && n_runs <= 0xffffu && (e - b) < (1 << 24);
    if (walk && fits) {
        uint32_t prev = 0, at = off;
        int start = b;
        for (int w = b >> 2; w <= (e - 1) >> 2; ++w) {
            const uint32_t cur = row[w];
            uint32_t m = boundaries(w, cur, prev);
            while (m) {
                const int j = __builtin_ctz(m) >> 3;              // byte index of the boundary inside the word
                const int pos = 4 * w + j;
                const uint32_t op = (j ? (cur >> (8 * (j - 1))) : (prev >> 24)) & 0xffu;   // the byte before the boundary
                runs[at++] = ((uint32_t)(pos - start) << 8) | op;
                start = pos;
                m &= m - 1;
            }
            prev = cur;
        }
        const uint32_t lw = row[(e - 1) >> 2];
        runs[at] = ((uint32_t)(e - start) << 8) | ((lw >> (8 * ((e - 1) & 3))) & 0xffu);
    }
    if (active) {
        aim_cigar_t h;
        h.idx = r.idx;
        h.score = r.score;
        h.run_offset = off;
        h.n_runs = (uint16_t)((walk && fits) ? n_runs : 0u);
        h.status = (uint16_t)((uint32_t)r.status | ((walk && !fits) ? AIM_CIGAR_OVERFLOW : 0u));
        hdr[pair] = h;
    }
}

__global__ __launch_bounds__(64) void cigar_rle_kernel(KArgs a, aim_cigar_t *hdr, uint32_t *runs, uint32_t runs_cap, uint32_t *cursor)
{
    const int lane = threadIdx.x;
    const uint32_t pair = blockIdx.x * kWave + lane;
    cigar_rle_wave(a, pair, pair < a.n_pairs, lane, hdr, runs, runs_cap, cursor);
}

// The same over a device-side list of pairs (the to-do list of wfa_group_kernel: {count @0, pair ids @16..}, wfa_lane.hpp
// LANE_TODO_*): the general kernel aligned those pairs into result_t + ops rows; a fused batch wants their compact CIGAR.
__global__ __launch_bounds__(64) void cigar_rle_todo_kernel(KArgs a, const uint32_t *todo, aim_cigar_t *hdr, uint32_t *runs, uint32_t runs_cap, uint32_t *cursor)
{
    const int lane = threadIdx.x;
    const uint32_t count = todo[0];
    for (uint32_t base = blockIdx.x * kWave; base < count; base += gridDim.x * kWave) {
        const uint32_t i = base + lane;
        const bool active = i < count;
        const uint32_t pair = active ? todo[16 + i] : 0u;
        cigar_rle_wave(a, pair, active, lane, hdr, runs, runs_cap, cursor);
    }
}

// ... and the expansion of exactly those pairs' packed rows into the ASCII rows the general kernel reads (8 bases per thread).
__global__ __launch_bounds__(256) void unpack_todo_rows_kernel(KArgs a, const uint32_t *todo, const uint32_t *packedP, const uint32_t *packedT, char *outP, char *outT)
{
    const int rs = a.p.read_size;
    const uint32_t per_row = (uint32_t)rs / 8u;
    const uint64_t total = (uint64_t)todo[0] * per_row * 2u;
    for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
        const bool is_text = t & 1u;
        const uint64_t u = t >> 1;
        const uint32_t pair = todo[16 + (uint32_t)(u / per_row)], piece = (uint32_t)(u % per_row);
        const aim_request_t rq = load_request(a, pair);
        const int len = is_text ? rq.text_len : rq.pattern_len;
        const uint16_t *src = reinterpret_cast<const uint16_t *>((is_text ? packedT : packedP) + (uint64_t)pair * packed_row_dwords(rs));
        const uint32_t bits = src[piece];
        const uint32_t lo = (bits & 3u) | ((bits & 0xCu) << 6) | ((bits & 0x30u) << 12) | ((bits & 0xC0u) << 18);
        const uint32_t hb = bits >> 8;
        const uint32_t hi = (hb & 3u) | ((hb & 0xCu) << 6) | ((hb & 0x30u) << 12) | ((hb & 0xC0u) << 18);
        uint32_t w0 = __builtin_amdgcn_perm(0u, 0x47544341u, lo);
        uint32_t w1 = __builtin_amdgcn_perm(0u, 0x47544341u, hi);
        const int rem = len - (int)piece * 8;
        if (rem < 8) {
            const uint64_t keep = rem <= 0 ? 0ull : ((1ull << (8 * rem)) - 1ull);
            w0 &= (uint32_t)keep;
            w1 &= (uint32_t)(keep >> 32);
        }
        reinterpret_cast<uint2 *>((is_text ? outT : outP) + (uint64_t)pair * rs)[piece] = make_uint2(w0, w1);
    }
}

}  // namespace aim
