/*
 * aim_hip.h -- C-ABI of the MI355X alignment engine (libaim_hip.so).
 *
 * This is the drop-in boundary for AIM's per-pair alignment path.  The
 * reference host program (safaad/aim, e.g. WFA/DPU-WRAM/host/host.c) talks to
 * the UPMEM SDK through nine calls and a byte-layout ABI in MRAM; every entry
 * point below names the reference call site it replaces.  Plain C: pointers,
 * sizes, POD structs; no C++ or torch types cross this boundary.
 *
 * All functions return AIM_OK (0) or a negative AIM_E* code; aim_last_error()
 * returns a thread-local human readable message for the last failure.
 * There is no CPU fallback: without a usable HIP device every compute entry
 * point fails with AIM_ENODEV.
 */
#ifndef AIM_HIP_H
#define AIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AIM_ABI_VERSION 2

/* ---- error codes ------------------------------------------------------- */
#define AIM_OK 0
#define AIM_EINVAL (-1)  /* bad argument / unsupported configuration          */
#define AIM_ENODEV (-2)  /* no HIP device / HIP runtime failure                */
#define AIM_ENOMEM (-3)  /* host or device allocation failed                  */
#define AIM_ESTATE (-4)  /* call sequence violated (e.g. launch before push)  */
#define AIM_EALIGN (-5)  /* at least one pair hit a reference abort condition;
                            see aim_result_t.status                           */

/* ---- algorithm selection (one per reference sub-project) ---------------- */
#define AIM_ALGO_NW 0  /* NW/DPU-{WRAM,MRAM}   nw_compute      nw.c:109-153   */
#define AIM_ALGO_SWG 1 /* SWG/DPU-{WRAM,MRAM}  swg_compute     swg.c:121-171  */
#define AIM_ALGO_WFA 2 /* WFA/DPU-{WRAM,MRAM}  affine_wfa_compute wfa.c:342-379 */
#define AIM_ALGO_GENASM 3 /* aim-genasm (BASELINE config 5): bit-vector edit distance + windowed traceback for long reads.
                             PARITY UNPINNED: the reference tree holds only an un-pinned, empty submodule
                             (.gitmodules:1-3); this implements the published GenASM algorithm (MICRO 2020) as restated
                             in oracle/genasm_oracle.c.  score = edit distance of the reported alignment; penalties and
                             max_score are ignored; ops are written forward (begin_offset = 0). */

/* ---- flags: the reference's compile-time -D switches, now run time ------ */
#define AIM_FLAG_BACKTRACE 0x1u /* -DBACKTRACE (run-*-pim-*.py -b)            */
#define AIM_FLAG_REDUCE 0x2u    /* -DREDUCE = WFA-adaptive (run-wfa-*.py -r)  */
#define AIM_FLAG_SWG_W16 0x4u   /* force int16 SWG cells (= SWG/DPU-MRAM,
                                   SWG/DPU-MRAM/common/common.h:91); default is
                                   the WRAM rule: int8 iff MAX_SCORE < 127
                                   (SWG/DPU-WRAM/common/common.h:71-75)       */

/* Opt-in compact I/O layouts (round 2; the 16-B / 24-B structs below stay the default ABI):
 *  AIM_FLAG_REQ8  requests[] are aim_request8_t = the reference's own WFA request_t, 8 B
 *                 (WFA/DPU-WRAM/common/common.h:172-177: int16 pattern_len, int16 text_len, uint32 idx);
 *  AIM_FLAG_RES8  results[] are aim_result8_t {idx, score}, 8 B -- exactly what the reference host prints in
 *                 score-only mode (host.c:339-341).  Not valid with AIM_FLAG_BACKTRACE (the CIGAR needs the offsets).
 *                 There is no status field: without BACKTRACE the only per-pair failure that exists is AIM_PAIR_NOMEM, which
 *                 the launch plans rule out (a score-only launch is refused with AIM_ENOMEM rather than given a window that
 *                 could overflow); should a kernel ever report one all the same, its score reads AIM_SCORE_FAILED.
 * Both apply to every entry point that takes requests / results (aim_set_push / aim_set_pull / aim_align_device). */
#define AIM_FLAG_REQ8 0x8u
#define AIM_FLAG_RES8 0x10u
#define AIM_SCORE_FAILED ((int32_t)0x80000000) /* aim_result8_t.score of a pair that stopped with a status (see above) */

/* Replaces the -D macro set the launchers pass to make
 * (WFA/DPU-WRAM/run-wfa-pim-wram.py:128-131; common.h:63-89). */
typedef struct aim_params {
    int32_t algo;      /* AIM_ALGO_*                                          */
    int32_t match;     /* MATCH     (SWG only; NW/WFA ignore it like the ref) */
    int32_t mismatch;  /* MISMATCH                                            */
    int32_t gap_o;     /* GAP_O     (SWG, WFA)                                */
    int32_t gap_e;     /* GAP_E     (SWG, WFA)                                */
    int32_t gap_i;     /* GAP_I     (NW)                                      */
    int32_t gap_d;     /* GAP_D     (NW)                                      */
    int32_t max_score; /* MAX_SCORE (WFA: score cap; SWG: "+infinity" value)  */
    int32_t read_size; /* READ_SIZE: row stride of patterns/texts, multiple of 8 */
    uint32_t flags;    /* AIM_FLAG_*                                          */
} aim_params_t;

/* Per-pair descriptor: byte-compatible with the NW/SWG request_t
 * (NW/DPU-WRAM/common/common.h:114-120).  The WFA variant of the reference
 * uses int16 lengths (WFA/DPU-WRAM/common/common.h:172-177); a binding widens
 * them when filling this struct. */
typedef struct aim_request {
    int32_t pattern_len;
    int32_t text_len;
    int32_t padding;
    uint32_t idx; /* global pair index, echoed into the result */
} aim_request_t;

/* AIM_FLAG_REQ8: byte-compatible with the WFA request_t (WFA/DPU-WRAM/common/common.h:172-177). */
typedef struct aim_request8 {
    int16_t pattern_len;
    int16_t text_len;
    uint32_t idx;
} aim_request8_t;

/* Per-pair result: byte-compatible with the NW/SWG result_t
 * (NW/DPU-WRAM/common/common.h:122-130); the reference's unused `padding`
 * word carries the per-pair status. */
#define AIM_PAIR_OK 0
#define AIM_PAIR_WFA_NO_LINK 1 /* wfa_backtracing.c:321-325: ref prints + exit(1) */
#define AIM_PAIR_SWG_NO_OP 2   /* swg.c:99-104: ref prints + exit(1)             */
#define AIM_PAIR_NOMEM 3       /* dpu_allocator_wram.c:19-23: ref prints + exit(1) */
typedef struct aim_result {
    int32_t max_operations; /* plen + tlen                                    */
    int32_t begin_offset;   /* CIGAR ops live in ops[begin_offset,end_offset) */
    int32_t end_offset;
    int32_t score;
    int32_t status; /* AIM_PAIR_* */
    uint32_t idx;
} aim_result_t;

/* AIM_FLAG_RES8: the two numbers the reference prints per pair without BACKTRACE (host.c:339-341). */
typedef struct aim_result8 {
    uint32_t idx;
    int32_t score;
} aim_result8_t;

/* ---- library / device discovery ----------------------------------------- */
int aim_abi_version(void);
const char *aim_last_error(void);
/* Number of usable gfx950 devices (0 and AIM_ENODEV when there is none). */
int aim_device_count(int *count);

/* ---- device set: replaces struct dpu_set_t and the nine SDK calls -------- */
typedef struct aim_set aim_set_t;

/* dpu_alloc(NR_DPUS, NULL, &set) + dpu_load(set, DPU_BINARY, NULL)
 * (host.c:186-187).  device_ids may be NULL (= 0..nr_devices-1). */
int aim_set_alloc(uint32_t nr_devices, const int *device_ids, aim_set_t **set);
/* dpu_get_nr_dpus (host.c:188) */
int aim_set_nr_devices(const aim_set_t *set, uint32_t *nr_devices);
/* The compile-time configuration plus the MRAM plan of host.c:215-241
 * (mram_heap_alloc of params/requests/results/patterns/texts/operations):
 * sizes every device-side buffer for up to max_pairs_per_device pairs. */
int aim_set_configure(aim_set_t *set, const aim_params_t *params, uint32_t max_pairs_per_device);
/* The four host->device scatters of host.c:246-268 (DPUParams, requests,
 * patterns, texts) for ONE device of the set.  patterns/texts are
 * [n_pairs][read_size] byte rows.  Asynchronous when the host buffers come
 * from aim_host_alloc; the copy is ordered before the next launch. */
int aim_set_push(aim_set_t *set, uint32_t device, uint32_t n_pairs, const void *requests /* aim_request_t[] or, with
                 AIM_FLAG_REQ8, aim_request8_t[] */, const char *patterns, const char *texts);
/* dpu_launch(set, DPU_SYNCHRONOUS) (host.c:289): runs the alignment kernel on
 * every device of the set and waits for all of them. */
int aim_set_launch(aim_set_t *set);
/* The device->host gathers of host.c:316-326: results[n_pairs] and, with
 * AIM_FLAG_BACKTRACE, ops[n_pairs][2*read_size] (may be NULL otherwise).
 * CONTRACT OF AN OPS ROW: ops[i][begin_offset, end_offset) holds pair i's edit operations -- the bytes edit_cigar_print reads
 * (host.c:347-349). The REST of the row is unspecified: the reference's memset(operations, 'M', 2*READ_SIZE) (wfa.c:465,
 * swg.c:261) is only performed where an operation can be printed, so a caller that compares or copies whole rows sees
 * whatever its buffer held before (tests run with AIM_DEBUG_POISON_OPS to keep every kernel honest about that). */
int aim_set_pull(aim_set_t *set, uint32_t device, void *results /* aim_result_t[] or, with AIM_FLAG_RES8,
                 aim_result8_t[] */, char *ops);
/* The three phase timers host.c prints ("CPU-DPU", "DPU Kernel", "DPU-CPU",
 * host.c:270-272, 297-299, 328-330), in milliseconds, accumulated. Devices of a set work concurrently: each figure is the
 * SLOWEST device's (aim_set_launch: per launch; aim_set_submit / aim_set_wait: each device's batches summed, then the
 * maximum over devices). With several slots the phases of one device's batches overlap each other, so the three figures are
 * device-time per phase, not a partition of the wall time. */
int aim_set_timers(const aim_set_t *set, float *h2d_ms, float *kernel_ms, float *d2h_ms);
/* How many pairs of the last launch on `device` left the short-read fast path
 * (sequences with bytes other than A/C/G/T) and were aligned by the general
 * kernel instead.  Diagnostic only; 0 when the configuration has no fast path. */
int aim_set_fallback_pairs(aim_set_t *set, uint32_t device, uint32_t *n_fallback);
/* One line naming the plan the last launch on `device` followed (before the first launch: the configure-time plan):
 * kernel, lanes / wavefronts per pair, grid, block, LDS, scratch, scratch bound.  The AIM_* environment switches
 * (experiment / debugging knobs; none changes results) are read once per aim_set_configure and frozen in the set, so
 * this line cannot change between a configure and its launches. */
int aim_set_plan_describe(const aim_set_t *set, uint32_t device, char *out, size_t cap);
/* dpu_free (host.c:371) */
int aim_set_free(aim_set_t *set);

/* ---- pipelined batches: packed input, compact CIGAR output, double buffering -----------------
 * The reference's host loop is strictly serial (host.c:246-330: scatter, launch, gather, print).  These entry points
 * keep its data (same pairs, same results) and overlap its phases: a set configured with aim_set_configure_slots owns
 * `slots` independent buffer sets and streams per device, aim_set_submit enqueues H2D + kernel(s) + D2H of one batch on a
 * slot and returns, aim_set_wait blocks until that slot's results are in the caller's buffers.  With two slots
 * pack(k+1) || H2D(k+1) || kernel(k) || D2H(k-1) || format(k-1).  aim_set_push / launch / pull keep working (slot 0).
 *
 * Packed input (opt-in): 2 bits per base, code = (ascii >> 1) & 3 (A 0, C 1, T 2, G 3); base i of a sequence sits at bits
 * [2*(i%16), 2*(i%16)+1] of dword i/16 of its row; a row is ceil(read_size/16) dwords.  The reference compares raw bytes
 * and accepts any character (host.c:126-127), so a pair with a byte outside A/C/G/T inside either sequence cannot be
 * packed: it is listed in raw_pairs[] (batch indices, ascending) and its two ASCII rows travel in raw_patterns /
 * raw_texts ([n_raw][read_size]); its packed rows are ignored.  The device expands the batch into the reference's own
 * char[n][READ_SIZE] layout before any alignment kernel runs, so results are bit-identical to the ASCII path.
 *
 * Compact CIGAR (opt-in, needs AIM_FLAG_BACKTRACE): instead of ops[n][2*read_size], the device run-length encodes
 * ops[begin_offset, end_offset) -- the loop of edit_cigar_print, host.c:69-89 -- and returns one aim_cigar_t per pair plus
 * a shared run buffer; run = (length << 8) | op character.  aim_cigar_format_runs prints it exactly like the reference. */
#define AIM_CIGAR_OVERFLOW 0x100u /* aim_cigar_t.status bit: the run buffer was too small for this pair's runs */
typedef struct aim_cigar {
    uint32_t idx;
    int32_t score;
    uint32_t run_offset; /* first run of this pair in the run buffer */
    uint16_t n_runs;     /* 0 when status != AIM_PAIR_OK */
    uint16_t status;     /* AIM_PAIR_* | AIM_CIGAR_OVERFLOW */
} aim_cigar_t;

typedef struct aim_batch_io {
    uint32_t n_pairs;
    const void *requests;            /* aim_request_t[n] or aim_request8_t[n] (AIM_FLAG_REQ8) */
    const char *patterns, *texts;    /* ASCII rows [n][read_size], or NULL when the batch is packed */
    const uint32_t *packed_patterns; /* packed rows [n][ceil(read_size/16)] dwords, or NULL */
    const uint32_t *packed_texts;
    uint32_t n_raw;                  /* packed batches: pairs that travel as raw rows */
    const uint32_t *raw_pairs;       /* [n_raw] batch indices */
    const char *raw_patterns, *raw_texts; /* [n_raw][read_size] */
    void *results;                   /* out: aim_result_t[n] / aim_result8_t[n]; may be NULL when cigars is given */
    char *ops;                       /* out: ops[n][2*read_size] (AIM_FLAG_BACKTRACE), or NULL */
    aim_cigar_t *cigars;             /* out: compact CIGAR headers [n], or NULL */
    uint32_t *runs;                  /* out: run buffer */
    uint32_t runs_cap;               /* capacity of runs[], in runs */
} aim_batch_io_t;

/* aim_set_configure with `slots` (1..4) buffer sets per device; max_raw_pairs bounds n_raw of a packed batch
 * (0 = packed input not used), max_runs the run buffer of a compact-CIGAR batch (0 = not used). */
int aim_set_configure_slots(aim_set_t *set, const aim_params_t *params, uint32_t max_pairs_per_device, uint32_t slots,
                            uint32_t max_raw_pairs, uint32_t max_runs);
/* Enqueue one batch on (device, slot): H2D, [unpack], alignment kernel(s), [CIGAR run-length encoding], D2H into the
 * buffers named in *io (which must stay valid, and should be pinned -- aim_host_alloc -- for the copies to overlap).
 * Returns immediately.  A slot holds one batch at a time: aim_set_wait it before submitting to it again. */
int aim_set_submit(aim_set_t *set, uint32_t device, uint32_t slot, const aim_batch_io_t *io);
/* Block until the batch on (device, slot) is complete.  n_runs (may be NULL) receives the number of runs written.
 * Returns AIM_EALIGN like aim_set_pull when a pair stopped with a status other than AIM_PAIR_OK. */
int aim_set_wait(aim_set_t *set, uint32_t device, uint32_t slot, uint32_t *n_runs);
/* Host-side packer used by the CLI and the tests: packs one sequence (len bytes) into row[ceil(read_size/16)]; returns 1
 * when every byte is A/C/G/T, 0 when the pair must travel raw (the row content is then unspecified). */
int aim_pack_sequence(const char *seq, int32_t len, int32_t read_size, uint32_t *row);
/* Whole-batch packer (host threads): ASCII rows -> packed rows + raw side list, exactly what aim_batch_io_t takes.
 * *n_raw receives the number of pairs that must travel raw; AIM_ENOMEM when it exceeds max_raw (ship the batch as ASCII). */
int aim_pack_batch(const aim_params_t *params, uint32_t n_pairs, const void *requests, const char *patterns, const char *texts,
                   uint32_t *packed_patterns, uint32_t *packed_texts, uint32_t *raw_pairs, char *raw_patterns,
                   char *raw_texts, uint32_t max_raw, uint32_t *n_raw, int threads);
/* edit_cigar_print (host.c:69-89) from runs: adjacent runs of the same op are merged; returns bytes written incl. '\n'. */
int aim_cigar_format_runs(const uint32_t *runs, uint32_t n_runs, char *out, int32_t cap);

/* Pinned host staging for aim_set_push / aim_set_pull / aim_set_submit. */
int aim_host_alloc(void **ptr, size_t bytes);
int aim_host_free(void *ptr);

/* ---- device-resident entry point ----------------------------------------
 * One alignment launch over buffers that already live in HBM (same layouts as
 * above).  Used by the benchmark and by callers that manage device memory
 * themselves.  hip_stream is a hipStream_t (NULL = default stream); the call
 * only enqueues work.  d_scratch must hold aim_scratch_bytes() bytes.  Scratch is need-capped; for table-heavy
 * configurations (full-DP CIGAR at long reads) it is bounded by AIM_SCRATCH_GB, default 3/4 of the device's free
 * memory read once per process, and a smaller bound only means fewer pairs in flight (more rounds), never an error
 * unless not even one workgroup's table fits (AIM_ENOMEM).
 * d_patterns / d_texts must be 16-byte aligned and carry >= 16 bytes of
 * addressable slack after the last row (the kernels read whole 16-byte chunks). */
size_t aim_scratch_bytes(const aim_params_t *params, uint32_t n_pairs);
int aim_align_device(const aim_params_t *params, uint32_t n_pairs, const void *d_requests,
                     const char *d_patterns, const char *d_texts, void *d_results,
                     char *d_ops, void *d_scratch, size_t scratch_bytes, void *hip_stream);
/* The plan aim_align_device would follow for (params, n_pairs) in this process right now, as one line (see
 * aim_set_plan_describe).  The stateless entry points read the AIM_* switches at every call. */
int aim_plan_describe(const aim_params_t *params, uint32_t n_pairs, char *out, size_t cap);
/* Name of the kernel aim_align_device would launch for this configuration
 * (matches the rocprofv3 kernel-trace name prefix). */
const char *aim_kernel_name(const aim_params_t *params);

/* ---- host-side helpers shared by the CLI and the Python binding ---------- */
/* MAX_SCORE / READ_SIZE heuristics of the launchers (run-wfa-pim-wram.py:57-68,
 * run-nw-pim-wram.py:50-57 [gap instead of gap_o+gap_e], run-swg-pim-wram.py:52-62). */
int aim_launcher_sizes(int32_t algo, int32_t read_length, double error, int32_t mismatch, int32_t gap_o,
                       int32_t gap_e, int32_t gap, int32_t *max_score, int32_t *read_size);
/* edit_cigar_print (host.c:69-89): RLE of ops[begin,end) + '\n' into out;
 * returns bytes written or AIM_EINVAL if cap is too small. */
int aim_cigar_format(const char *ops, int32_t begin_offset, int32_t end_offset, char *out, int32_t cap);
/* Seeded synthetic pairs (DESIGN.md "Synthetic data"): pattern = len uniform
 * ACGT bases; text = pattern after ceil(len*error) sequential uniform
 * substitute/insert/delete edits.  Pair i depends only on (seed, first_idx+i). */
int aim_gen_pairs(uint64_t seed, uint64_t first_idx, uint32_t n_pairs, int32_t len, double error,
                  int32_t read_size, aim_request_t *requests, char *patterns, char *texts);

#ifdef __cplusplus
}
#endif
#endif /* AIM_HIP_H */
