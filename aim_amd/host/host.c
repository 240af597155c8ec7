/*
 * host.c -- the AIM host program over the MI355X C-ABI (include/aim_hip.h).
 *
 * Keeps the reference CLI and I/O contract of the six host/host.c programs of safaad/aim:
 *     host <input> <output> <total_nb_reads>
 * same stdout progress lines, same output file format ("idx, score, \n" and,
 * with backtrace, one RLE CIGAR line per pair), same exit codes
 * (WFA/DPU-WRAM/host/host.c:136-376).  What the reference fixes at compile
 * time through -D macros is passed at run time as optional flags after the
 * three positional arguments (aim_amd/launch.py emits them):
 *     --algo nw|swg|wfa  --max-score S  --read-size R  --match M --mismatch X
 *     --gap-o G --gap-e A --gap G [--gap-i GI --gap-d GD]  --backtrace  --reduce  --swg-w16
 *     --nr-dpus D   logical partition count of the reference (host.c:191: the
 *                   file is consumed in D blocks of ROUND_UP_8(n/D) pairs; n is
 *                   not a cap) -- kept so the set of aligned pairs is identical
 *     --gpus N      physical MI355X devices to shard each batch over
 *     --batch B     pairs per device per launch (default 4194304)
 *     --threads T   host threads for parsing / formatting (default: all cores, max 512; two thirds parse + pack the next
 *                   batch while one third formats the previous one and one more thread writes the one before)
 *     --packed-input  <input> is a packed batch file (written by --pack-only or `python -m aim_amd.gen_dataset --packed`):
 *                   2 bits per base + raw side list, ready for the device; no text is parsed
 * The UPMEM dispatch (dpu_alloc/dpu_load/dpu_push_xfer/dpu_launch) is replaced
 * by aim_set_* calls; there is no CPU alignment path.
 *
 * The two host hot loops of the reference -- get_reads (host.c:91-134: getline
 * x2 + strcpy per pair) and the output loop (host.c:331-352: fprintf per pair)
 * -- are kept in meaning but run in parallel: the input is mapped, newlines
 * are indexed by all threads, every thread packs a contiguous range of pairs
 * straight into the pinned batch, and results are formatted into per-thread
 * buffers that are written in order.
 */
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/resource.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include "aim_hip.h"

#define ROUND_UP_MULTIPLE_8(x) ((((x) + 7) / 8) * 8)
#define MAX_THREADS 512

static double now_ms(void)
{
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return tv.tv_sec * 1e3 + tv.tv_usec * 1e-3;
}

static void die_aim(const char *what, int rc)
{
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, aim_last_error());
    exit(1);
}

/* ---- persistent worker pools ------------------------------------------------------------------------
 * pool_start hands fn(tid, nthreads, arg) to the pool's workers and returns; pool_join waits for all of them. Two pools
 * exist so that the parse + pack of batch k+1 and the format + write of batch k-1 run at the same time (the reference's
 * loop is strictly serial, host.c:246-352). A worker that could not be created has its share run inside pool_join. */
typedef void (*range_fn)(int tid, int nthreads, void *arg);
struct pool;
typedef struct { struct pool *p; int tid; } worker_t;
typedef struct pool {
    pthread_t th[MAX_THREADS];
    int created[MAX_THREADS];
    worker_t workers[MAX_THREADS];
    int n, pending, stop;
    unsigned long gen;
    range_fn fn;
    void *arg;
    const cpu_set_t *cpus;   /* workers run on these CPUs (a lane's share of the machine), or NULL */
    pthread_mutex_t mu;
    pthread_cond_t go, idle;
} pool_t;

static void *pool_worker(void *a)
{
    worker_t *w = a;
    pool_t *p = w->p;
    unsigned long seen = 0;
    if (p->cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof(cpu_set_t), p->cpus);
    for (;;) {
        pthread_mutex_lock(&p->mu);
        while (p->gen == seen && !p->stop) pthread_cond_wait(&p->go, &p->mu);
        if (p->stop) { pthread_mutex_unlock(&p->mu); return NULL; }
        seen = p->gen;
        range_fn fn = p->fn;
        void *arg = p->arg;
        const int n = p->n;
        pthread_mutex_unlock(&p->mu);
        fn(w->tid, n, arg);
        pthread_mutex_lock(&p->mu);
        if (--p->pending == 0) pthread_cond_signal(&p->idle);
        pthread_mutex_unlock(&p->mu);
    }
}
static void pool_init(pool_t *p, int n, const cpu_set_t *cpus)
{
    memset(p, 0, sizeof *p);
    if (n < 1) n = 1;
    if (n > MAX_THREADS) n = MAX_THREADS;
    p->n = n;
    p->cpus = cpus;
    pthread_mutex_init(&p->mu, NULL);
    pthread_cond_init(&p->go, NULL);
    pthread_cond_init(&p->idle, NULL);
    for (int t = 0; t < n; ++t) {
        p->workers[t] = (worker_t){p, t};
        p->created[t] = pthread_create(&p->th[t], NULL, pool_worker, &p->workers[t]) == 0;
    }
}
static void pool_start(pool_t *p, range_fn fn, void *arg)
{
    pthread_mutex_lock(&p->mu);
    p->fn = fn; p->arg = arg;
    p->pending = 0;
    for (int t = 0; t < p->n; ++t) p->pending += p->created[t];
    ++p->gen;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
}
static void pool_join(pool_t *p)
{
    for (int t = 0; t < p->n; ++t)
        if (!p->created[t]) p->fn(t, p->n, p->arg);   /* no thread to be had (resource limit): this share runs here */
    pthread_mutex_lock(&p->mu);
    while (p->pending) pthread_cond_wait(&p->idle, &p->mu);
    pthread_mutex_unlock(&p->mu);
}
static void pool_run(pool_t *p, range_fn fn, void *arg) { pool_start(p, fn, arg); pool_join(p); }
static void pool_stop(pool_t *p)
{
    pthread_mutex_lock(&p->mu);
    p->stop = 1;
    pthread_cond_broadcast(&p->go);
    pthread_mutex_unlock(&p->mu);
    for (int t = 0; t < p->n; ++t)
        if (p->created[t]) pthread_join(p->th[t], NULL);
}

/* ---- input: mapped file + line index --------------------------------------------------------------- */
typedef struct {
    const char *data;
    size_t size;
    size_t *line_start;   /* [n_lines + 1]; line i = [line_start[i], line_start[i+1]) including its '\n' if any */
    size_t n_lines;
    size_t counts[MAX_THREADS + 1];
    size_t *found[MAX_THREADS];   /* per-thread newline positions (+1) of the one text pass */
    int populate;
} input_t;

/* ONE pass over the text: every thread collects the line starts of its byte range, then they are concatenated. (Round 2
 * counted and filled in two passes: at tens of gigabytes of input the second read of the text was the price.) */
static void find_newlines(int tid, int nt, void *arg)
{
    input_t *in = arg;
    size_t lo = in->size * tid / nt, hi = in->size * (tid + 1) / nt, c = 0, cap = (hi - lo) / 48 + 64;
    size_t *v = malloc(cap * sizeof *v);
#ifdef MADV_POPULATE_READ
    if (in->populate && hi > lo) {   /* fault this thread's share of the mapping in one call instead of page by page */
        const size_t pg = (size_t)sysconf(_SC_PAGESIZE), a0 = lo / pg * pg;
        (void)madvise((void *)(in->data + a0), hi - a0, MADV_POPULATE_READ);
    }
#endif
    const char *p = in->data + lo, *e = in->data + hi;
    while (v && p < e && (p = memchr(p, '\n', (size_t)(e - p)))) {
        if (c == cap) { cap *= 2; size_t *nv = realloc(v, cap * sizeof *v); if (!nv) { free(v); v = NULL; break; } v = nv; }
        v[c++] = (size_t)(p - in->data) + 1;   /* each '\n' at position x starts a line at x+1 */
        ++p;
    }
    in->found[tid] = v;
    in->counts[tid + 1] = v ? c : (size_t)-1;
}
static void place_newlines(int tid, int nt, void *arg)
{
    input_t *in = arg;
    (void)nt;
    const size_t n = in->counts[tid + 1] - in->counts[tid];
    if (n) memcpy(in->line_start + 1 + in->counts[tid], in->found[tid], n * sizeof(size_t));   /* line_start[0] = 0 */
    free(in->found[tid]);
}
static void index_lines(input_t *in, pool_t *pool)
{
    const int nt = pool->n;
    in->counts[0] = 0;
    pool_run(pool, find_newlines, in);
    for (int t = 0; t < nt; ++t) {
        if (in->counts[t + 1] == (size_t)-1) { fprintf(stderr, "out of host memory\n"); exit(1); }
        in->counts[t + 1] += in->counts[t];
    }
    size_t n_nl = in->counts[nt];
    in->line_start = malloc((n_nl + 2) * sizeof(size_t));
    if (!in->line_start) { fprintf(stderr, "out of host memory\n"); exit(1); }
    in->line_start[0] = 0;
    pool_run(pool, place_newlines, in);
    /* a final line without '\n' still is a line for getline() */
    in->n_lines = n_nl;
    if (in->size > 0 && in->data[in->size - 1] != '\n') in->line_start[++in->n_lines] = in->size;
}

/* ---- get_reads (host.c:91-134) over a contiguous range of pairs, in parallel ------------------------ */
/* One job = one batch on one (device, slot): everything the parser threads fill and the device returns. */
typedef struct {
    uint32_t n;                      /* pairs in this job */
    size_t first_pair;               /* global index of pair 0 */
    void *req;                       /* aim_request8_t[] (the reference's own 8-byte WFA request_t, AIM_FLAG_REQ8) when lengths
                                        fit int16, else aim_request_t[] (long reads) */
    int req8;
    uint32_t *pkP, *pkT;             /* packed rows (2 bits per base) */
    uint32_t *raw_idx; char *rawP, *rawT; uint32_t n_raw;   /* side list: pairs with a byte outside A/C/G/T */
    char *pat, *txt;                 /* ASCII rows: --no-pack, or a batch whose side list overflowed (allocated lazily) */
    int ascii;                       /* this job travels as ASCII rows */
    uint8_t *is_raw;                 /* [n] pass-0 verdict per pair (plain malloc) */
    aim_result8_t *res8;             /* score-only results */
    aim_cigar_t *cig; uint32_t *runs; uint32_t n_runs;      /* compact CIGAR */
    aim_result_t *res; char *ops;    /* --full-ops: the reference's own result_t + ops rows */
    int use_full;                    /* this job's output came back as result_t + ops rows (--full-ops, or the run buffer overflowed) */
    aim_batch_io_t io;               /* what was submitted (kept for a re-submission) */
    uint32_t device, slot;
    int in_flight;
    int dry_filled;                  /* AIM_HOST_DRY: the stand-in results were written into this job's buffers */
} job_t2;

typedef struct {
    const input_t *in;
    job_t2 *job;
    int read_size;
    uint32_t max_raw;
    uint32_t raw_count[MAX_THREADS + 1];   /* per-thread raw pairs (pass 1), then exclusive offsets */
    int pass;
} pack_t;

#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("sse4.1,ssse3,bmi2"))) static int pack_seq_simd(const char *seq, long len, uint32_t *row, uint32_t row_dw)
{
    /* 16 bases per step: code = (c >> 1) & 3, validated by decoding back through "ACTG" (pshufb) */
    const __m128i lut = _mm_setr_epi8('A', 'C', 'T', 'G', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0);
    const __m128i three = _mm_set1_epi8(3);
    __m128i bad = _mm_setzero_si128();
    long i = 0;
    uint32_t w = 0;
    for (; i + 16 <= len; i += 16, ++w) {
        const __m128i x = _mm_loadu_si128((const __m128i *)(seq + i));
        const __m128i c = _mm_and_si128(_mm_srli_epi16(x, 1), three);
        bad = _mm_or_si128(bad, _mm_xor_si128(x, _mm_shuffle_epi8(lut, c)));
        const uint64_t lo = (uint64_t)_mm_cvtsi128_si64(c), hi = (uint64_t)_mm_extract_epi64(c, 1);
        row[w] = (uint32_t)_pext_u64(lo, 0x0303030303030303ull) | ((uint32_t)_pext_u64(hi, 0x0303030303030303ull) << 16);
    }
    int ok = _mm_testz_si128(bad, bad);
    if (i < len) {
        uint32_t v = 0;
        for (int sh = 0; i < len; ++i, sh += 2) {
            const unsigned char ch = (unsigned char)seq[i];
            const uint32_t code = (ch >> 1) & 3u;
            ok &= ch == (unsigned char)"ACTG"[code];
            v |= code << sh;
        }
        row[w++] = v;
    }
    for (; w < row_dw; ++w) row[w] = 0;
    return ok;
}
#endif

static int g_simd = 0;
static inline int pack_seq(const char *seq, long len, int read_size, uint32_t *row, uint32_t row_dw)
{
#if defined(__x86_64__)
    if (g_simd) return pack_seq_simd(seq, len, row, row_dw);
#endif
    return aim_pack_sequence(seq, (int32_t)len, read_size, row) == 1;
}

/* H1: getline length includes the '\n'; the first character and the last one are dropped. */
static inline void pair_lines(const input_t *in, size_t pair, const char **p, long *pl, const char **t, long *tl)
{
    const size_t *ls = in->line_start + 2 * pair;
    *pl = (long)(ls[1] - ls[0]) - 2;
    *tl = (long)(ls[2] - ls[1]) - 2;
    *p = in->data + ls[0] + 1;
    *t = in->data + ls[1] + 1;
}

static void pack_range(int tid, int nt, void *arg)
{
    pack_t *pk = arg;
    job_t2 *j = pk->job;
    const size_t lo = (size_t)j->n * tid / nt, hi = (size_t)j->n * (tid + 1) / nt;
    const int rs = pk->read_size;
    const uint32_t dw = (uint32_t)(rs + 15) / 16u;
    if (pk->pass == 0) {   /* requests + packed rows; note and count the pairs that cannot be packed */
        uint32_t raw = 0;
        for (size_t i = lo; i < hi; ++i) {
            const char *p, *t; long pl, tl;
            pair_lines(pk->in, j->first_pair + i, &p, &pl, &t, &tl);
            if (j->req8) {
                aim_request8_t *r = (aim_request8_t *)j->req + i;
                r->pattern_len = (int16_t)pl; r->text_len = (int16_t)tl; r->idx = (uint32_t)(j->first_pair + i);
            } else {
                aim_request_t *r = (aim_request_t *)j->req + i;
                r->pattern_len = (int32_t)pl; r->text_len = (int32_t)tl; r->padding = 0; r->idx = (uint32_t)(j->first_pair + i);
            }
            if (j->ascii) {
                char *dp = j->pat + i * rs, *dt = j->txt + i * rs;
                memcpy(dp, p, (size_t)pl); memset(dp + pl, 0, (size_t)(rs - pl));
                memcpy(dt, t, (size_t)tl); memset(dt + tl, 0, (size_t)(rs - tl));
            } else {
                const int okp = pack_seq(p, pl, rs, j->pkP + i * dw, dw), okt = pack_seq(t, tl, rs, j->pkT + i * dw, dw);
                j->is_raw[i] = !(okp && okt);
                raw += j->is_raw[i];
            }
        }
        pk->raw_count[tid + 1] = raw;
    } else {               /* side list in ascending pair order: thread t owns slots [raw_count[t], raw_count[t+1]) */
        uint32_t at = pk->raw_count[tid];
        for (size_t i = lo; i < hi; ++i) {
            if (!j->is_raw[i]) continue;
            const char *p, *t; long pl, tl;
            pair_lines(pk->in, j->first_pair + i, &p, &pl, &t, &tl);
            char *dp = j->rawP + (size_t)at * rs, *dt = j->rawT + (size_t)at * rs;
            memcpy(dp, p, (size_t)pl); memset(dp + pl, 0, (size_t)(rs - pl));
            memcpy(dt, t, (size_t)tl); memset(dt + tl, 0, (size_t)(rs - tl));
            j->raw_idx[at++] = (uint32_t)i;
        }
    }
}

/* Fill one job from the mapped input: requests, packed rows + raw side list (or ASCII rows). pack_begin starts the big pass
 * on the pack pool and returns (the caller formats the previous batch meanwhile); pack_finish joins it and assembles the raw
 * side list (or falls back to ASCII rows for an unusually dirty batch). */
static void *(*g_big_alloc)(size_t);
static void pack_begin(pool_t *pool, pack_t *pk, const input_t *inp, job_t2 *j, int read_size, uint32_t max_raw, int no_pack)
{
    memset(pk, 0, sizeof *pk);
    pk->in = inp; pk->job = j; pk->read_size = read_size; pk->max_raw = max_raw; pk->pass = 0;
    j->ascii = no_pack;
    j->n_raw = 0;
    pool_start(pool, pack_range, pk);
}
static void pack_finish(pool_t *pool, pack_t *pk, job_t2 *j, uint32_t batch)
{
    const int threads = pool->n;
    const size_t rs = (size_t)pk->read_size;
    pool_join(pool);
    if (j->ascii) return;
    uint32_t total_raw = 0;
    for (int t = 0; t < threads; ++t) { const uint32_t c = pk->raw_count[t + 1]; pk->raw_count[t] = total_raw; total_raw += c; }
    pk->raw_count[threads] = total_raw;
    if (!total_raw) return;
    if (total_raw > pk->max_raw) {   /* unusually dirty batch: ship it as ASCII rows instead (allocated on first need) */
        if (!j->pat) { j->pat = g_big_alloc((size_t)batch * rs); j->txt = g_big_alloc((size_t)batch * rs); }
        j->ascii = 1;
        pk->pass = 0;
        pool_run(pool, pack_range, pk);
        return;
    }
    pk->pass = 1;
    pool_run(pool, pack_range, pk);
    j->n_raw = total_raw;
}

/* ---- packed batch files (--pack-only writes them, --packed-input reads them, gen_dataset --packed writes them too) ------
 * 64-byte file header, then per batch {n, ascii, n_raw, read_size} + the arrays aim_batch_io_t takes, in that order:
 * requests, then ASCII rows (patterns, texts) or packed rows (patterns, texts) + raw side list (indices, patterns, texts). */
typedef struct {
    char magic[8];          /* "AIMPK\0\0\1" */
    uint32_t version;       /* 1 */
    uint32_t read_size;
    uint32_t req_bytes;     /* 8 (aim_request8_t) or 16 (aim_request_t) */
    uint32_t batch_pairs;   /* no batch holds more pairs than this */
    uint64_t total_pairs;
    uint8_t pad[32];
} pkfile_hdr_t;
static const char PKFILE_MAGIC[8] = {'A', 'I', 'M', 'P', 'K', 0, 0, 1};

typedef struct { const char *src; char *dst; size_t bytes; } copy_item_t;
typedef struct { copy_item_t it[8]; int n; } copy_set_t;
static void copy_range(int tid, int nt, void *arg)
{
    const copy_set_t *cs = arg;
    for (int i = 0; i < cs->n; ++i) {
        const size_t lo = cs->it[i].bytes * tid / nt, hi = cs->it[i].bytes * (tid + 1) / nt;
        if (hi > lo) memcpy(cs->it[i].dst + lo, cs->it[i].src + lo, hi - lo);
    }
}
/* Size of the batch that starts at `at` (0: malformed) -- the walk that finds a packed file's batch boundaries. */
static size_t packed_batch_bytes(const char *at, size_t left, int req8, int read_size, uint32_t *n_out)
{
    if (left < 16) return 0;
    uint32_t hdr[4];
    memcpy(hdr, at, 16);
    const uint32_t n = hdr[0], ascii = hdr[1], n_raw = hdr[2];
    const size_t rs = (size_t)read_size, dw = (size_t)(read_size + 15) / 16, rq = req8 ? sizeof(aim_request8_t) : sizeof(aim_request_t);
    if (hdr[3] != (uint32_t)read_size || n_raw > n) return 0;
    const size_t need = 16 + n * rq + (ascii ? 2 * n * rs : 2 * n * dw * 4 + (size_t)n_raw * (4 + 2 * rs));
    if (left < need) return 0;
    *n_out = n;
    return need;
}
/* One batch of a packed file into the job's (pinned) buffers; returns the bytes consumed, 0 on a malformed batch. `take` <= n
 * pairs of it are used (the reference's partition rule may end inside a batch). */
static size_t packed_begin(pool_t *pool, copy_set_t *cs, const char *at, size_t left, job_t2 *j, int read_size, uint32_t batch, uint32_t max_raw, uint32_t take)
{
    if (left < 16) return 0;
    uint32_t hdr[4];
    memcpy(hdr, at, 16);
    const uint32_t n = hdr[0], ascii = hdr[1], n_raw = hdr[2];
    const size_t rs = (size_t)read_size, dw = (size_t)(read_size + 15) / 16, rq = j->req8 ? sizeof(aim_request8_t) : sizeof(aim_request_t);
    if (hdr[3] != (uint32_t)read_size || n > batch || n_raw > n || (!ascii && n_raw > max_raw) || take > n) return 0;
    const size_t need = 16 + n * rq + (ascii ? 2 * n * rs : 2 * n * dw * 4 + (size_t)n_raw * (4 + 2 * rs));
    if (left < need) return 0;
    const char *q = at + 16;
    cs->n = 0;
    cs->it[cs->n++] = (copy_item_t){q, (char *)j->req, take * rq}; q += n * rq;
    j->ascii = (int)ascii;
    j->n_raw = 0;
    if (ascii) {
        if (!j->pat) { j->pat = g_big_alloc((size_t)batch * rs); j->txt = g_big_alloc((size_t)batch * rs); }
        cs->it[cs->n++] = (copy_item_t){q, j->pat, take * rs}; q += n * rs;
        cs->it[cs->n++] = (copy_item_t){q, j->txt, take * rs};
    } else {
        cs->it[cs->n++] = (copy_item_t){q, (char *)j->pkP, take * dw * 4}; q += n * dw * 4;
        cs->it[cs->n++] = (copy_item_t){q, (char *)j->pkT, take * dw * 4}; q += n * dw * 4;
        uint32_t keep = n_raw;                       /* side-list entries beyond `take` (ascending indices) are dropped */
        if (take < n) { keep = 0; while (keep < n_raw) { uint32_t v; memcpy(&v, q + 4 * (size_t)keep, 4); if (v >= take) break; ++keep; } }
        cs->it[cs->n++] = (copy_item_t){q, (char *)j->raw_idx, (size_t)keep * 4}; q += (size_t)n_raw * 4;
        cs->it[cs->n++] = (copy_item_t){q, j->rawP, (size_t)keep * rs}; q += (size_t)n_raw * rs;
        cs->it[cs->n++] = (copy_item_t){q, j->rawT, (size_t)keep * rs};
        j->n_raw = keep;
    }
    pool_start(pool, copy_range, cs);
    return need;
}

/* Whole-input validation BEFORE anything is launched or written, like get_reads (host.c:119-123 runs inside the read loop,
 * ahead of every launch): over-length read -> message + exit(0) with an empty output file; short line -> error. */
typedef struct { const input_t *in; size_t n; int read_size; int too_long, malformed; } scan_t;
static void scan_range(int tid, int nt, void *arg)
{
    scan_t *sc = arg;
    const size_t lo = sc->n * tid / nt, hi = sc->n * (tid + 1) / nt;
    int too_long = 0, malformed = 0;
    for (size_t i = lo; i < hi; ++i) {
        const size_t *ls = sc->in->line_start + 2 * i;
        const long pl = (long)(ls[1] - ls[0]) - 2, tl = (long)(ls[2] - ls[1]) - 2;
        if (pl > sc->read_size || tl > sc->read_size) too_long = 1;
        if (pl < 0 || tl < 0) malformed = 1;
    }
    if (too_long) __atomic_store_n(&sc->too_long, 1, __ATOMIC_RELAXED);
    if (malformed) __atomic_store_n(&sc->malformed, 1, __ATOMIC_RELAXED);
}

/* ---- output loop (host.c:331-352) ---------------------------------------------------------------- */
typedef struct {
    const job_t2 *job;                 /* result buffers of the batch being printed */
    uint32_t n;                        /* its pair count (the job's own n already belongs to the next batch) */
    int use_full;
    int backtrace, read_size, full_ops;
    int fd;
    int set;                           /* which of the two text buffer sets this batch is printed into (the other one is being written) */
    char *bufs[2][MAX_THREADS];        /* per-thread text, kept and grown across batches */
    size_t caps[2][MAX_THREADS], lens[2][MAX_THREADS], offs[2][MAX_THREADS];
    char **buf;                        /* = bufs[set] etc. */
    size_t *cap, *len, *off;
    int seq;                           /* output is no regular file: sequential write() by one thread, in thread order */
    char *map;                         /* MAP_SHARED window of the output file for this batch (NULL: pwrite) */
    size_t map_base;                   /* file offset of map[0] */
    int failed;
} fmt_t;

static inline char *put_int(char *o, int v)
{
    char tmp[16];
    int n = 0;
    unsigned u = v < 0 ? 0u - (unsigned)v : (unsigned)v;
    if (v < 0) *o++ = '-';
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *o++ = tmp[--n];
    return o;
}

static void format_range(int tid, int nt, void *arg)
{
    fmt_t *f = arg;
    const job_t2 *j = f->job;
    const size_t lo = (size_t)f->n * tid / nt, hi = (size_t)f->n * (tid + 1) / nt;
    const size_t rs = (size_t)f->read_size;
    /* worst case per pair: "idx, score, \n" (<= 26 bytes) + one "%d%c" per op (<= 2 bytes per op when every run is 1) + '\n' */
    size_t cap = (hi - lo) * (32 + (f->backtrace ? 4 * rs + 16 : 0)) + 64;
    if (cap > f->cap[tid]) { free(f->buf[tid]); f->buf[tid] = malloc(cap); f->cap[tid] = cap; }
    char *o = f->buf[tid], *start = o;
    if (!o) { fprintf(stderr, "out of host memory\n"); exit(1); }
    for (size_t i = lo; i < hi; ++i) {
        /* fprintf(out, "%d, %d, \n", idx, score) */
        const int full = f->full_ops || f->use_full;
        const uint32_t idx = !f->backtrace ? j->res8[i].idx : (full ? j->res[i].idx : j->cig[i].idx);
        const int score = !f->backtrace ? j->res8[i].score : (full ? j->res[i].score : j->cig[i].score);
        o = put_int(o, (int)idx); *o++ = ','; *o++ = ' ';
        o = put_int(o, score); *o++ = ','; *o++ = ' '; *o++ = '\n';
        if (!f->backtrace) continue;
        if (full) {                                                  /* edit_cigar_print, host.c:69-89 */
            const aim_result_t *r = &j->res[i];
            const char *ops = j->ops + i * 2 * rs;
            char last = ops[r->begin_offset];
            int run = 1;
            for (int k = r->begin_offset + 1; k < r->end_offset; ++k) {
                if (ops[k] == last) ++run;
                else { o = put_int(o, run); *o++ = last; last = ops[k]; run = 1; }
            }
            o = put_int(o, run); *o++ = last; *o++ = '\n';
        } else {                                                     /* the same loop over device-side runs */
            const uint32_t *r = j->runs + j->cig[i].run_offset;
            const uint32_t nr = j->cig[i].n_runs;
            uint32_t run = nr ? r[0] >> 8 : 1;
            char last = nr ? (char)(r[0] & 0xff) : 'M';
            for (uint32_t k = 1; k < nr; ++k) {
                const char op = (char)(r[k] & 0xff);
                if (op == last) run += r[k] >> 8;
                else { o = put_int(o, (int)run); *o++ = last; last = op; run = r[k] >> 8; }
            }
            o = put_int(o, (int)run); *o++ = last; *o++ = '\n';
        }
    }
    f->len[tid] = (size_t)(o - start);
}

/* The third stage of the host pipeline: ONE thread writes a printed batch to the output file while the format pool prints the
 * next one and the pack pool parses the one after. Writes to a single file serialise on its inode lock whatever the number of
 * callers (measured on the 256-thread host: 64 M pairs, 928 MB of output -- parallel pwrite from 4 .. 32 threads, a shared mapping
 * filled by all threads and one thread writing all land within 10 % of each other, ~4-5 GB/s, and more threads only disturb the
 * parser), so the write is taken OFF the critical path instead of spread over threads. */
typedef struct {
    pthread_t th;
    pthread_mutex_t mu;
    pthread_cond_t cv;
    int started, busy, stop, failed;
    int fd, n, seq;                      /* seq: the output is no regular file (pipe, FIFO, /dev/stdout piped): write() in order, no offsets */
    char **buf;
    size_t *len, *off;
} writer_t;
static void *writer_main(void *arg)
{
    writer_t *w = arg;
    for (;;) {
        pthread_mutex_lock(&w->mu);
        while (!w->busy && !w->stop) pthread_cond_wait(&w->cv, &w->mu);
        if (!w->busy && w->stop) { pthread_mutex_unlock(&w->mu); return NULL; }
        pthread_mutex_unlock(&w->mu);
        for (int t = 0; t < w->n && !w->failed; ++t) {
            const char *q = w->buf[t];
            size_t left = w->len[t], off = w->off[t];
            while (left) {
                const ssize_t k = w->seq ? write(w->fd, q, left) : pwrite(w->fd, q, left, (off_t)off);
                if (k <= 0) { w->failed = 1; break; }
                q += k; off += (size_t)k; left -= (size_t)k;
            }
        }
        pthread_mutex_lock(&w->mu);
        w->busy = 0;
        pthread_cond_broadcast(&w->cv);
        pthread_mutex_unlock(&w->mu);
    }
}
static void writer_wait_idle(writer_t *w)
{
    pthread_mutex_lock(&w->mu);
    while (w->busy) pthread_cond_wait(&w->cv, &w->mu);
    pthread_mutex_unlock(&w->mu);
}
static void writer_submit(writer_t *w, int n, char **buf, size_t *len, size_t *off)
{
    writer_wait_idle(w);
    pthread_mutex_lock(&w->mu);
    w->n = n; w->buf = buf; w->len = len; w->off = off;
    w->busy = 1;
    pthread_cond_broadcast(&w->cv);
    pthread_mutex_unlock(&w->mu);
}

/* (A/B alternatives, AIM_HOST_OUT=pwrite | mmap) every thread writes its own text at its own offset: the copy into the page cache runs on all of them (one fwrite of
 * 50-100 MB per batch was the longest serial piece of the loop) */
static void write_range(int tid, int nt, void *arg)
{
    fmt_t *f = arg;
    (void)nt;
    if (f->map) {   /* the batch's window of the output file is mapped: every thread copies its text into place */
        if (f->len[tid]) memcpy(f->map + (f->off[tid] - f->map_base), f->buf[tid], f->len[tid]);
        return;
    }
    const char *q = f->buf[tid];
    size_t left = f->len[tid], off = f->off[tid];
    while (left) {
        const ssize_t w = f->seq ? write(f->fd, q, left) : pwrite(f->fd, q, left, (off_t)off);
        if (w <= 0) { __atomic_store_n(&f->failed, 1, __ATOMIC_RELAXED); return; }
        q += w; off += (size_t)w; left -= (size_t)w;
    }
}

static void *plain(size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (!p) { fprintf(stderr, "out of host memory\n"); exit(1); }
    return p;
}

static void *pinned(size_t bytes)
{
    void *p = NULL;
    int rc = aim_host_alloc(&p, bytes ? bytes : 1);
    if (rc) die_aim("aim_host_alloc", rc);
    return p;
}

/* ---- lanes ------------------------------------------------------------------------------------------
 * A LANE is the whole host pipeline -- pack pool, the (device, slot) job ring on its own device set, format pool, one writer,
 * one output file -- over one contiguous range of the input's pairs. The default is one lane writing <output>: the reference's
 * contract. `--out-shards K` runs K lanes side by side, lane k writing <output>.00k; `cat <output>.0*` is byte-identical to the
 * single file. That is what feeds several GPUs: one output file takes ~4-5 GB/s whoever writes it (its inode lock), i.e. ~3e8 pairs/s
 * score-only, and one lane's stages are sized for one device. Lanes share nothing but the read-only mapped input and its line
 * index; devices are dealt to lanes round-robin (more lanes than devices: several lanes -- each with its own set -- per device). */
typedef struct {
    aim_params_t p;
    int backtrace, use_req8, no_pack, full_ops, packed_input;
    uint32_t batch, slots, max_raw, runs_cap;
    const input_t *inp;
    const char *in_name, *out_name;
} cfg_t;

typedef struct lane {
    int id;
    const cfg_t *c;
    /* its share of the input: text = pairs [first_pair, first_pair + n_pairs); packed file = whole batches from byte pk_at */
    uint64_t first_pair, n_pairs;
    size_t pk_at;
    uint64_t n_jobs;
    /* its devices */
    int dev_ids[64];
    uint32_t gpus;
    aim_set_t *set;
    /* its threads */
    int pack_threads, fmt_threads;
    cpu_set_t cpus;
    int have_cpus;
    pool_t pack_pool, fmt_pool;
    pack_t pk;
    copy_set_t cs;
    writer_t writer;
    fmt_t f;
    int out_fd;
    char out_path[4096];
    pthread_t th;
    /* what it did */
    uint64_t done, pairs_first_done;
    double t_first_done, t_end, t_loop;
    double parse_ms, write_ms, wait_ms, join_ms, submit_ms, wrwait_ms;
    float h2d, kern, d2h;
} lane_t;

static pthread_barrier_t g_lanes_ready;   /* every lane has its device set and pinned buffers: the loops start together */
static pthread_mutex_t g_say_mu = PTHREAD_MUTEX_INITIALIZER;
static int g_said_copy, g_said_retrieve;
static void say_once(int *flag, const char *text)   /* the reference's progress lines, once per run whatever the lane count */
{
    pthread_mutex_lock(&g_say_mu);
    if (!*flag) { fputs(text, stdout); *flag = 1; }
    pthread_mutex_unlock(&g_say_mu);
}

static void *lane_main(void *arg)
{
    lane_t *L = arg;
    const cfg_t *c = L->c;
    const aim_params_t *p = &c->p;
    const size_t rs = (size_t)p->read_size;
    const uint32_t dw = (uint32_t)(p->read_size + 15) / 16u;
    const uint32_t batch = c->batch, slots = c->slots, gpus = L->gpus;
    const int backtrace = c->backtrace, full_ops = c->full_ops, use_req8 = c->use_req8;
    int rc;
    if (L->have_cpus) (void)pthread_setaffinity_np(pthread_self(), sizeof L->cpus, &L->cpus);   /* pinned buffers are first touched here */
    pool_init(&L->pack_pool, L->pack_threads, L->have_cpus ? &L->cpus : NULL);
    pool_init(&L->fmt_pool, L->fmt_threads, L->have_cpus ? &L->cpus : NULL);
    if ((rc = aim_set_alloc(gpus, L->dev_ids, &L->set))) die_aim("aim_set_alloc", rc);
    rc = aim_set_configure_slots(L->set, p, batch, slots, c->no_pack ? 0 : c->max_raw, c->runs_cap);
    if (rc) {
        if (rc == AIM_EINVAL) { printf("%s\n", aim_last_error()); exit(1); }
        die_aim("aim_set_configure_slots", rc);
    }
    const uint32_t ring = gpus * slots;
    job_t2 *jobs = calloc(ring, sizeof *jobs);
    for (uint32_t k = 0; k < ring; ++k) {
        job_t2 *j = &jobs[k];
        j->device = k % gpus; j->slot = k / gpus;
        j->req8 = use_req8;
        j->req = pinned((size_t)batch * (use_req8 ? sizeof(aim_request8_t) : sizeof(aim_request_t)));
        j->is_raw = plain(batch);
        if (c->no_pack) { j->pat = pinned((size_t)batch * rs); j->txt = pinned((size_t)batch * rs); }
        else {
            j->pkP = pinned((size_t)batch * dw * 4); j->pkT = pinned((size_t)batch * dw * 4);
            j->raw_idx = pinned((size_t)c->max_raw * 4); j->rawP = pinned((size_t)c->max_raw * rs); j->rawT = pinned((size_t)c->max_raw * rs);
        }
        if (!backtrace) j->res8 = pinned((size_t)batch * sizeof(aim_result8_t));
        else if (full_ops) { j->res = pinned((size_t)batch * sizeof(aim_result_t)); j->ops = pinned((size_t)batch * 2 * rs); }
        else { j->cig = pinned((size_t)batch * sizeof(aim_cigar_t)); j->runs = pinned((size_t)c->runs_cap * 4); }
    }

    const uint64_t n_jobs = L->n_jobs;
    uint64_t sent = 0;
    size_t out_at = 0, pk_at = L->pk_at;
    fmt_t *f = &L->f;   /* host.c:331-352; per-thread text buffers live across batches */
    memset(f, 0, sizeof *f);
    f->backtrace = backtrace; f->read_size = p->read_size; f->full_ops = full_ops; f->fd = L->out_fd;
    f->buf = f->bufs[0]; f->cap = f->caps[0]; f->len = f->lens[0]; f->off = f->offs[0];
    writer_t *W = &L->writer;
    memset(W, 0, sizeof *W);
    pthread_mutex_init(&W->mu, NULL);
    pthread_cond_init(&W->cv, NULL);
    W->fd = L->out_fd;
    const char *out_mode = getenv("AIM_HOST_OUT");
    /* A pipe / FIFO / character device cannot be written at offsets (pwrite: ESPIPE): like the reference's fopen(out, "w") the
       text then goes out sequentially -- one writer, batches and per-thread buffers in order. */
    int out_seq = 0;
    {
        struct stat os;
        if (fstat(L->out_fd, &os) || !S_ISREG(os.st_mode)) out_seq = 1;
    }
    W->seq = f->seq = out_seq;
    const int out_async = out_seq || !out_mode || !strcmp(out_mode, "async");
    if (out_async) W->started = pthread_create(&W->th, NULL, writer_main, W) == 0;
    int out_mmap = !out_seq && out_mode && !strcmp(out_mode, "mmap");   /* A/B switch; see the loop */
    const int out_serial = out_seq || (out_mode && !strcmp(out_mode, "serial"));   /* one thread writes (also the fallback when no writer thread could be created) */
    if (out_serial) out_mmap = 0;
    pthread_barrier_wait(&g_lanes_ready);
    /* measurement aids, not product paths: AIM_HOST_TRACE=1 prints every iteration's stage times to stderr; AIM_HOST_DRY=1 skips the
       device (no submit / wait: the result buffers keep {idx, 0}) -- the HOST-side rate of parse + pack + format + write alone */
    const int trace = getenv("AIM_HOST_TRACE") != NULL, dry = getenv("AIM_HOST_DRY") != NULL;
    L->t_loop = now_ms();
    for (uint64_t it = 0; it < n_jobs + ring; ++it) {
        job_t2 *j = &jobs[it % ring];
        const int have_old = j->in_flight;
        const double t_it = now_ms();
        double d_wait = 0, d_fmt = 0, d_wr = 0, d_join = 0, d_sub = 0;
        if (have_old) {   /* job it - ring: results are needed now (and its buffers next) */
            double t0 = now_ms();
            say_once(&g_said_retrieve, "Retrieve results\n");
            rc = dry ? 0 : aim_set_wait(L->set, j->device, j->slot, &j->n_runs);
            if (rc == AIM_ENOMEM && j->cig && !j->use_full) {
                /* more runs than READ_SIZE/4 + 2 per pair on average (e.g. SWG with MAX_SCORE as +infinity on dissimilar reads):
                   run this batch again and gather result_t + ops rows like the reference (host.c:316-326); the inputs are
                   still in the job's buffers */
                if (!j->res) { j->res = pinned((size_t)batch * sizeof(aim_result_t)); j->ops = pinned((size_t)batch * 2 * rs); }
                j->io.cigars = NULL; j->io.runs = NULL; j->io.runs_cap = 0;
                j->io.results = j->res; j->io.ops = j->ops;
                j->use_full = 1;
                if ((rc = aim_set_submit(L->set, j->device, j->slot, &j->io))) die_aim("aim_set_submit", rc);
                rc = aim_set_wait(L->set, j->device, j->slot, NULL);
            }
            if (rc == AIM_EALIGN) { /* the reference prints from the DPU and exits 1 */
                const char *msg = strstr(aim_last_error(), "(");
                pthread_mutex_lock(&g_say_mu);   /* (held: one lane prints, the process ends) */
                printf("%s\n", msg ? msg + 1 : aim_last_error());
                exit(1);
            }
            if (rc) die_aim("aim_set_wait", rc);
            L->wait_ms += (d_wait = now_ms() - t0);
            f->job = j; f->n = j->n; f->use_full = j->use_full;
            j->in_flight = 0;
        }
        /* the device is done with this job's input buffers: the next batch's parse + pack starts on the pack pool ... */
        const int have_new = it < n_jobs;
        double t_pack = now_ms();
        if (have_new) {
            const uint64_t left = L->n_pairs - sent;
            j->n = left < batch ? (uint32_t)left : batch;
            j->first_pair = (size_t)(L->first_pair + sent);
            if (c->packed_input) {
                uint32_t in_file = 0;
                if (c->inp->size - pk_at >= 16) memcpy(&in_file, c->inp->data + pk_at, 4);
                if (j->n > in_file) j->n = in_file;          /* batches are taken as the file holds them */
                const size_t used = packed_begin(&L->pack_pool, &L->cs, c->inp->data + pk_at, c->inp->size - pk_at, j, p->read_size, batch, c->max_raw, j->n);
                if (!used || j->n == 0) { fprintf(stderr, "'%s': malformed packed batch at byte %zu\n", c->in_name, pk_at); exit(1); }
                pk_at += used;
            } else {
                pack_begin(&L->pack_pool, &L->pk, c->inp, j, p->read_size, c->max_raw, c->no_pack);
            }
        }
        /* ... while the format pool prints the previous batch of this job (host.c:331-352) and writes it */
        if (have_old) {
            double t0 = now_ms();
            pool_run(&L->fmt_pool, format_range, f);
            d_fmt = now_ms() - t0;
            const size_t batch_at = out_at;
            for (int t = 0; t < L->fmt_pool.n; ++t) { f->off[t] = out_at; out_at += f->len[t]; }
            if (W->started) {   /* hand the printed batch to the writer; the next one is printed into the other buffer set */
                const double tw = now_ms();
                writer_submit(W, L->fmt_pool.n, f->buf, f->len, f->off);
                L->wrwait_ms += (d_wr = now_ms() - tw);
                if (W->failed) { fprintf(stderr, "Output file '%s' couldn't be written\n", L->out_path); exit(1); }
                f->set ^= 1;
                f->buf = f->bufs[f->set]; f->cap = f->caps[f->set]; f->len = f->lens[f->set]; f->off = f->offs[f->set];
            } else {
            /* Writes to one file serialise on its inode lock (measured: 2.5 GB/s however many threads call pwrite, and the output
               is 14-22 bytes per pair); a shared mapping of the batch's window lets all threads fill the page cache at once.
               Falls back to pwrite where the file cannot be extended / mapped; no regular file at all (pipes): out_seq above. */
            f->map = NULL;
            if (out_mmap && out_at > batch_at && ftruncate(L->out_fd, (off_t)out_at) == 0) {
                const size_t pg = (size_t)sysconf(_SC_PAGESIZE);
                f->map_base = batch_at / pg * pg;
                void *m = mmap(NULL, out_at - f->map_base, PROT_READ | PROT_WRITE, MAP_SHARED, L->out_fd, (off_t)f->map_base);
                if (m != MAP_FAILED) f->map = m; else out_mmap = 0;
            } else if (out_mmap && out_at > batch_at) out_mmap = 0;
            if (out_serial) { for (int t = 0; t < L->fmt_pool.n; ++t) write_range(t, L->fmt_pool.n, f); }
            else pool_run(&L->fmt_pool, write_range, f);
            if (f->map) { munmap(f->map, out_at - f->map_base); f->map = NULL; }
            }
            if (f->failed) { fprintf(stderr, "Output file '%s' couldn't be written\n", L->out_path); exit(1); }
            L->write_ms += now_ms() - t0;
            L->done += f->n;
            if (!L->pairs_first_done) { L->pairs_first_done = L->done; L->t_first_done = now_ms(); }
        }
        if (have_new) {
            const double tj = now_ms();
            if (c->packed_input) pool_join(&L->pack_pool);
            else pack_finish(&L->pack_pool, &L->pk, j, batch);
            L->join_ms += (d_join = now_ms() - tj);        /* what the pack still needed after the format stage was done */
            L->parse_ms += now_ms() - t_pack;              /* (overlaps the format + write above) */
            say_once(&g_said_copy, "Copying data to DPU\nRun program on DPU(s)\n");
            aim_batch_io_t io;
            memset(&io, 0, sizeof io);
            io.n_pairs = j->n;
            io.requests = j->req;
            if (j->ascii) { io.patterns = j->pat; io.texts = j->txt; }
            else {
                io.packed_patterns = j->pkP; io.packed_texts = j->pkT;
                io.n_raw = j->n_raw; io.raw_pairs = j->raw_idx; io.raw_patterns = j->rawP; io.raw_texts = j->rawT;
            }
            if (!backtrace) io.results = j->res8;
            else if (full_ops) { io.results = j->res; io.ops = j->ops; }
            else { io.cigars = j->cig; io.runs = j->runs; io.runs_cap = c->runs_cap; }
            j->io = io;
            j->use_full = 0;
            const double ts = now_ms();
            if (dry && j->dry_filled) { }
            else if (dry) {   /* (what the device would have returned, as far as the formatter's work goes: the right idx, score 0, "<n>M") */
                for (uint32_t q = 0; q < j->n && !backtrace; ++q) { j->res8[q].idx = (uint32_t)(j->first_pair + q); j->res8[q].score = 0; }
                for (uint32_t q = 0; q < j->n && backtrace && !full_ops; ++q) {
                    j->cig[q].idx = (uint32_t)(j->first_pair + q); j->cig[q].score = 0; j->cig[q].run_offset = q; j->cig[q].n_runs = 1; j->cig[q].status = 0;
                    j->runs[q] = (100u << 8) | 'M';
                }
                j->dry_filled = 1;
            } else if ((rc = aim_set_submit(L->set, j->device, j->slot, &io))) die_aim("aim_set_submit", rc);
            L->submit_ms += (d_sub = now_ms() - ts);
            j->in_flight = 1;
            sent += j->n;
            if (c->packed_input && sent < L->n_pairs && it + 1 == n_jobs) { fprintf(stderr, "'%s': fewer pairs than its header states\n", c->in_name); exit(1); }
        }
        if (trace) fprintf(stderr, "[lane %d it %llu] t %.3f total %.3f: wait %.3f format %.3f writer %.3f join %.3f submit %.3f\n", L->id, (unsigned long long)it, t_it - L->t_loop,
                           now_ms() - t_it, d_wait, d_fmt, d_wr, d_join, d_sub);
    }
    if (W->started) {
        writer_wait_idle(W);
        pthread_mutex_lock(&W->mu);
        W->stop = 1;
        pthread_cond_broadcast(&W->cv);
        pthread_mutex_unlock(&W->mu);
        pthread_join(W->th, NULL);
        if (W->failed) { fprintf(stderr, "Output file '%s' couldn't be written\n", L->out_path); exit(1); }
    }
    L->t_end = now_ms();
    aim_set_timers(L->set, &L->h2d, &L->kern, &L->d2h);
    pool_stop(&L->pack_pool);
    pool_stop(&L->fmt_pool);
    for (int t = 0; t < MAX_THREADS; ++t) { free(f->bufs[0][t]); free(f->bufs[1][t]); }
    for (uint32_t k = 0; k < ring; ++k) {
        job_t2 *j = &jobs[k];
        void *bufs[] = {j->req, j->pkP, j->pkT, j->raw_idx, j->rawP, j->rawT, j->pat, j->txt, j->res8, j->cig, j->runs, j->res, j->ops};
        for (size_t b = 0; b < sizeof bufs / sizeof bufs[0]; ++b)
            if (bufs[b]) aim_host_free(bufs[b]);
        free(j->is_raw);
    }
    free(jobs);
    aim_set_free(L->set);
    if (close(L->out_fd)) { fprintf(stderr, "Output file '%s' couldn't be written\n", L->out_path); exit(1); }
    return NULL;
}

int main(int argc, char *argv[])
{
    if (argc < 4) {
        printf("wrong number of arguments\n");
        exit(1);
    }
    char *in = argv[1], *out = argv[2];
    const long n_arg = atol(argv[3]);            /* signed: a negative count is "Invalid nb of reads", not 4 billion */

    /* defaults = the reference's common.h defaults for WFA (common.h:63-89), READ_SIZE rounded to 8 */
    static cfg_t cfg;
    aim_params_t p;
    memset(&p, 0, sizeof p);
    p.algo = AIM_ALGO_WFA;
    p.match = 0; p.mismatch = 3; p.gap_o = 4; p.gap_e = 1; p.gap_i = 4; p.gap_d = 4;
    p.max_score = 250; p.read_size = 112;
    uint32_t nr_dpus = 1, gpus = 1, batch = 4u << 20, slots = 2, shards = 0;
    int no_pack = 0, full_ops = 0, packed_input = 0, pin = -1;
    const char *pack_only = NULL;   /* write the packed batches to this file and exit (no GPU is touched) */
    int dev_ids[64], n_dev_ids = 0; /* --device-ids a,b,c: physical device of every set member (tests put one GPU in twice) */
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    int threads = 0;
    int pack_threads_arg = 0, fmt_threads_arg = 0;
    for (int i = 4; i < argc; ++i) {
        const char *f = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(f, "--backtrace")) p.flags |= AIM_FLAG_BACKTRACE;
        else if (!strcmp(f, "--reduce")) p.flags |= AIM_FLAG_REDUCE;
        else if (!strcmp(f, "--swg-w16")) p.flags |= AIM_FLAG_SWG_W16;
        else if (!strcmp(f, "--no-pack")) no_pack = 1;       /* ship ASCII rows like the reference (host.c:258-268) */
        else if (!strcmp(f, "--full-ops")) full_ops = 1;     /* gather result_t + ops rows like the reference (host.c:316-326) */
        else if (!strcmp(f, "--packed-input")) packed_input = 1;
        else if (!strcmp(f, "--pin")) pin = 1;               /* every lane's threads on its own contiguous share of the CPUs */
        else if (!strcmp(f, "--no-pin")) pin = 0;
        else if (!v) { printf("wrong number of arguments\n"); exit(1); }
        else if (!strcmp(f, "--pack-only")) { pack_only = v; ++i; }
        else if (!strcmp(f, "--device-ids")) {
            for (const char *q = v; *q && n_dev_ids < 64; ) { dev_ids[n_dev_ids++] = atoi(q); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
            gpus = (uint32_t)n_dev_ids;
            ++i;
        }
        else if (!strcmp(f, "--algo")) {
            if (!strcmp(v, "nw")) p.algo = AIM_ALGO_NW;
            else if (!strcmp(v, "swg")) p.algo = AIM_ALGO_SWG;
            else if (!strcmp(v, "wfa")) p.algo = AIM_ALGO_WFA;
            else if (!strcmp(v, "genasm")) p.algo = AIM_ALGO_GENASM;
            else { fprintf(stderr, "unknown --algo %s\n", v); exit(1); }
            ++i;
        }
        else if (!strcmp(f, "--max-score")) { p.max_score = atoi(v); ++i; }
        else if (!strcmp(f, "--read-size")) { p.read_size = atoi(v); ++i; }
        else if (!strcmp(f, "--match")) { p.match = atoi(v); ++i; }
        else if (!strcmp(f, "--mismatch")) { p.mismatch = atoi(v); ++i; }
        else if (!strcmp(f, "--gap-o")) { p.gap_o = atoi(v); ++i; }
        else if (!strcmp(f, "--gap-e")) { p.gap_e = atoi(v); ++i; }
        else if (!strcmp(f, "--gap")) { p.gap_i = p.gap_d = atoi(v); ++i; }
        else if (!strcmp(f, "--gap-i")) { p.gap_i = atoi(v); ++i; }   /* NW: -DGAP_I / -DGAP_D set apart (nw.c:67-153) */
        else if (!strcmp(f, "--gap-d")) { p.gap_d = atoi(v); ++i; }
        else if (!strcmp(f, "--nr-dpus")) { nr_dpus = (uint32_t)atoi(v); ++i; }
        else if (!strcmp(f, "--gpus")) { gpus = (uint32_t)atoi(v); ++i; }
        else if (!strcmp(f, "--batch")) { batch = (uint32_t)atoi(v); ++i; }
        else if (!strcmp(f, "--slots")) { slots = (uint32_t)atoi(v); ++i; }
        else if (!strcmp(f, "--threads")) { threads = atoi(v); ++i; }
        else if (!strcmp(f, "--pack-threads")) { pack_threads_arg = atoi(v); ++i; }
        else if (!strcmp(f, "--format-threads")) { fmt_threads_arg = atoi(v); ++i; }
        else if (!strcmp(f, "--out-shards")) { shards = (uint32_t)atoi(v); ++i; }
        else { fprintf(stderr, "unknown flag %s\n", f); exit(1); }
    }
    if (shards > 64) { fprintf(stderr, "--out-shards 1..64\n"); exit(1); }
    const uint32_t n_lanes = shards ? shards : 1;
    /* threads: per lane 48 at most (measured on a 256-thread host, 13.7 GB of text: 32 + 16 beats 64, 128 and 256 within ONE lane),
       all lanes together no more than the machine has */
    if (threads <= 0) {
        long per = ncpu < 1 ? 1 : (n_lanes > 1 ? ncpu / 2 : ncpu) / (long)n_lanes;   /* several lanes: one thread per physical core of an SMT-2 host in total
                                                                                     (4 lanes x (16 + 8) measured 1.5x faster than 4 x (32 + 16) on a loaded 256-thread box) */
        if (per > 48) per = 48;
        if (per < 3) per = 3;
        threads = (int)(per * n_lanes);
    }
    if (threads < 1) threads = 1;
    if (threads > MAX_THREADS) threads = MAX_THREADS;
    const int backtrace = (p.flags & AIM_FLAG_BACKTRACE) != 0;
    const int use_req8 = p.read_size < 32760;                   /* int16 lengths */
    if (use_req8) p.flags |= AIM_FLAG_REQ8;                     /* 8-byte WFA request_t on the wire (common.h:172-177) */
    if (!backtrace) p.flags |= AIM_FLAG_RES8;                   /* score-only: {idx, score} back */
    if (packed_input && (no_pack || pack_only)) { fprintf(stderr, "--packed-input cannot be combined with --no-pack / --pack-only\n"); exit(1); }
#if defined(__x86_64__)
    g_simd = __builtin_cpu_supports("sse4.1") && __builtin_cpu_supports("ssse3") && __builtin_cpu_supports("bmi2");
#endif

    int fd = open(in, O_RDONLY);
    /* one output file, or with --out-shards K the K files <output>.000 ... (cat in that order = the single file) */
    static lane_t *lanes;
    lanes = calloc(n_lanes, sizeof *lanes);
    if (!lanes) { fprintf(stderr, "out of host memory\n"); exit(1); }
    int out_bad = 0;
    for (uint32_t k = 0; k < n_lanes; ++k) {
        lane_t *L = &lanes[k];
        L->id = (int)k; L->c = &cfg;
        if (shards) snprintf(L->out_path, sizeof L->out_path, "%s.%03u", out, k);
        else snprintf(L->out_path, sizeof L->out_path, "%s", out);
        L->out_fd = open(L->out_path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
        if (L->out_fd < 0) out_bad = 1;
    }
    FILE *dpu_file = fopen("dpu-out", "w"); /* host.c:162: kept (empty) for scripts that expect it */
    if (fd < 0) { fprintf(stderr, "Input file '%s' couldn't be opened\n", in); exit(1); }
    if (out_bad) { fprintf(stderr, "Output file '%s' couldn't be opened\n", out); exit(1); }
    if (n_arg <= 0 || n_arg > 0x7fffffffL) { fprintf(stderr, "Invalid nb of reads\n"); exit(1); }
    const uint32_t total_nb_reads = (uint32_t)n_arg;
    if (nr_dpus == 0 || total_nb_reads <= nr_dpus) { printf("Allocated DPUs more than needed\n"); exit(1); }
    if (gpus == 0 || gpus > 64 || batch == 0 || slots == 0 || slots > 4) { fprintf(stderr, "--gpus 1..64, --batch must be positive, --slots 1..4\n"); exit(1); }

    g_big_alloc = pack_only ? plain : pinned;
    printf("Allocated %d DPU(s)\n", (int)nr_dpus);
    printf("AIM-HIP: %u MI355X device(s), kernel %s, %d host thread(s)\n", gpus, aim_kernel_name(&p), threads);

    uint32_t nb_reads_per_dpu = (uint32_t)ROUND_UP_MULTIPLE_8((total_nb_reads / nr_dpus));
    printf("NumReads per dpu = %u\n", nb_reads_per_dpu);
    const uint64_t pair_cap = (uint64_t)nb_reads_per_dpu * nr_dpus; /* H3: what the reference would consume */

    /* per lane, two pools: parse + pack of the next batch || format + write of the previous one. The index pool (all threads) maps
       and indexes the input once for all lanes. */
    const int lane_threads = threads / (int)n_lanes > 0 ? threads / (int)n_lanes : 1;
    int fmt_threads = lane_threads >= 3 ? lane_threads / 3 : 1, pack_threads = lane_threads >= 3 ? lane_threads - fmt_threads : lane_threads;   /* 32 + 16 of 48: on one box 3.3-3.5e8 pairs/s against 2.8-3.3e8 for 28 + 20 and 3.1-3.2e8 for 24 + 24 */
    if (pack_threads_arg > 0) pack_threads = pack_threads_arg;
    if (fmt_threads_arg > 0) fmt_threads = fmt_threads_arg;
    static pool_t idx_pool;
    int idx_threads = pack_threads * (int)n_lanes;
    if (idx_threads > MAX_THREADS) idx_threads = MAX_THREADS;
    pool_init(&idx_pool, idx_threads, NULL);

    /* map the input; text: index its lines (replaces the getline loop) and validate; packed file: read the header */
    double t_index = now_ms();
    static input_t inp;
    memset(&inp, 0, sizeof inp);
    struct stat st;
    if (fstat(fd, &st)) { fprintf(stderr, "Input file '%s' couldn't be opened\n", in); exit(1); }
    inp.size = (size_t)st.st_size;
    if (inp.size) {
        inp.data = mmap(NULL, inp.size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (inp.data == MAP_FAILED) { fprintf(stderr, "Input file '%s' couldn't be mapped\n", in); exit(1); }
        madvise((void *)inp.data, inp.size, MADV_SEQUENTIAL);
    }
    uint64_t total_pairs;
    double index_ms = 0;
    size_t pk_at = 0;                    /* packed input: byte offset of the next batch */
    if (packed_input) {
        pkfile_hdr_t fh;
        if (inp.size < sizeof fh) { fprintf(stderr, "'%s' is not a packed batch file\n", in); exit(1); }
        memcpy(&fh, inp.data, sizeof fh);
        if (memcmp(fh.magic, PKFILE_MAGIC, 8) || fh.version != 1) { fprintf(stderr, "'%s' is not a packed batch file (magic / version)\n", in); exit(1); }
        if (fh.read_size != (uint32_t)p.read_size || fh.req_bytes != (use_req8 ? 8u : 16u)) {
            printf("READ LENGTH less than length of the input reads");   /* the file was packed for another READ_SIZE (host.c:119-123) */
            exit(0);
        }
        total_pairs = fh.total_pairs < pair_cap ? fh.total_pairs : pair_cap;
        batch = fh.batch_pairs ? fh.batch_pairs : 1;           /* batches are taken as the file holds them */
        pk_at = sizeof fh;
    } else {
        inp.populate = !(getenv("AIM_HOST_POPULATE") && !strcmp(getenv("AIM_HOST_POPULATE"), "0"));
        index_lines(&inp, &idx_pool);
        index_ms = now_ms() - t_index;
        uint64_t pairs_in_file = inp.n_lines / 2;   /* a trailing unpaired line ends the reference's loop as well */
        total_pairs = pairs_in_file < pair_cap ? pairs_in_file : pair_cap;
        /* validate every pair the run will touch before the first launch */
        scan_t sc = {&inp, (size_t)total_pairs, p.read_size, 0, 0};
        pool_run(&idx_pool, scan_range, &sc);
        if (sc.too_long) { /* host.c:119-123 */
            printf("READ LENGTH less than length of the input reads");
            exit(0);
        }
        if (sc.malformed) { fprintf(stderr, "malformed input (a line shorter than 2 characters)\n"); exit(1); }
    }
    const double setup_ms = now_ms() - t_index;

    /* batches: enough of them to keep every (device, slot) of every lane busy, none larger than --batch */
    const uint32_t ring_all = (gpus > n_lanes ? gpus : n_lanes) * slots;
    if (!packed_input) {
        uint64_t want = (total_pairs + 2 * ring_all - 1) / (2 * ring_all);
        if (n_lanes > 1) {   /* a lane's pipeline fills and drains over `slots` batches: give it at least 8 (and less pinned memory to set up) */
            const uint64_t w8 = (total_pairs + 8 * n_lanes - 1) / (8 * n_lanes);
            if (w8 < want) want = w8;
            if (want < 262144) want = 262144;
        }
        if (want < 65536) want = 65536;
        if (want < batch) batch = (uint32_t)want;
        if ((uint64_t)batch > total_pairs) batch = (uint32_t)(total_pairs ? total_pairs : 1);
    }
    const size_t rs = (size_t)p.read_size;
    const uint32_t dw = (uint32_t)(p.read_size + 15) / 16u;
    /* device-side CIGAR runs: room for max(8, READ_SIZE/4 + 2) runs per pair (an alignment with e errors has <= 2e+1 runs; the
       launchers size READ_SIZE for e <= l*error), at most 1 GiB per job -- longer reads get smaller batches */
    const uint32_t rpp = (uint32_t)(p.read_size / 4 + 2 > 8 ? p.read_size / 4 + 2 : 8);
    if (!packed_input) {
        if (backtrace && !full_ops && (uint64_t)batch * rpp > (1ull << 28)) {
            batch = (uint32_t)((1ull << 28) / rpp);
            if (batch < 64) batch = 64;
        }
        /* also bound the pinned / device sequence buffers of one job to ~1 GiB (long reads) */
        while (batch > 64 && (uint64_t)batch * rs > (1ull << 30)) batch /= 2;
    }
    const uint32_t runs_cap = (backtrace && !full_ops) ? (uint32_t)((uint64_t)batch * rpp > 0xffffffffull ? 0xffffffffu : batch * rpp) : 0;
    const uint32_t max_raw = no_pack ? 0 : (batch / 16 < 1024 ? (batch < 1024 ? batch : 1024) : batch / 16);
    if (pack_only) {   /* a packed batch file: header, then per batch {n, ascii, n_raw, read_size} + the arrays the device would receive */
        job_t2 job, *j = &job;
        memset(j, 0, sizeof *j);
        j->req8 = use_req8;
        j->req = plain((size_t)batch * (use_req8 ? sizeof(aim_request8_t) : sizeof(aim_request_t)));
        j->is_raw = plain(batch);
        if (no_pack) { j->pat = plain((size_t)batch * rs); j->txt = plain((size_t)batch * rs); }
        else {
            j->pkP = plain((size_t)batch * dw * 4); j->pkT = plain((size_t)batch * dw * 4);
            j->raw_idx = plain((size_t)max_raw * 4); j->rawP = plain((size_t)max_raw * rs); j->rawT = plain((size_t)max_raw * rs);
        }
        static pack_t pk;
        FILE *df = fopen(pack_only, "wb");
        if (!df) { fprintf(stderr, "cannot write %s\n", pack_only); exit(1); }
        pkfile_hdr_t fh;
        memset(&fh, 0, sizeof fh);
        memcpy(fh.magic, PKFILE_MAGIC, 8);
        fh.version = 1; fh.read_size = (uint32_t)p.read_size; fh.req_bytes = use_req8 ? 8u : 16u; fh.batch_pairs = batch; fh.total_pairs = total_pairs;
        fwrite(&fh, sizeof fh, 1, df);
        uint64_t at = 0;
        while (at < total_pairs) {
            j->n = total_pairs - at < batch ? (uint32_t)(total_pairs - at) : batch;
            j->first_pair = (size_t)at;
            pack_begin(&idx_pool, &pk, &inp, j, p.read_size, max_raw, no_pack);
            pack_finish(&idx_pool, &pk, j, batch);
            uint32_t hdr[4] = {j->n, (uint32_t)j->ascii, j->n_raw, (uint32_t)p.read_size};
            fwrite(hdr, 4, 4, df);
            fwrite(j->req, j->req8 ? sizeof(aim_request8_t) : sizeof(aim_request_t), j->n, df);
            if (j->ascii) { fwrite(j->pat, rs, j->n, df); fwrite(j->txt, rs, j->n, df); }
            else {
                fwrite(j->pkP, (size_t)dw * 4, j->n, df); fwrite(j->pkT, (size_t)dw * 4, j->n, df);
                fwrite(j->raw_idx, 4, j->n_raw, df); fwrite(j->rawP, rs, j->n_raw, df); fwrite(j->rawT, rs, j->n_raw, df);
            }
            at += j->n;
        }
        if (ferror(df) | fclose(df)) { fprintf(stderr, "cannot write %s\n", pack_only); exit(1); }
        printf("AIM-HIP: packed %llu pairs, batch %u, max_raw %u\n", (unsigned long long)total_pairs, batch, max_raw);
        return 0;
    }
    pool_stop(&idx_pool);   /* the lanes bring their own threads */

    cfg.p = p; cfg.backtrace = backtrace; cfg.use_req8 = use_req8; cfg.no_pack = no_pack; cfg.full_ops = full_ops; cfg.packed_input = packed_input;
    cfg.batch = batch; cfg.slots = slots; cfg.max_raw = max_raw; cfg.runs_cap = runs_cap; cfg.inp = &inp; cfg.in_name = in; cfg.out_name = out;

    /* deal the input and the devices to the lanes: lane k takes the k-th contiguous run of whole batches */
    const uint64_t n_jobs = (total_pairs + batch - 1) / batch;
    {
        size_t at = pk_at;
        uint64_t pairs_at = 0;
        for (uint32_t k = 0; k < n_lanes; ++k) {
            lane_t *L = &lanes[k];
            const uint64_t j0 = n_jobs * k / n_lanes, j1 = n_jobs * (k + 1) / n_lanes;
            L->n_jobs = j1 - j0;
            L->first_pair = pairs_at;
            L->pk_at = at;
            if (packed_input) {   /* batches are taken as the file holds them: walk this lane's headers */
                uint64_t got = 0;
                for (uint64_t b = j0; b < j1; ++b) {
                    uint32_t nb = 0;
                    const size_t used = packed_batch_bytes(inp.data + at, inp.size - at, use_req8, p.read_size, &nb);
                    if (!used || nb == 0 || nb > batch) { fprintf(stderr, "'%s': malformed packed batch at byte %zu\n", in, at); exit(1); }
                    at += used;
                    got += nb;
                }
                if (pairs_at + got > total_pairs) got = total_pairs - pairs_at;   /* the partition rule may end inside the last batch */
                L->n_pairs = got;
            } else {
                const uint64_t lo = j0 * batch, hi = j1 * batch < total_pairs ? j1 * batch : total_pairs;
                L->n_pairs = hi > lo ? hi - lo : 0;
            }
            pairs_at += L->n_pairs;
            /* devices: round-robin over the lanes; more lanes than devices: lane k shares device k % gpus (a set of its own) */
            L->gpus = 0;
            if (n_lanes <= gpus) {
                for (uint32_t g = k; g < gpus; g += n_lanes) L->dev_ids[L->gpus++] = n_dev_ids ? dev_ids[g] : (int)g;
            } else {
                L->gpus = 1;
                L->dev_ids[0] = n_dev_ids ? dev_ids[k % gpus] : (int)(k % gpus);
            }
            L->pack_threads = pack_threads; L->fmt_threads = fmt_threads;
        }
        if (packed_input && pairs_at < total_pairs) { fprintf(stderr, "'%s': fewer pairs than its header states\n", in); exit(1); }
    }
    /* CPU placement (--pin; default on with several lanes): lane k's threads stay on the k-th contiguous share of the CPUs this
       process may use, so a lane's pinned buffers, its text buffers and the threads that fill them share a socket / NUMA node */
    if (pin < 0) pin = n_lanes > 1;
    if (pin) {
        cpu_set_t all;
        CPU_ZERO(&all);
        if (sched_getaffinity(0, sizeof all, &all) == 0) {
            int ids[CPU_SETSIZE], n = 0;
            for (int cpu = 0; cpu < CPU_SETSIZE; ++cpu) if (CPU_ISSET(cpu, &all)) ids[n++] = cpu;
            if (n >= (int)n_lanes)
                for (uint32_t k = 0; k < n_lanes; ++k) {
                    lane_t *L = &lanes[k];
                    CPU_ZERO(&L->cpus);
                    for (int q = (int)((long)n * k / n_lanes); q < (int)((long)n * (k + 1) / n_lanes); ++q) CPU_SET(ids[q], &L->cpus);
                    L->have_cpus = 1;
                }
        }
    }
    const double t_loop = now_ms();
    uint32_t started = 0, busy = 0;
    for (uint32_t k = 0; k < n_lanes; ++k) busy += lanes[k].n_jobs != 0;
    pthread_barrier_init(&g_lanes_ready, NULL, busy ? busy : 1);
    for (uint32_t k = 0; k < n_lanes; ++k) {
        lane_t *L = &lanes[k];
        if (L->n_jobs == 0) { close(L->out_fd); continue; }   /* (an empty shard file) */
        if (n_lanes == 1) lane_main(L);
        else if (pthread_create(&L->th, NULL, lane_main, L)) { fprintf(stderr, "cannot start lane %u\n", k); exit(1); }
        ++started;
    }
    if (n_lanes > 1)
        for (uint32_t k = 0; k < n_lanes; ++k)
            if (lanes[k].n_jobs) pthread_join(lanes[k].th, NULL);
    const double t_end = now_ms();
    if (!started) { aim_set_t *set = NULL; int rc = aim_set_alloc(1, n_dev_ids ? dev_ids : NULL, &set); if (rc) die_aim("aim_set_alloc", rc); aim_set_free(set); }   /* nothing to align: a run without a device still fails like the reference */
    /* device phases: lanes (sets) run side by side -- the slowest one; host phases: likewise the slowest lane's */
    float h2d = 0, kern = 0, d2h = 0;
    double parse_ms = setup_ms, write_ms = 0, wait_ms = 0, join_ms = 0, submit_ms = 0, wrwait_ms = 0, t_first = 0, loop_ms = 0, lane_setup_ms = 0, t_last = 0;
    uint64_t done = 0, first_done = 0;
    for (uint32_t k = 0; k < n_lanes; ++k) {
        const lane_t *L = &lanes[k];
        if (!L->n_jobs) continue;
        if (L->h2d > h2d) h2d = L->h2d;
        if (L->kern > kern) kern = L->kern;
        if (L->d2h > d2h) d2h = L->d2h;
        if (setup_ms + L->parse_ms > parse_ms) parse_ms = setup_ms + L->parse_ms;
        if (L->write_ms > write_ms) write_ms = L->write_ms;
        if (L->wait_ms > wait_ms) wait_ms = L->wait_ms;
        if (L->join_ms > join_ms) join_ms = L->join_ms;
        if (L->submit_ms > submit_ms) submit_ms = L->submit_ms;
        if (L->wrwait_ms > wrwait_ms) wrwait_ms = L->wrwait_ms;
        if (L->t_end - L->t_loop > loop_ms) loop_ms = L->t_end - L->t_loop;
        if (L->t_loop - t_loop > lane_setup_ms) lane_setup_ms = L->t_loop - t_loop;
        if (L->t_end > t_last) t_last = L->t_end;
        done += L->done; first_done += L->pairs_first_done;
        if (L->t_first_done > 0 && (t_first == 0 || L->t_first_done < t_first)) t_first = L->t_first_done;
    }
    if (n_jobs == 0) { printf("Copying data to DPU\n"); printf("Run program on DPU(s)\n"); printf("Retrieve results\n"); }
    printf("CPU-DPU: %f ms\n", h2d);
    printf("DPU Kernel: %f ms\n", kern);
    printf("DPU-CPU: %f ms\n", d2h);
    /* steady state: from the moment the first batch is on disk to the last one (start-up -- HIP context, pinned buffers, the
       pipeline filling -- excluded); with several lanes: from the first lane's first batch, every lane's first batch not counted */
    const double steady = (done > first_done && t_first > 0 && t_last > t_first) ? (double)(done - first_done) / ((t_last - t_first) * 1e-3) : 0.0;
    printf("AIM-HIP: %llu pairs in %llu batch(es) of <= %u over %u device(s) x %u slot(s); parse+pack %.3f ms (line index %.3f ms), wait %.3f ms, format+write %.3f ms, loop %.3f ms, steady %.4g pairs/s (in the loop: pack join %.3f ms, writer hand-over %.3f ms, submit %.3f ms); input %s, output %s; %u lane(s) x (%d + %d) threads%s, device set + pinned buffers %.3f ms%s\n",
           (unsigned long long)done, (unsigned long long)n_jobs, batch, gpus, slots, parse_ms, index_ms, wait_ms, write_ms, loop_ms, steady, join_ms, wrwait_ms, submit_ms,
           packed_input ? "packed batch file" : (no_pack ? "ASCII rows" : "packed 2 bit/base"), !backtrace ? "{idx, score}" : (full_ops ? "ops rows" : "device-side RLE"),
           n_lanes, pack_threads, fmt_threads, pin ? ", pinned" : "", lane_setup_ms, getenv("AIM_HOST_DRY") ? " [AIM_HOST_DRY: no device work, results void]" : "");
    (void)t_end;
    if (getenv("AIM_HOST_RUSAGE")) {   /* measurement aid: where the process's time went (user / system), page faults, context switches */
        struct rusage ru;
        getrusage(RUSAGE_SELF, &ru);
        fprintf(stderr, "[rusage] user %.3f s, system %.3f s, minor faults %ld, major faults %ld, voluntary switches %ld, involuntary %ld, max RSS %ld MB\n",
                ru.ru_utime.tv_sec + ru.ru_utime.tv_usec * 1e-6, ru.ru_stime.tv_sec + ru.ru_stime.tv_usec * 1e-6, ru.ru_minflt, ru.ru_majflt, ru.ru_nvcsw, ru.ru_nivcsw,
                ru.ru_maxrss / 1024);
    }

    free(lanes); free(inp.line_start);
    if (inp.size) munmap((void *)inp.data, inp.size);
    close(fd);
    if (dpu_file) fclose(dpu_file);
    return 0;
}
