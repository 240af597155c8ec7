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
 *     --gap-o G --gap-e A --gap G  --backtrace  --reduce  --swg-w16
 *     --nr-dpus D   logical partition count of the reference (host.c:191: the
 *                   file is consumed in D blocks of ROUND_UP_8(n/D) pairs; n is
 *                   not a cap) -- kept so the set of aligned pairs is identical
 *     --gpus N      physical MI355X devices to shard each batch over
 *     --batch B     pairs per device per launch (default 4194304)
 *     --threads T   host threads for parsing / formatting (default: all cores, max 64)
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
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <unistd.h>

#include "aim_hip.h"

#define ROUND_UP_MULTIPLE_8(x) ((((x) + 7) / 8) * 8)
#define MAX_THREADS 64

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

/* ---- tiny fork/join helper ------------------------------------------------------------------------ */
typedef void (*range_fn)(int tid, int nthreads, void *arg);
typedef struct { range_fn fn; int tid, nthreads; void *arg; } job_t;
static void *job_tramp(void *p) { job_t *j = p; j->fn(j->tid, j->nthreads, j->arg); return NULL; }
static void parallel_run(int nthreads, range_fn fn, void *arg)
{
    pthread_t th[MAX_THREADS];
    job_t jobs[MAX_THREADS];
    memset(jobs, 0, sizeof jobs);
    if (nthreads < 1) nthreads = 1;
    for (int t = 0; t < nthreads; ++t) {
        jobs[t] = (job_t){fn, t, nthreads, arg};
        if (t) pthread_create(&th[t], NULL, job_tramp, &jobs[t]);
    }
    job_tramp(&jobs[0]);
    for (int t = 1; t < nthreads; ++t) pthread_join(th[t], NULL);
}

/* ---- input: mapped file + line index --------------------------------------------------------------- */
typedef struct {
    const char *data;
    size_t size;
    size_t *line_start;   /* [n_lines + 1]; line i = [line_start[i], line_start[i+1]) including its '\n' if any */
    size_t n_lines;
    size_t counts[MAX_THREADS + 1];
} input_t;

static void count_newlines(int tid, int nt, void *arg)
{
    input_t *in = arg;
    size_t lo = in->size * tid / nt, hi = in->size * (tid + 1) / nt, c = 0;
    const char *p = in->data + lo, *e = in->data + hi;
    while (p < e && (p = memchr(p, '\n', (size_t)(e - p)))) { ++c; ++p; }
    in->counts[tid + 1] = c;
}
static void fill_newlines(int tid, int nt, void *arg)
{
    input_t *in = arg;
    size_t lo = in->size * tid / nt, hi = in->size * (tid + 1) / nt;
    size_t at = in->counts[tid] + 1;   /* line_start[0] = 0; each '\n' at position x starts a line at x+1 */
    const char *p = in->data + lo, *e = in->data + hi;
    while (p < e && (p = memchr(p, '\n', (size_t)(e - p)))) { in->line_start[at++] = (size_t)(p - in->data) + 1; ++p; }
}
static void index_lines(input_t *in, int nthreads)
{
    in->counts[0] = 0;
    parallel_run(nthreads, count_newlines, in);
    for (int t = 0; t < nthreads; ++t) in->counts[t + 1] += in->counts[t];
    size_t n_nl = in->counts[nthreads];
    in->line_start = malloc((n_nl + 2) * sizeof(size_t));
    in->line_start[0] = 0;
    parallel_run(nthreads, fill_newlines, in);
    /* a final line without '\n' still is a line for getline() */
    in->n_lines = n_nl;
    if (in->size > 0 && in->data[in->size - 1] != '\n') in->line_start[++in->n_lines] = in->size;
}

/* ---- get_reads (host.c:91-134) over a contiguous range of pairs, in parallel ------------------------ */
typedef struct {
    const input_t *in;
    size_t first_pair;       /* global index of slot 0 */
    uint32_t n;              /* pairs to pack */
    int read_size;
    aim_request_t *req;
    char *pat, *txt;
    int too_long, malformed;
} pack_t;

static void pack_range(int tid, int nt, void *arg)
{
    pack_t *pk = arg;
    const size_t lo = (size_t)pk->n * tid / nt, hi = (size_t)pk->n * (tid + 1) / nt;
    const int rs = pk->read_size;
    for (size_t i = lo; i < hi; ++i) {
        const size_t pair = pk->first_pair + i;
        const size_t *ls = pk->in->line_start + 2 * pair;
        /* getline length includes the '\n'; the first character and the last one are dropped (H1) */
        const long pl = (long)(ls[1] - ls[0]) - 2, tl = (long)(ls[2] - ls[1]) - 2;
        if (pl > rs || tl > rs) { pk->too_long = 1; continue; }
        if (pl < 0 || tl < 0) { pk->malformed = 1; continue; }
        char *p = pk->pat + i * rs, *t = pk->txt + i * rs;
        memcpy(p, pk->in->data + ls[0] + 1, (size_t)pl);
        memset(p + pl, 0, (size_t)(rs - pl));
        memcpy(t, pk->in->data + ls[1] + 1, (size_t)tl);
        memset(t + tl, 0, (size_t)(rs - tl));
        pk->req[i].pattern_len = (int32_t)pl;
        pk->req[i].text_len = (int32_t)tl;
        pk->req[i].padding = 0;
        pk->req[i].idx = (uint32_t)pair;
    }
}

/* ---- output loop (host.c:331-352) ---------------------------------------------------------------- */
typedef struct {
    uint32_t n;
    int backtrace, read_size;
    const aim_result_t *res;
    const char *ops;
    char *buf[MAX_THREADS];
    size_t len[MAX_THREADS];
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
    const size_t lo = (size_t)f->n * tid / nt, hi = (size_t)f->n * (tid + 1) / nt;
    const size_t rs = (size_t)f->read_size;
    /* worst case per pair: "idx, score, \n" (<= 26 bytes) + one "%d%c" per op (<= 2 bytes per op when every run is 1) + '\n' */
    size_t cap = (hi - lo) * (32 + (f->backtrace ? 4 * rs + 16 : 0)) + 64;
    char *o = f->buf[tid] = malloc(cap), *start = o;
    for (size_t i = lo; i < hi; ++i) {
        const aim_result_t *r = &f->res[i];
        o = put_int(o, (int)r->idx); *o++ = ','; *o++ = ' ';       /* fprintf(out, "%d, %d, \n", idx, score) */
        o = put_int(o, r->score); *o++ = ','; *o++ = ' '; *o++ = '\n';
        if (f->backtrace) {                                          /* edit_cigar_print, host.c:69-89 */
            const char *ops = f->ops + i * 2 * rs;
            char last = ops[r->begin_offset];
            int run = 1;
            for (int k = r->begin_offset + 1; k < r->end_offset; ++k) {
                if (ops[k] == last) ++run;
                else { o = put_int(o, run); *o++ = last; last = ops[k]; run = 1; }
            }
            o = put_int(o, run); *o++ = last; *o++ = '\n';
        }
    }
    f->len[tid] = (size_t)(o - start);
}

int main(int argc, char *argv[])
{
    if (argc < 4) {
        printf("wrong number of arguments\n");
        exit(1);
    }
    char *in = argv[1], *out = argv[2];
    uint32_t total_nb_reads = (uint32_t)atoi(argv[3]);

    /* defaults = the reference's common.h defaults for WFA (common.h:63-89), READ_SIZE rounded to 8 */
    aim_params_t p;
    memset(&p, 0, sizeof p);
    p.algo = AIM_ALGO_WFA;
    p.match = 0; p.mismatch = 3; p.gap_o = 4; p.gap_e = 1; p.gap_i = 4; p.gap_d = 4;
    p.max_score = 250; p.read_size = 112;
    uint32_t nr_dpus = 1, gpus = 1, batch = 4u << 20;
    long ncpu = sysconf(_SC_NPROCESSORS_ONLN);
    int threads = (int)(ncpu < 1 ? 1 : (ncpu > MAX_THREADS ? MAX_THREADS : ncpu));
    for (int i = 4; i < argc; ++i) {
        const char *f = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(f, "--backtrace")) p.flags |= AIM_FLAG_BACKTRACE;
        else if (!strcmp(f, "--reduce")) p.flags |= AIM_FLAG_REDUCE;
        else if (!strcmp(f, "--swg-w16")) p.flags |= AIM_FLAG_SWG_W16;
        else if (!v) { printf("wrong number of arguments\n"); exit(1); }
        else if (!strcmp(f, "--algo")) {
            if (!strcmp(v, "nw")) p.algo = AIM_ALGO_NW;
            else if (!strcmp(v, "swg")) p.algo = AIM_ALGO_SWG;
            else if (!strcmp(v, "wfa")) p.algo = AIM_ALGO_WFA;
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
        else if (!strcmp(f, "--nr-dpus")) { nr_dpus = (uint32_t)atoi(v); ++i; }
        else if (!strcmp(f, "--gpus")) { gpus = (uint32_t)atoi(v); ++i; }
        else if (!strcmp(f, "--batch")) { batch = (uint32_t)atoi(v); ++i; }
        else if (!strcmp(f, "--threads")) { threads = atoi(v); ++i; }
        else { fprintf(stderr, "unknown flag %s\n", f); exit(1); }
    }
    if (threads < 1) threads = 1;
    if (threads > MAX_THREADS) threads = MAX_THREADS;
    const int backtrace = (p.flags & AIM_FLAG_BACKTRACE) != 0;

    int fd = open(in, O_RDONLY);
    FILE *output_file = fopen(out, "w");
    FILE *dpu_file = fopen("dpu-out", "w"); /* host.c:162: kept (empty) for scripts that expect it */
    if (fd < 0) { fprintf(stderr, "Input file '%s' couldn't be opened\n", in); exit(1); }
    if (output_file == NULL) { fprintf(stderr, "Output file '%s' couldn't be opened\n", out); exit(1); }
    if (total_nb_reads <= 0) { fprintf(stderr, "Invalid nb of reads\n"); exit(1); }
    if (nr_dpus == 0 || total_nb_reads <= nr_dpus) { printf("Allocated DPUs more than needed\n"); exit(1); }
    if (gpus == 0 || batch == 0) { fprintf(stderr, "--gpus and --batch must be positive\n"); exit(1); }

    aim_set_t *set = NULL;
    int rc = aim_set_alloc(gpus, NULL, &set);
    if (rc) die_aim("aim_set_alloc", rc);
    printf("Allocated %d DPU(s)\n", (int)nr_dpus);
    printf("AIM-HIP: %u MI355X device(s), kernel %s, %d host thread(s)\n", gpus, aim_kernel_name(&p), threads);

    uint32_t nb_reads_per_dpu = (uint32_t)ROUND_UP_MULTIPLE_8((total_nb_reads / nr_dpus));
    printf("NumReads per dpu = %u\n", nb_reads_per_dpu);
    const uint64_t pair_cap = (uint64_t)nb_reads_per_dpu * nr_dpus; /* H3: what the reference would consume */

    /* map + index the input (replaces the getline loop) */
    double t_index = now_ms();
    input_t inp;
    memset(&inp, 0, sizeof inp);
    struct stat st;
    if (fstat(fd, &st)) { fprintf(stderr, "Input file '%s' couldn't be opened\n", in); exit(1); }
    inp.size = (size_t)st.st_size;
    if (inp.size) {
        inp.data = mmap(NULL, inp.size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (inp.data == MAP_FAILED) { fprintf(stderr, "Input file '%s' couldn't be mapped\n", in); exit(1); }
        madvise((void *)inp.data, inp.size, MADV_SEQUENTIAL);
    }
    index_lines(&inp, threads);
    uint64_t pairs_in_file = inp.n_lines / 2;   /* a trailing unpaired line ends the reference's loop as well */
    const uint64_t total_pairs = pairs_in_file < pair_cap ? pairs_in_file : pair_cap;
    double parse_ms = now_ms() - t_index, write_ms = 0;

    if ((uint64_t)batch > (total_pairs + gpus - 1) / gpus) batch = (uint32_t)((total_pairs + gpus - 1) / gpus);
    if (batch == 0) batch = 1;
    rc = aim_set_configure(set, &p, batch);
    if (rc) {
        if (rc == AIM_EINVAL) { printf("%s\n", aim_last_error()); exit(1); }
        die_aim("aim_set_configure", rc);
    }
    const size_t rs = (size_t)p.read_size;
    aim_request_t **req = calloc(gpus, sizeof *req);
    aim_result_t **res = calloc(gpus, sizeof *res);
    char **pat = calloc(gpus, sizeof *pat), **txt = calloc(gpus, sizeof *txt), **ops = calloc(gpus, sizeof *ops);
    uint32_t *cnt = calloc(gpus, sizeof *cnt);
    for (uint32_t g = 0; g < gpus; ++g) {
        if ((rc = aim_host_alloc((void **)&req[g], (size_t)batch * sizeof(aim_request_t))) ||
            (rc = aim_host_alloc((void **)&res[g], (size_t)batch * sizeof(aim_result_t))) ||
            (rc = aim_host_alloc((void **)&pat[g], (size_t)batch * rs)) ||
            (rc = aim_host_alloc((void **)&txt[g], (size_t)batch * rs)) ||
            (backtrace && (rc = aim_host_alloc((void **)&ops[g], (size_t)batch * 2 * rs))))
            die_aim("aim_host_alloc", rc);
    }

    uint64_t sent = 0;
    int first = 1;
    while (sent < total_pairs) {
        double t0 = now_ms();
        for (uint32_t g = 0; g < gpus; ++g) {
            uint64_t left = total_pairs - sent;
            cnt[g] = left < batch ? (uint32_t)left : batch;
            if (cnt[g]) {
                pack_t pk = {&inp, (size_t)sent, cnt[g], p.read_size, req[g], pat[g], txt[g], 0, 0};
                parallel_run(threads, pack_range, &pk);
                if (pk.too_long) { /* host.c:119-123 */
                    printf("READ LENGTH less than length of the input reads");
                    exit(0);
                }
                if (pk.malformed) { fprintf(stderr, "malformed input near pair %llu\n", (unsigned long long)sent); exit(1); }
            }
            sent += cnt[g];
        }
        parse_ms += now_ms() - t0;
        if (first) printf("Copying data to DPU\n");
        for (uint32_t g = 0; g < gpus; ++g)
            if ((rc = aim_set_push(set, g, cnt[g], req[g], pat[g], txt[g]))) die_aim("aim_set_push", rc);
        if (first) printf("Run program on DPU(s)\n");
        if ((rc = aim_set_launch(set))) die_aim("aim_set_launch", rc);
        if (first) printf("Retrieve results\n");
        for (uint32_t g = 0; g < gpus; ++g) {
            rc = aim_set_pull(set, g, res[g], ops[g]);
            if (rc == AIM_EALIGN) { /* the reference prints from the DPU and exits 1 */
                const char *msg = strstr(aim_last_error(), "(");
                printf("%s\n", msg ? msg + 1 : aim_last_error());
                exit(1);
            }
            if (rc) die_aim("aim_set_pull", rc);
        }
        first = 0;
        t0 = now_ms();
        for (uint32_t g = 0; g < gpus; ++g) { /* host.c:331-352 */
            if (!cnt[g]) continue;
            fmt_t f;
            memset(&f, 0, sizeof f);
            f.n = cnt[g]; f.backtrace = backtrace; f.read_size = p.read_size; f.res = res[g]; f.ops = ops[g];
            parallel_run(threads, format_range, &f);
            for (int t = 0; t < threads; ++t) {
                if (f.len[t]) fwrite(f.buf[t], 1, f.len[t], output_file);
                free(f.buf[t]);
            }
        }
        write_ms += now_ms() - t0;
    }
    float h2d = 0, kern = 0, d2h = 0;
    aim_set_timers(set, &h2d, &kern, &d2h);
    if (first) { printf("Copying data to DPU\n"); printf("Run program on DPU(s)\n"); printf("Retrieve results\n"); }
    printf("CPU-DPU: %f ms\n", h2d);
    printf("DPU Kernel: %f ms\n", kern);
    printf("DPU-CPU: %f ms\n", d2h);
    printf("AIM-HIP: %llu pairs, parse %.3f ms, write %.3f ms\n", (unsigned long long)sent, parse_ms, write_ms);

    for (uint32_t g = 0; g < gpus; ++g) {
        aim_host_free(req[g]); aim_host_free(res[g]); aim_host_free(pat[g]); aim_host_free(txt[g]);
        if (ops[g]) aim_host_free(ops[g]);
    }
    free(req); free(res); free(pat); free(txt); free(ops); free(cnt); free(inp.line_start);
    if (inp.size) munmap((void *)inp.data, inp.size);
    close(fd);
    aim_set_free(set);
    if (dpu_file) fclose(dpu_file);
    fclose(output_file);
    return 0;
}
