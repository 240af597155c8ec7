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
 * The UPMEM dispatch (dpu_alloc/dpu_load/dpu_push_xfer/dpu_launch) is replaced
 * by aim_set_* calls; there is no CPU path.
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "aim_hip.h"

#define ROUND_UP_MULTIPLE_8(x) ((((x) + 7) / 8) * 8)

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

/* get_reads, host.c:91-134: fills up to `want` pairs starting at slot 0 */
static uint32_t get_reads(FILE *in, aim_request_t *req, char *patterns, char *texts, uint32_t want, int read_size,
                          uint32_t nb_sent_requests, char **line1, size_t *cap1, char **line2, size_t *cap2)
{
    uint32_t nb_reads;
    for (nb_reads = 0; nb_reads < want; ++nb_reads) {
        ssize_t l1 = getline(line1, cap1, in);
        if (l1 == -1) break;
        ssize_t l2 = getline(line2, cap2, in);
        if (l2 == -1) break;
        int pattern_length = (int)l1 - 2, text_length = (int)l2 - 2;
        if (text_length > read_size || pattern_length > read_size) {
            printf("READ LENGTH less than length of the input reads");
            exit(0);
        }
        if (pattern_length < 0 || text_length < 0) {
            fprintf(stderr, "malformed input at pair %u\n", nb_reads + nb_sent_requests);
            exit(1);
        }
        char *p = patterns + (size_t)nb_reads * read_size, *t = texts + (size_t)nb_reads * read_size;
        memcpy(p, *line1 + 1, (size_t)pattern_length);
        memset(p + pattern_length, 0, (size_t)(read_size - pattern_length));
        memcpy(t, *line2 + 1, (size_t)text_length);
        memset(t + text_length, 0, (size_t)(read_size - text_length));
        req[nb_reads].pattern_len = pattern_length;
        req[nb_reads].text_len = text_length;
        req[nb_reads].padding = 0;
        req[nb_reads].idx = nb_reads + nb_sent_requests;
    }
    return nb_reads;
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
        else { fprintf(stderr, "unknown flag %s\n", f); exit(1); }
    }
    const int backtrace = (p.flags & AIM_FLAG_BACKTRACE) != 0;

    FILE *input_file = fopen(in, "r");
    FILE *output_file = fopen(out, "w");
    FILE *dpu_file = fopen("dpu-out", "w"); /* host.c:162: kept (empty) for scripts that expect it */
    if (input_file == NULL) { fprintf(stderr, "Input file '%s' couldn't be opened\n", in); exit(1); }
    if (output_file == NULL) { fprintf(stderr, "Output file '%s' couldn't be opened\n", out); exit(1); }
    if (total_nb_reads <= 0) { fprintf(stderr, "Invalid nb of reads\n"); exit(1); }
    if (nr_dpus == 0 || total_nb_reads <= nr_dpus) { printf("Allocated DPUs more than needed\n"); exit(1); }
    if (gpus == 0 || batch == 0) { fprintf(stderr, "--gpus and --batch must be positive\n"); exit(1); }

    aim_set_t *set = NULL;
    int rc = aim_set_alloc(gpus, NULL, &set);
    if (rc) die_aim("aim_set_alloc", rc);
    printf("Allocated %d DPU(s)\n", (int)nr_dpus);
    printf("AIM-HIP: %u MI355X device(s), kernel %s\n", gpus, aim_kernel_name(&p));

    uint32_t nb_reads_per_dpu = (uint32_t)ROUND_UP_MULTIPLE_8((total_nb_reads / nr_dpus));
    printf("NumReads per dpu = %u\n", nb_reads_per_dpu);
    const uint64_t pair_cap = (uint64_t)nb_reads_per_dpu * nr_dpus; /* H3: what the reference would consume */

    if ((uint64_t)batch > (pair_cap + gpus - 1) / gpus) batch = (uint32_t)((pair_cap + gpus - 1) / gpus);
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

    char *line1 = NULL, *line2 = NULL;
    size_t cap1 = 0, cap2 = 0;
    size_t cig_cap = 8 * rs + 64;
    char *cig = malloc(cig_cap);
    uint64_t sent = 0;
    double parse_ms = 0, write_ms = 0;
    int first = 1, eof = 0;
    while (!eof && sent < pair_cap) {
        double t0 = now_ms();
        uint32_t got_total = 0;
        for (uint32_t g = 0; g < gpus; ++g) {
            uint64_t left = pair_cap - sent;
            uint32_t want = left < batch ? (uint32_t)left : batch;
            cnt[g] = eof ? 0 : get_reads(input_file, req[g], pat[g], txt[g], want, p.read_size, (uint32_t)sent, &line1,
                                         &cap1, &line2, &cap2);
            if (cnt[g] < want) eof = 1;
            sent += cnt[g];
            got_total += cnt[g];
        }
        parse_ms += now_ms() - t0;
        if (got_total == 0) break;
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
            for (uint32_t i = 0; i < cnt[g]; ++i) {
                fprintf(output_file, "%d, %d, \n", (int)res[g][i].idx, res[g][i].score);
                if (backtrace) {
                    int w = aim_cigar_format(ops[g] + (size_t)i * 2 * rs, res[g][i].begin_offset, res[g][i].end_offset, cig,
                                             (int)cig_cap);
                    if (w < 0) die_aim("aim_cigar_format", w);
                    fwrite(cig, 1, (size_t)w, output_file);
                }
            }
        }
        write_ms += now_ms() - t0;
    }
    float h2d = 0, kern = 0, d2h = 0;
    aim_set_timers(set, &h2d, &kern, &d2h);
    printf("CPU-DPU: %f ms\n", h2d);
    printf("DPU Kernel: %f ms\n", kern);
    printf("DPU-CPU: %f ms\n", d2h);
    printf("AIM-HIP: %llu pairs, parse %.3f ms, write %.3f ms\n", (unsigned long long)sent, parse_ms, write_ms);

    for (uint32_t g = 0; g < gpus; ++g) {
        aim_host_free(req[g]); aim_host_free(res[g]); aim_host_free(pat[g]); aim_host_free(txt[g]);
        if (ops[g]) aim_host_free(ops[g]);
    }
    free(req); free(res); free(pat); free(txt); free(ops); free(cnt); free(cig); free(line1); free(line2);
    aim_set_free(set);
    fclose(input_file);
    if (dpu_file) fclose(dpu_file);
    fclose(output_file);
    return 0;
}
