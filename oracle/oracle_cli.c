/*
 * oracle_cli.c -- file-in / file-out driver around the CPU oracle.  TEST
 * INFRASTRUCTURE ONLY.  It restates the reference host program's parser,
 * partition rule and output writer so whole-file digests can be compared with
 * the digests recorded from the reference:
 *   get_reads          WFA/DPU-WRAM/host/host.c:91-134
 *   partition          WFA/DPU-WRAM/host/host.c:175-209
 *   output writer      WFA/DPU-WRAM/host/host.c:331-352
 *
 * usage: oracle_cli <nw|swg|wfa> -i IN -o OUT -n N -l LEN -e ERR
 *                   [-m M] [-x X] [-g G] [-a A] [-b] [-r] [-d NR_DPUS] [-t THREADS]
 *                   [--max-score S] [--read-size R] [--swg-cell 1|2]
 * The flag letters are those of run-*-pim-*.py.
 */
#define _GNU_SOURCE
#include "aim_oracle.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ROUND_UP_MULTIPLE_8(x) ((((x) + 7) / 8) * 8)

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: oracle_cli <nw|swg|wfa> -i IN -o OUT -n N -l LEN -e ERR [...]\n");
        return 2;
    }
    orc_params_t p;
    memset(&p, 0, sizeof p);
    if (!strcmp(argv[1], "nw")) p.algo = ORC_ALGO_NW;
    else if (!strcmp(argv[1], "swg")) p.algo = ORC_ALGO_SWG;
    else if (!strcmp(argv[1], "wfa")) p.algo = ORC_ALGO_WFA;
    else { fprintf(stderr, "unknown algorithm %s\n", argv[1]); return 2; }

    const char *in = NULL, *out = NULL;
    long n = 0;
    int len = 0, nr_dpus = 1, threads = 1, ms_override = -1, rs_override = -1;
    double err = 0.0;
    int m = 0, x = 3, g = 4, a = 1, gi = -1, gd = -1;
    for (int i = 2; i < argc; ++i) {
        const char *f = argv[i];
        const char *v = (i + 1 < argc) ? argv[i + 1] : NULL;
        if (!strcmp(f, "-b")) p.backtrace = 1;
        else if (!strcmp(f, "-r")) p.reduce = 1;
        else if (!v) { fprintf(stderr, "missing value for %s\n", f); return 2; }
        else if (!strcmp(f, "-i")) { in = v; ++i; }
        else if (!strcmp(f, "-o")) { out = v; ++i; }
        else if (!strcmp(f, "-n")) { n = atol(v); ++i; }
        else if (!strcmp(f, "-l")) { len = atoi(v); ++i; }
        else if (!strcmp(f, "-e")) { err = atof(v); ++i; }
        else if (!strcmp(f, "-m")) { m = atoi(v); ++i; }
        else if (!strcmp(f, "-x")) { x = atoi(v); ++i; }
        else if (!strcmp(f, "-g")) { g = atoi(v); ++i; }
        else if (!strcmp(f, "-a")) { a = atoi(v); ++i; }
        else if (!strcmp(f, "-d")) { nr_dpus = atoi(v); ++i; }
        else if (!strcmp(f, "-t")) { threads = atoi(v); ++i; }
        else if (!strcmp(f, "--max-score")) { ms_override = atoi(v); ++i; }
        else if (!strcmp(f, "--read-size")) { rs_override = atoi(v); ++i; }
        else if (!strcmp(f, "--gap-i")) { gi = atoi(v); ++i; }   /* NW: -DGAP_I / -DGAP_D set apart (nw.c:67-153) */
        else if (!strcmp(f, "--gap-d")) { gd = atoi(v); ++i; }
        else if (!strcmp(f, "--swg-cell")) { p.swg_cell_bytes = atoi(v); ++i; }
        else { fprintf(stderr, "unknown flag %s\n", f); return 2; }
    }
    if (!in || !out || n <= 0 || len <= 0) { fprintf(stderr, "need -i -o -n -l -e\n"); return 2; }
    p.match = m; p.mismatch = x; p.gap_o = g; p.gap_e = a; p.gap_i = gi >= 0 ? gi : g; p.gap_d = gd >= 0 ? gd : g;
    orc_launcher_sizes(p.algo, len, err, x, g, a, g, &p.max_score, &p.read_size);
    if (ms_override >= 0) p.max_score = ms_override;
    if (rs_override >= 0) p.read_size = rs_override;

    FILE *fi = fopen(in, "r");
    FILE *fo = fopen(out, "w");
    if (!fi) { fprintf(stderr, "Input file '%s' couldn't be opened\n", in); return 1; }
    if (!fo) { fprintf(stderr, "Output file '%s' couldn't be opened\n", out); return 1; }
    if (n <= nr_dpus) { printf("Allocated DPUs more than needed\n"); return 1; } /* host.c:180-184 */

    /* host.c:191: n is not a cap; the file is consumed in nr_dpus blocks of npd pairs */
    uint32_t npd = (uint32_t)ROUND_UP_MULTIPLE_8(((uint32_t)n / (uint32_t)nr_dpus));
    size_t cap = (size_t)npd * (size_t)nr_dpus;
    const size_t rs = (size_t)p.read_size;
    char *patterns = calloc(cap + 1, rs);
    char *texts = calloc(cap + 1, rs);
    int32_t *plen = calloc(cap, sizeof *plen), *tlen = calloc(cap, sizeof *tlen);
    if (!patterns || !texts || !plen || !tlen) { fprintf(stderr, "out of memory\n"); return 1; }

    char *line1 = NULL, *line2 = NULL;
    size_t a1 = 0, a2 = 0;
    size_t count = 0;
    while (count < cap) { /* get_reads, host.c:103-131 */
        ssize_t l1 = getline(&line1, &a1, fi);
        if (l1 == -1) break;
        ssize_t l2 = getline(&line2, &a2, fi);
        if (l2 == -1) break;
        int pl = (int)l1 - 2, tl = (int)l2 - 2;
        if (pl < 0 || tl < 0) { fprintf(stderr, "malformed line at pair %zu\n", count); return 1; }
        if (tl > p.read_size || pl > p.read_size) {
            printf("READ LENGTH less than length of the input reads");
            return 0; /* exit(0) in the reference */
        }
        memcpy(patterns + count * rs, line1 + 1, (size_t)pl);
        memcpy(texts + count * rs, line2 + 1, (size_t)tl);
        plen[count] = pl;
        tlen[count] = tl;
        ++count;
    }
    free(line1);
    free(line2);
    fclose(fi);

    orc_result_t *res = calloc(count ? count : 1, sizeof *res);
    char *ops = p.backtrace ? malloc((count ? count : 1) * 2 * rs) : NULL;
    int worst = orc_align_batch(&p, (uint32_t)count, plen, tlen, patterns, texts, res, ops, threads);
    if (worst == ORC_ERR_WFA_NO_LINK) { printf("Backtrace error: No link found during backtrace\n"); return 1; }
    if (worst == ORC_ERR_SWG_NO_OP) { printf("SWG backtrace. No backtrace operation found"); return 1; }
    if (worst != ORC_OK) { fprintf(stderr, "oracle failure %d\n", worst); return 1; }

    size_t linecap = 4 * rs + 64;
    char *line = malloc(linecap);
    for (size_t i = 0; i < count; ++i) { /* host.c:339-349 */
        fprintf(fo, "%d, %d, \n", (int)i, res[i].score);
        if (p.backtrace) {
            int w = orc_cigar_format(ops + i * 2 * rs, res[i].begin_offset, res[i].end_offset, line,
                                     (int)linecap);
            if (w < 0) { fprintf(stderr, "cigar overflow\n"); return 1; }
            fwrite(line, 1, (size_t)w, fo);
        }
    }
    fclose(fo);
    free(line); free(ops); free(res); free(patterns); free(texts); free(plen); free(tlen);
    return 0;
}
