/*
 * aim_oracle.h -- CPU restatement of AIM's per-pair alignment kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (aim_amd/, include/,
 * the host CLI) may include, link or call this.  Only tests/, the smoke check
 * in __graft_entry__.py and bench.py's cpu_baseline leg use it, and there only
 * as the checker / the reported CPU baseline.
 *
 * Pinning status: the reference (safaad/aim) ships no tests and no expected
 * outputs, and its kernels cannot be compiled in this image (they need the
 * UPMEM SDK headers <dpu.h>, <mram.h>, <defs.h>, <alloc.h>, which are absent;
 * writing stand-ins for them is not allowed).  The only reference-produced
 * data available are the output digests and score histograms recorded in
 * SURVEY.md section 8a / BASELINE.md section 2 for Datasets/sample-l100-e1-40K
 * (WFA/SWG+CIGAR, NW+CIGAR, WFA score-only).  This oracle reproduces all of
 * them (tests/test_oracle_golden.py).  Everything outside those digests
 * (higher error rates, plen>tlen aliasing, long reads, MAX_SCORE overflow) is
 * a line-by-line restatement that is *parity unpinned* by executed reference
 * output; see DESIGN.md "Oracle".
 */
#ifndef AIM_ORACLE_H
#define AIM_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { ORC_ALGO_NW = 0, ORC_ALGO_SWG = 1, ORC_ALGO_WFA = 2, ORC_ALGO_GENASM = 3 /* genasm_oracle.c: PARITY UNPINNED */ };

enum {
    ORC_OK = 0,
    ORC_ERR_WFA_NO_LINK = 1,   /* wfa_backtracing.c:321-325 prints + exit(1) */
    ORC_ERR_SWG_NO_OP = 2,     /* swg.c:99-104 prints + exit(1)              */
    ORC_ERR_NOMEM = 3
};

/* Compile-time -D configuration of the reference, carried at run time.
 * WFA/DPU-WRAM/common/common.h:63-89, NW/DPU-WRAM/common/common.h:63-85,
 * SWG/DPU-WRAM/common/common.h:55-86 */
typedef struct orc_params {
    int algo;        /* ORC_ALGO_* */
    int match;       /* MATCH    (ignored by NW: nw.c:143 uses literal 0; unused by WFA) */
    int mismatch;    /* MISMATCH */
    int gap_o;       /* GAP_O (SWG, WFA) */
    int gap_e;       /* GAP_E (SWG, WFA) */
    int gap_i;       /* GAP_I (NW) */
    int gap_d;       /* GAP_D (NW) */
    int max_score;   /* MAX_SCORE */
    int read_size;   /* READ_SIZE */
    int backtrace;   /* -DBACKTRACE */
    int reduce;      /* -DREDUCE (WFA-adaptive) */
    int swg_cell_bytes; /* 0: pick like SWG/DPU-WRAM (1 if MAX_SCORE<127 else 2); 1 or 2 forces it.
                           SWG/DPU-MRAM is always 2 (SWG/DPU-MRAM/common/common.h:91). */
} orc_params_t;

/* Mirrors result_t minus the unused cycles/padding fields
 * (WFA/DPU-WRAM/common/common.h:179-187). */
typedef struct orc_result {
    int32_t max_operations;
    int32_t begin_offset;
    int32_t end_offset;
    int32_t score;
    uint32_t idx;
    int32_t status;   /* ORC_OK or the reference's abort condition */
} orc_result_t;

/* One pair.  `ops` must hold at least max(2*read_size, plen+tlen) bytes when
 * p->backtrace is set (ignored otherwise).  Returns status (also in res). */
int orc_align_pair(const orc_params_t *p, const char *pattern, int plen,
                   const char *text, int tlen, char *ops, orc_result_t *res);

/* GenASM (genasm_oracle.c): windowed Bitap distance + traceback, the published algorithm (parity unpinned). */
int orc_genasm_pair(const orc_params_t *p, const char *pattern, int plen, const char *text, int tlen, char *ops, orc_result_t *res);

/* Batch in the reference's wire layout: patterns/texts are [n][read_size]
 * byte rows, ops is [n][2*read_size] (may be NULL without backtrace).
 * nthreads<=1 runs on the calling thread. */
int orc_align_batch(const orc_params_t *p, uint32_t n, const int32_t *plen,
                    const int32_t *tlen, const char *patterns, const char *texts,
                    orc_result_t *results, char *ops, int nthreads);

/* Run-length CIGAR of ops[begin,end) exactly like edit_cigar_print
 * (WFA/DPU-WRAM/host/host.c:69-89).  Writes into out (NUL terminated, newline
 * included), returns number of bytes written (excluding NUL). */
int orc_cigar_format(const char *ops, int begin_offset, int end_offset, char *out, int cap);

/* Launcher heuristics (run-wfa-pim-wram.py:57-68, run-nw-pim-wram.py:50-57,
 * run-swg-pim-wram.py:52-62): MAX_SCORE and READ_SIZE from (l, e, costs). */
void orc_launcher_sizes(int algo, int read_length, double error, int mismatch,
                        int gap_o, int gap_e, int gap, int *max_score, int *read_size);

#ifdef __cplusplus
}
#endif
#endif
