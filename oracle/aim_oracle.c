/*
 * aim_oracle.c -- CPU restatement of AIM's NW / SWG / WFA / WFA-adaptive
 * per-pair kernels, quirk for quirk.  TEST INFRASTRUCTURE ONLY (see the
 * header).  Plain C11, no dependencies.  Every function cites the reference
 * file:line it follows (paths relative to the safaad/aim tree).
 *
 * Parity status: see aim_oracle.h ("parity unpinned" outside the recorded
 * sample-l100-e1-40K digests).
 */
#define _GNU_SOURCE
#include "aim_oracle.h"

#include <limits.h>
#include <math.h>
#include <pthread.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OMIN(a, b) (((a) <= (b)) ? (a) : (b))
#define OMAX(a, b) (((a) >= (b)) ? (a) : (b))

/* ------------------------------------------------------------------------- */
/* scratch arena reused across pairs by one thread                            */
/* ------------------------------------------------------------------------- */
typedef struct {
    void *buf;
    size_t cap;
} orc_scratch_t;

static void *scratch_need(orc_scratch_t *s, size_t bytes)
{
    if (bytes > s->cap) {
        free(s->buf);
        size_t cap = bytes + bytes / 4 + 4096;
        s->buf = malloc(cap);
        s->cap = s->buf ? cap : 0;
    }
    return s->buf;
}

/* edit_cigar_allocate: WFA/DPU-WRAM/dpu/wfa.c:57-67 (same in nw.c:54-64, swg.c:31-41;
 * the initial score differs per file but is always overwritten before output). */
static void cigar_init(orc_result_t *r, int plen, int tlen)
{
    r->max_operations = plen + tlen;
    r->begin_offset = r->max_operations - 1;
    r->end_offset = r->max_operations;
    r->score = INT32_MIN;
    r->status = ORC_OK;
}

/* ========================================================================= */
/* WFA / WFA-adaptive                                                         */
/* ========================================================================= */
typedef int16_t awf_t;                 /* AFFINE_WAVEFRONT_W16, common.h:92-100 */
#define AWF_NULL (INT16_MIN / 2)       /* AFFINE_WAVEFRONT_OFFSET_NULL = -16384 */

/* wfa_component, WFA/DPU-WRAM/common/common.h:126-138.  m/i/d are biased so
 * that m[k] is valid for lo_base <= k <= hi_base. */
typedef struct {
    int present;       /* wavefronts[s] != NULL */
    int klo, khi;
    int lo_base, hi_base;
    awf_t *m, *i, *d;  /* i / d == NULL when not allocated (kernel mask, wfa.c:152-174) */
    bool m_null, i_null, d_null;
} wf_comp_t;

typedef struct {
    wf_comp_t *comp;   /* [max_score + 2] */
    awf_t *pool;       /* bump arena for offsets */
    size_t pool_cap, pool_used;
} wf_state_t;

/* allocate_new_score: wfa.c:143-183 */
static int wf_new_score(wf_state_t *st, int score, int lo, int hi, int kernel)
{
    wf_comp_t *c = &st->comp[score];
    int len = hi - lo + 1;
    int narr = 1 + ((kernel & 1) ? 1 : 0) + ((kernel & 2) ? 1 : 0);
    if (st->pool_used + (size_t)len * narr > st->pool_cap) return ORC_ERR_NOMEM;
    c->present = 1;
    c->m = st->pool + st->pool_used - lo;
    st->pool_used += len;
    if (kernel == 3 || kernel == 1) {
        c->d = st->pool + st->pool_used - lo;
        st->pool_used += len;
        c->d_null = false;
    } else {
        c->d = NULL;
        c->d_null = true;
    }
    if (kernel == 3 || kernel == 2) {
        c->i = st->pool + st->pool_used - lo;
        st->pool_used += len;
        c->i_null = false;
    } else {
        c->i = NULL;
        c->i_null = true;
    }
    c->m_null = false;
    c->klo = lo;
    c->khi = hi;
    c->lo_base = lo;
    c->hi_base = hi;
    return ORC_OK;
}

/* affine_wfa_extend: wfa.c:186-208 */
static void wf_extend(wf_comp_t *c, const char *pattern, const char *text, int plen, int tlen)
{
    if (!c->present || c->m_null) return;
    for (int k = c->klo; k <= c->khi; ++k) {
        int moffset = c->m[k];
        if (moffset < 0) continue;
        int v = moffset - k;
        int h = moffset;
        int count = 0;
        while ((v < plen && h < tlen && v >= 0 && h >= 0) && pattern[v++] == text[h++]) ++count;
        c->m[k] = (awf_t)(c->m[k] + count);
    }
}

/* affine_wfa_reduce_wvs (WFA-adaptive): wfa.c:69-140 */
static void wf_reduce(wf_comp_t *c, int plen, int tlen)
{
    const int min_wavefront_length = 10;
    const int max_distance_threshold = 50;
    int alignment_k = tlen - plen;
    if (!c->present || c->m_null) return;
    if ((c->khi - c->klo + 1) < min_wavefront_length) return;

    int min_distance = OMAX(plen, tlen);
    int klo = c->klo, khi = c->khi;
    for (int k = klo; k <= khi; ++k) {
        awf_t offset = c->m[k];
        int v = offset - k, h = offset;
        int left_v = plen - v, left_h = tlen - h;
        int distance = OMAX(left_v, left_h);
        min_distance = OMIN(distance, min_distance);
    }
    /* reduce from bottom */
    int top_limit = OMIN(alignment_k - 1, khi);
    for (int k = c->klo; k < top_limit; ++k) {
        awf_t offset = c->m[k];
        int v = offset - k, h = offset;
        int left_v = plen - v, left_h = tlen - h;
        int distance = OMAX(left_v, left_h);
        if ((distance - min_distance) <= max_distance_threshold) break;
        c->klo = c->klo + 1;
    }
    /* reduce from top */
    int bottom_limit = OMAX(alignment_k + 1, c->klo);
    for (int k = khi; k > bottom_limit; --k) {
        awf_t offset = c->m[k];
        int v = offset - k, h = offset;
        int left_v = plen - v, left_h = tlen - h;
        int distance = OMAX(left_v, left_h);
        if (distance - min_distance <= max_distance_threshold) break;
        c->khi = c->khi - 1;
    }
    if (c->klo > c->khi) {
        c->m_null = true;
        c->i_null = true;
        c->d_null = true;
        c->khi = khi;
        c->klo = klo;
    }
}

/* affine_wfa_end_reached: wfa.c:210-230 */
static bool wf_end_reached(const wf_comp_t *c, int plen, int tlen)
{
    if (!c->present || c->m_null) return false;
    int alignment_k = tlen - plen;
    int alignment_offset = tlen;
    if (c->klo <= alignment_k && c->khi >= alignment_k) {
        int offset = c->m[alignment_k];
        if (offset >= alignment_offset) return true;
    }
    return false;
}

/* AFFINE_WAVEFRONT_COND_FETCH: common.h:121-124 */
#define COND_FETCH(null_, lo_, hi_, idx_, val_) \
    ((!(null_) && (lo_) <= (idx_) && (idx_) <= (hi_)) ? (val_) : AWF_NULL)

/* affine_wfa_compute_next + affine_wfa_compute_offsets: wfa.c:231-340 */
static int wf_compute_next(wf_state_t *st, const orc_params_t *p, int score)
{
    wf_comp_t *wfs = st->comp;
    int mismatch_score = score - p->mismatch;
    int o_score = score - p->gap_o - p->gap_e;
    int e_score = score - p->gap_e;

    /* wfa.c:278-284 */
    bool m_sub_null = (mismatch_score < 0) || !wfs[mismatch_score].present || wfs[mismatch_score].m_null;
    bool m_o_null = (o_score < 0) || !wfs[o_score].present || wfs[o_score].m_null;
    bool i_e_null = (e_score < 0) || !wfs[e_score].present || wfs[e_score].i == NULL || wfs[e_score].i_null;
    bool d_e_null = (e_score < 0) || !wfs[e_score].present || wfs[e_score].d == NULL || wfs[e_score].d_null;
    bool i_out_null = m_o_null && i_e_null;
    bool d_out_null = m_o_null && d_e_null;

    if (m_sub_null && (i_out_null && d_out_null)) { /* wfa.c:287-291 */
        wfs[score].present = 0;
        return ORC_OK;
    }
    int m_sub_lo = 1, m_sub_hi = -1, m_o_lo = 1, m_o_hi = -1, e_lo = 1, e_hi = -1; /* wfa.c:294-328 */
    const wf_comp_t *w_sub = NULL, *w_o = NULL, *w_e = NULL;
    if (!m_sub_null) {
        w_sub = &wfs[mismatch_score];
        m_sub_lo = w_sub->klo;
        m_sub_hi = w_sub->khi;
    }
    if (!m_o_null) {
        w_o = &wfs[o_score];
        m_o_lo = w_o->klo;
        m_o_hi = w_o->khi;
    }
    if (!(i_e_null && d_e_null)) {
        w_e = &wfs[e_score];
        e_lo = w_e->klo;
        e_hi = w_e->khi;
    }
    int lo = OMIN(m_sub_lo, m_o_lo);
    lo = OMIN(lo, e_lo) - 1;
    int hi = OMAX(m_sub_hi, m_o_hi);
    hi = OMAX(hi, e_hi) + 1;
    int kernel = ((!i_out_null) << 1) | (!d_out_null); /* wfa.c:335 */

    int rc = wf_new_score(st, score, lo, hi, kernel);
    if (rc) return rc;
    wf_comp_t *out = &wfs[score];

    for (int k = lo; k <= hi; ++k) { /* wfa.c:234-265 */
        awf_t ins = -10;
        if (!m_o_null || !i_e_null) {
            awf_t ins_g = COND_FETCH(m_o_null, m_o_lo, m_o_hi, k - 1, w_o->m[k - 1]);
            awf_t ins_i = COND_FETCH(i_e_null, e_lo, e_hi, k - 1, w_e->i[k - 1]);
            if (ins_g == AWF_NULL && ins_i == AWF_NULL)
                ins = AWF_NULL;
            else
                ins = (awf_t)(OMAX(ins_g, ins_i) + 1);
            out->i[k] = ins;
        }
        awf_t del = -10;
        if (!m_o_null || !d_e_null) {
            awf_t del_g = COND_FETCH(m_o_null, m_o_lo, m_o_hi, k + 1, w_o->m[k + 1]);
            awf_t del_d = COND_FETCH(d_e_null, e_lo, e_hi, k + 1, w_e->d[k + 1]);
            del = OMAX(del_g, del_d);
            out->d[k] = del;
        }
        awf_t sub = -10;
        if (!m_sub_null) sub = (awf_t)COND_FETCH(m_sub_null, m_sub_lo, m_sub_hi, k, w_sub->m[k] + 1);
        awf_t nw = OMAX(sub, ins);
        out->m[k] = OMAX(del, nw);
    }
    return ORC_OK;
}

/* Backtrace source fetchers: wfa_backtracing.c:73-172.  Note the asymmetric
 * null tests (del_ext: !d_null, :102; ins_ext: iwavefront != NULL, :142) and
 * that m_null is never consulted. */
static awf_t bt_del_open(const wf_comp_t *wfs, int score, int k)
{
    if (score < 0) return AWF_NULL;
    const wf_comp_t *w = &wfs[score];
    if (w->present && w->klo <= k + 1 && k + 1 <= w->khi) return w->m[k + 1];
    return AWF_NULL;
}
static awf_t bt_del_ext(const wf_comp_t *wfs, int score, int k)
{
    if (score < 0) return AWF_NULL;
    const wf_comp_t *w = &wfs[score];
    if (w->present && !w->d_null && w->klo <= k + 1 && k + 1 <= w->khi) return w->d[k + 1];
    return AWF_NULL;
}
static awf_t bt_ins_open(const wf_comp_t *wfs, int score, int k)
{
    if (score < 0) return AWF_NULL;
    const wf_comp_t *w = &wfs[score];
    if (w->present && w->klo <= k - 1 && k - 1 <= w->khi) return (awf_t)(w->m[k - 1] + 1);
    return AWF_NULL;
}
static awf_t bt_ins_ext(const wf_comp_t *wfs, int score, int k)
{
    if (score < 0) return AWF_NULL;
    const wf_comp_t *w = &wfs[score];
    if (w->present && w->i != NULL && w->klo <= k - 1 && k - 1 <= w->khi) return (awf_t)(w->i[k - 1] + 1);
    return AWF_NULL;
}
static awf_t bt_misms(const wf_comp_t *wfs, int score, int k)
{
    if (score < 0) return AWF_NULL;
    const wf_comp_t *w = &wfs[score];
    if (w->present && w->klo <= k && k <= w->khi) return (awf_t)(w->m[k] + 1);
    return AWF_NULL;
}

/* affine_wavefronts_valid_location: wfa_backtracing.c:36-47 */
static bool bt_valid_location(int k, awf_t offset, int plen, int tlen)
{
    int v = offset - k, h = offset;
    return (v > 0 && v <= plen && h > 0 && h <= tlen);
}

/* affine_wavefronts_backtrace: wfa_backtracing.c:210-351 */
static int wf_backtrace(const wf_comp_t *wfs, const orc_params_t *p, orc_result_t *cig, char *ops,
                        int plen, int tlen, int alignment_score)
{
    enum { BT_M = 0, BT_I = 1, BT_D = 2 };
    int alignment_k = tlen - plen;
    int score = alignment_score;
    int k = alignment_k;
    awf_t offset = wfs[alignment_score].m[k];
    bool valid_location = bt_valid_location(k, offset, plen, tlen);
    int bt = BT_M;
    int v = offset - k, h = offset;
    while (v > 0 && h > 0 && score > 0) {
        if (!valid_location) {
            valid_location = bt_valid_location(k, offset, plen, tlen);
            if (valid_location) { /* add_trailing_gap: wfa_backtracing.c:48-69 */
                int sentinel = cig->begin_offset;
                if (k < alignment_k) {
                    for (int i = k; i < alignment_k; ++i) ops[sentinel--] = 'I';
                } else if (k > alignment_k) {
                    for (int i = alignment_k; i < k; ++i) ops[sentinel--] = 'D';
                }
                cig->begin_offset = sentinel;
            }
        }
        int gap_open_score = score - p->gap_o - p->gap_e;
        int gap_extend_score = score - p->gap_e;
        int mismatch_score = score - p->mismatch;
        awf_t del_ext = (bt == BT_I) ? AWF_NULL : bt_del_ext(wfs, gap_extend_score, k);
        awf_t del_open = (bt == BT_I) ? AWF_NULL : bt_del_open(wfs, gap_open_score, k);
        awf_t ins_ext = (bt == BT_D) ? AWF_NULL : bt_ins_ext(wfs, gap_extend_score, k);
        awf_t ins_open = (bt == BT_D) ? AWF_NULL : bt_ins_open(wfs, gap_open_score, k);
        awf_t misms = (bt != BT_M) ? AWF_NULL : bt_misms(wfs, mismatch_score, k);
        awf_t max_del = OMAX(del_ext, del_open);
        awf_t max_ins = OMAX(ins_ext, ins_open);
        awf_t max_all = OMAX(misms, OMAX(max_ins, max_del));
        if (bt == BT_M) {
            int num_matches = offset - max_all;
            for (int i = 0; i < num_matches; ++i) ops[(cig->begin_offset)--] = 'M';
            offset = max_all;
            v = offset - k;
            h = offset;
            if (v <= 0 || h <= 0) break;
        }
        if (max_all == del_ext) {
            if (valid_location) ops[(cig->begin_offset)--] = 'D';
            score = gap_extend_score;
            ++k;
            bt = BT_D;
        } else if (max_all == del_open) {
            if (valid_location) ops[(cig->begin_offset)--] = 'D';
            score = gap_open_score;
            ++k;
            bt = BT_M;
        } else if (max_all == ins_ext) {
            if (valid_location) ops[(cig->begin_offset)--] = 'I';
            score = gap_extend_score;
            --k;
            --offset;
            bt = BT_I;
        } else if (max_all == ins_open) {
            if (valid_location) ops[(cig->begin_offset)--] = 'I';
            score = gap_open_score;
            --k;
            --offset;
            bt = BT_M;
        } else if (max_all == misms) {
            if (valid_location) ops[(cig->begin_offset)--] = 'X';
            score = mismatch_score;
            --offset;
        } else {
            return ORC_ERR_WFA_NO_LINK; /* "Backtrace error: No link found during backtrace", exit(1) */
        }
        v = offset - k;
        h = offset;
    }
    if (score == 0) {
        int n = offset;
        for (int i = 0; i < n; ++i) ops[(cig->begin_offset)--] = 'M';
    } else {
        while (v > 0) {
            ops[(cig->begin_offset)--] = 'D';
            --v;
        }
        while (h > 0) {
            ops[(cig->begin_offset)--] = 'I';
            --h;
        }
    }
    ++(cig->begin_offset);
    return ORC_OK;
}

/* affine_wfa_compute + the per-pair part of main(): wfa.c:342-379, 460-495 */
static int wfa_pair(const orc_params_t *p, orc_scratch_t *scr, const char *pattern, int plen,
                    const char *text, int tlen, char *ops, orc_result_t *res)
{
    const int ms = p->max_score;
    size_t ncomp = (size_t)ms + 2;
    /* worst case: score s allocates 3 arrays of width <= 2s+1 */
    size_t pool_cap = 3 * (size_t)(ms + 2) * (size_t)(ms + 2) + 16;
    size_t bytes = ncomp * sizeof(wf_comp_t) + pool_cap * sizeof(awf_t);
    char *mem = scratch_need(scr, bytes);
    if (!mem) return res->status = ORC_ERR_NOMEM;
    wf_state_t st;
    st.comp = (wf_comp_t *)mem;
    st.pool = (awf_t *)(mem + ncomp * sizeof(wf_comp_t));
    st.pool_cap = pool_cap;
    st.pool_used = 0;
    memset(st.comp, 0, ncomp * sizeof(wf_comp_t));

    cigar_init(res, plen, tlen);
    if (p->backtrace) memset(ops, 'M', (size_t)2 * p->read_size); /* wfa.c:463-465 */

    wf_new_score(&st, 0, 0, 0, 0); /* wfa.c:347-348 */
    st.comp[0].m[0] = 0;
    int score = 0;
    for (;;) {
        wf_extend(&st.comp[score], pattern, text, plen, tlen);
        if (p->reduce) wf_reduce(&st.comp[score], plen, tlen);
        if (wf_end_reached(&st.comp[score], plen, tlen)) {
            if (p->backtrace) {
                int rc = wf_backtrace(st.comp, p, res, ops, plen, tlen, score);
                if (rc) res->status = rc;
            }
            res->score = score;
            return res->status;
        }
        ++score;
        if (score > ms) {
            /* wfa.c:368-376.  With BACKTRACE the WRAM variant indexes
             * wavefronts[MAX_SCORE+1] (out of bounds, UB); the MRAM variant
             * (WFA/DPU-MRAM/dpu/wfa.c:400-404) returns without backtrace, which
             * is the defined behaviour restated here: CIGAR = ops[max-1,max). */
            res->score = score;
            return res->status;
        }
        int rc = wf_compute_next(&st, p, score);
        if (rc) return res->status = rc;
    }
}

/* ========================================================================= */
/* NW (linear gap, int16 cells, flat table with stride tlen+1)               */
/* ========================================================================= */
/* nw_compute + nw_traceback: NW/DPU-WRAM/dpu/nw.c:67-153.  The table is kept
 * flat and indexed num_cols*h + v exactly like the reference so that the
 * row aliasing for plen > tlen (v runs past num_cols) is reproduced. */
static int nw_pair(const orc_params_t *p, orc_scratch_t *scr, const char *pattern, int plen,
                   const char *text, int tlen, char *ops, orc_result_t *res)
{
    typedef int16_t cell_t; /* NW_W16, NW/DPU-WRAM/common/common.h:87-97 */
    const int GAP_D = p->gap_d, GAP_I = p->gap_i, MISMATCH = p->mismatch;
    int num_rows = plen + 1;
    int num_cols = tlen + 1;
    size_t ncell = (size_t)num_cols * (size_t)(tlen + 1) + (size_t)plen + 2;
    cell_t *dp = scratch_need(scr, ncell * sizeof(cell_t));
    if (!dp) return res->status = ORC_ERR_NOMEM;
    cigar_init(res, plen, tlen);
    (void)num_rows;

    int cell = 0;
    dp[0] = (cell_t)cell;
    for (int v = 1; v <= plen; ++v) {
        cell = cell + GAP_D;
        dp[v] = (cell_t)cell;
    }
    cell = 0;
    for (int h = 1; h <= tlen; ++h) {
        cell = cell + GAP_I;
        dp[(size_t)num_cols * h] = (cell_t)cell;
    }
    cell_t score = 0;
    for (int h = 1; h <= tlen; ++h) {
        for (int v = 1; v <= plen; ++v) {
            cell_t del = (cell_t)(dp[(size_t)num_cols * h + v - 1] + GAP_D);
            cell_t ins = (cell_t)(dp[(size_t)num_cols * (h - 1) + v] + GAP_I);
            cell_t m_match = (cell_t)(dp[(size_t)num_cols * (h - 1) + v - 1] +
                                      ((pattern[v - 1] == text[h - 1]) ? 0 : MISMATCH));
            score = dp[(size_t)num_cols * h + v] = (cell_t)OMIN(m_match, OMIN(ins, del));
        }
    }
    res->score = (int)score;
    if (p->backtrace) { /* nw_traceback: nw.c:67-107 */
        int op_sentinel = res->end_offset - 1;
        int h = num_cols - 1;
        int v = num_rows - 1;
        while (h > 0 && v > 0) {
            size_t at = (size_t)num_cols * h + v;
            if (dp[at] == dp[at - 1] + GAP_D) {
                ops[op_sentinel--] = 'D';
                --v;
            } else if (dp[at] == dp[at - num_cols] + GAP_I) {
                ops[op_sentinel--] = 'I';
                --h;
            } else {
                ops[op_sentinel--] = (dp[at] == dp[at - num_cols - 1] + MISMATCH) ? 'X' : 'M';
                --h;
                --v;
            }
        }
        while (h > 0) {
            ops[op_sentinel--] = 'I';
            --h;
        }
        while (v > 0) {
            ops[op_sentinel--] = 'D';
            --v;
        }
        res->begin_offset = op_sentinel + 1;
    }
    return res->status;
}

/* ========================================================================= */
/* SWG (global Gotoh, int8 or int16 cells, MAX_SCORE as +infinity)           */
/* ========================================================================= */
/* swg_compute + swg_traceback: SWG/DPU-WRAM/dpu/swg.c:45-171.  Instantiated
 * for both cell widths (SWG/DPU-WRAM/common/common.h:71-86: int8 when
 * MAX_SCORE < 127, else int16; SWG/DPU-MRAM is always int16). */
#define SWG_IMPL(NAME, CELL_T)                                                                    \
    static int NAME(const orc_params_t *p, orc_scratch_t *scr, const char *pattern, int plen,     \
                    const char *text, int tlen, char *ops, orc_result_t *res)                     \
    {                                                                                             \
        typedef struct { CELL_T M, I, D; } dp_cell_t;                                             \
        const int GAP_O = p->gap_o, GAP_E = p->gap_e, MATCH = p->match, MISMATCH = p->mismatch;   \
        const int MAX_SCORE = p->max_score;                                                       \
        int num_rows = plen + 1;                                                                  \
        int num_cols = tlen + 1;                                                                  \
        size_t ncell = (size_t)num_cols * (size_t)(tlen + 1) + (size_t)plen + 2;                  \
        dp_cell_t *dp = scratch_need(scr, ncell * sizeof(dp_cell_t));                             \
        if (!dp) return res->status = ORC_ERR_NOMEM;                                              \
        cigar_init(res, plen, tlen);                                                              \
        if (p->backtrace) memset(ops, 'M', (size_t)2 * p->read_size); /* swg.c:259-262 */        \
        dp[0].D = (CELL_T)MAX_SCORE;                                                              \
        dp[0].I = (CELL_T)MAX_SCORE;                                                              \
        dp[0].M = 0;                                                                              \
        for (int v = 1; v <= plen; ++v) {                                                         \
            dp[v].D = (CELL_T)(GAP_O + v * GAP_E);                                                \
            dp[v].I = (CELL_T)MAX_SCORE;                                                          \
            dp[v].M = dp[v].D;                                                                    \
        }                                                                                         \
        for (int h = 1; h <= tlen; ++h) {                                                         \
            dp[(size_t)num_cols * h].D = (CELL_T)MAX_SCORE;                                       \
            dp[(size_t)num_cols * h].I = (CELL_T)(GAP_O + h * GAP_E);                             \
            dp[(size_t)num_cols * h].M = dp[(size_t)num_cols * h].I;                              \
        }                                                                                         \
        int score = 0;                                                                            \
        for (int h = 1; h <= tlen; ++h) {                                                         \
            for (int v = 1; v <= plen; ++v) {                                                     \
                size_t at = (size_t)num_cols * h + v;                                             \
                CELL_T del_new = (CELL_T)(dp[at - 1].M + GAP_O + GAP_E);                          \
                CELL_T del_ext = (CELL_T)(dp[at - 1].D + GAP_E);                                  \
                CELL_T del = OMIN(del_new, del_ext);                                              \
                dp[at].D = del;                                                                   \
                CELL_T ins_new = (CELL_T)(dp[at - num_cols].M + GAP_O + GAP_E);                   \
                CELL_T ins_ext = (CELL_T)(dp[at - num_cols].I + GAP_E);                           \
                CELL_T ins = OMIN(ins_new, ins_ext);                                              \
                dp[at].I = ins;                                                                   \
                CELL_T m_match = (CELL_T)(dp[at - num_cols - 1].M +                               \
                                          ((pattern[v - 1] == text[h - 1]) ? MATCH : MISMATCH));  \
                score = dp[at].M = (CELL_T)OMIN(m_match, OMIN(ins, del));                         \
            }                                                                                     \
        }                                                                                         \
        res->score = score;                                                                       \
        if (p->backtrace) { /* swg_traceback: swg.c:45-119 */                                     \
            enum { L_M, L_I, L_D };                                                               \
            int op_sentinel = res->end_offset - 1;                                                \
            int h = num_cols - 1, v = num_rows - 1;                                               \
            int layer = L_M;                                                                      \
            while (h > 0 && v > 0) {                                                              \
                size_t at = (size_t)num_cols * h + v;                                             \
                if (layer == L_D) {                                                               \
                    ops[op_sentinel--] = 'D';                                                     \
                    if (dp[at].D == dp[at - 1].M + GAP_O + GAP_E) layer = L_M;                    \
                    --v;                                                                          \
                } else if (layer == L_I) {                                                        \
                    ops[op_sentinel--] = 'I';                                                     \
                    if (dp[at].I == dp[at - num_cols].M + GAP_O + GAP_E) layer = L_M;             \
                    --h;                                                                          \
                } else {                                                                          \
                    if (dp[at].M == dp[at].D) {                                                   \
                        layer = L_D;                                                              \
                    } else if (dp[at].M == dp[at].I) {                                            \
                        layer = L_I;                                                              \
                    } else if (dp[at].M == dp[at - num_cols - 1].M + MATCH) {                     \
                        ops[op_sentinel--] = 'M';                                                 \
                        --h;                                                                      \
                        --v;                                                                      \
                    } else if (dp[at].M == dp[at - num_cols - 1].M + MISMATCH) {                  \
                        ops[op_sentinel--] = 'X';                                                 \
                        --h;                                                                      \
                        --v;                                                                      \
                    } else {                                                                      \
                        /* "SWG backtrace. No backtrace operation found", exit(1) */              \
                        res->begin_offset = op_sentinel + 1;                                      \
                        return res->status = ORC_ERR_SWG_NO_OP;                                   \
                    }                                                                             \
                }                                                                                 \
            }                                                                                     \
            while (h > 0) {                                                                       \
                ops[op_sentinel--] = 'I';                                                         \
                --h;                                                                              \
            }                                                                                     \
            while (v > 0) {                                                                       \
                ops[op_sentinel--] = 'D';                                                         \
                --v;                                                                              \
            }                                                                                     \
            res->begin_offset = op_sentinel + 1;                                                  \
        }                                                                                         \
        return res->status;                                                                       \
    }

SWG_IMPL(swg_pair_w8, int8_t)
SWG_IMPL(swg_pair_w16, int16_t)

/* ========================================================================= */
/* dispatch                                                                  */
/* ========================================================================= */
static int align_pair_scr(const orc_params_t *p, orc_scratch_t *scr, const char *pattern, int plen,
                          const char *text, int tlen, char *ops, orc_result_t *res)
{
    switch (p->algo) {
    case ORC_ALGO_WFA:
        return wfa_pair(p, scr, pattern, plen, text, tlen, ops, res);
    case ORC_ALGO_NW:
        return nw_pair(p, scr, pattern, plen, text, tlen, ops, res);
    case ORC_ALGO_SWG: {
        int w = p->swg_cell_bytes ? p->swg_cell_bytes : (p->max_score < 127 ? 1 : 2);
        return (w == 1) ? swg_pair_w8(p, scr, pattern, plen, text, tlen, ops, res)
                        : swg_pair_w16(p, scr, pattern, plen, text, tlen, ops, res);
    }
    case ORC_ALGO_GENASM:
        return orc_genasm_pair(p, pattern, plen, text, tlen, p->backtrace ? ops : NULL, res);
    default:
        return -1;
    }
}

int orc_align_pair(const orc_params_t *p, const char *pattern, int plen, const char *text, int tlen,
                   char *ops, orc_result_t *res)
{
    orc_scratch_t scr = {0, 0};
    res->idx = 0;
    int rc = align_pair_scr(p, &scr, pattern, plen, text, tlen, ops, res);
    free(scr.buf);
    return rc;
}

typedef struct {
    const orc_params_t *p;
    uint32_t begin, end;
    const int32_t *plen, *tlen;
    const char *patterns, *texts;
    orc_result_t *results;
    char *ops;
    int worst;
} batch_job_t;

static void *batch_worker(void *arg)
{
    batch_job_t *j = arg;
    orc_scratch_t scr = {0, 0};
    const size_t rs = (size_t)j->p->read_size;
    for (uint32_t i = j->begin; i < j->end; ++i) {
        char *ops = j->ops ? j->ops + (size_t)i * 2 * rs : NULL;
        int rc = align_pair_scr(j->p, &scr, j->patterns + i * rs, j->plen[i], j->texts + i * rs,
                                j->tlen[i], ops, &j->results[i]);
        j->results[i].idx = i;
        if (rc > j->worst) j->worst = rc;
    }
    free(scr.buf);
    return NULL;
}

int orc_align_batch(const orc_params_t *p, uint32_t n, const int32_t *plen, const int32_t *tlen,
                    const char *patterns, const char *texts, orc_result_t *results, char *ops,
                    int nthreads)
{
    if (p->backtrace && !ops) return -1;
    if (nthreads < 1) nthreads = 1;
    if ((uint32_t)nthreads > n) nthreads = n ? (int)n : 1;
    batch_job_t *jobs = calloc((size_t)nthreads, sizeof(*jobs));
    pthread_t *tids = calloc((size_t)nthreads, sizeof(*tids));
    uint32_t per = (n + (uint32_t)nthreads - 1) / (uint32_t)nthreads;
    int worst = 0;
    for (int t = 0; t < nthreads; ++t) {
        uint32_t b = (uint32_t)t * per, e = b + per;
        if (b > n) b = n;
        if (e > n) e = n;
        jobs[t] = (batch_job_t){p, b, e, plen, tlen, patterns, texts, results, p->backtrace ? ops : NULL, 0};
        if (nthreads == 1)
            batch_worker(&jobs[t]);
        else
            pthread_create(&tids[t], NULL, batch_worker, &jobs[t]);
    }
    for (int t = 0; t < nthreads; ++t) {
        if (nthreads > 1) pthread_join(tids[t], NULL);
        if (jobs[t].worst > worst) worst = jobs[t].worst;
    }
    free(jobs);
    free(tids);
    return worst;
}

/* edit_cigar_print: WFA/DPU-WRAM/host/host.c:69-89 */
int orc_cigar_format(const char *ops, int begin_offset, int end_offset, char *out, int cap)
{
    int n = 0;
    char last_op = ops[begin_offset];
    int last_op_length = 1;
    for (int i = begin_offset + 1; i < end_offset; ++i) {
        if (ops[i] == last_op) {
            ++last_op_length;
        } else {
            n += snprintf(out + n, (size_t)(cap - n), "%d%c", last_op_length, last_op);
            if (n >= cap) return -1;
            last_op = ops[i];
            last_op_length = 1;
        }
    }
    n += snprintf(out + n, (size_t)(cap - n), "%d%c\n", last_op_length, last_op);
    return n >= cap ? -1 : n;
}

/* run-wfa-pim-wram.py:57-68; run-nw-pim-wram.py:50-57 (gap only);
 * run-swg-pim-wram.py:52-62.  Python float arithmetic == C double here. */
void orc_launcher_sizes(int algo, int read_length, double error, int mismatch, int gap_o, int gap_e,
                        int gap, int *max_score, int *read_size)
{
    double nr_of_wrong_bases = (double)read_length * error;
    double a = nr_of_wrong_bases * (double)mismatch;
    double b = (algo == ORC_ALGO_NW) ? nr_of_wrong_bases * (double)gap
                                     : nr_of_wrong_bases * (double)(gap_o + gap_e);
    *max_score = (int)ceil(a > b ? a : b);
    *read_size = (int)ceil((((double)read_length + nr_of_wrong_bases) + 7.0) / 8.0) * 8;
}
