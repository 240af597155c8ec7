/*
 * genasm_oracle.c -- CPU restatement of GenASM (bit-vector approximate string matching with windowed
 * traceback) for BASELINE config 5 ("GenASM bitvector edit-distance l=100000 e=10% long reads").
 *
 * TEST INFRASTRUCTURE ONLY (see aim_oracle.h).
 *
 * PARITY UNPINNED.  AIM's GenASM lives in an un-vendored git submodule: /root/reference/.gitmodules:1-3 names
 * https://github.com/safaad/aim-genasm with NO pinned commit, the directory /root/reference/aim-genasm is empty,
 * there are no call sites, tests or golden vectors for it anywhere in the reference tree.  What is restated here
 * is therefore the PUBLISHED algorithm, not AIM's code:
 *   Senol Cali et al., "GenASM: A High-Performance, Low-Power Approximate String Matching Acceleration
 *   Framework for Genome Sequence Analysis", MICRO 2020 -- Algorithm 1 (GenASM-DC: Bitap with the four
 *   bit-vectors match / substitution / deletion / insertion per error level) and Section 6 (GenASM-TB and the
 *   divide-and-conquer windows: window size W = 64, overlap O = 24, i.e. W - O = 40 characters of a window's
 *   traceback are committed before the next window starts where the committed part ended).
 * Every choice the paper leaves open is fixed below and marked [spec]; the HIP kernel (aim_amd/csrc/genasm_wave.hpp)
 * implements exactly this text, and the tests check both against each other bit for bit, the CIGARs against their
 * defining properties, and the distance against the exact edit distance on small inputs (it is an upper bound, and
 * equal to it whenever the optimal path stays inside the windows).
 *
 * Window problem.  Window pattern p[0..m), window text t[0..n), m, n <= 64.  R_a[d] (a = n .. 0, d = 0 .. 63) is a
 * 64-bit vector; bit (m-1-b) of R_a[d] is 0  <=>  p[b..m) matches a PREFIX of t[a..n) with at most d edits (the
 * Bitap status vector of the reversed strings, so that the traceback runs forward from (a, b) = (0, 0)):
 *     R_n[d]  = ~0 << d                                            (only pattern-only edits are left)
 *     PM[c]   : bit j = 0 <=> p[m-1-j] == c  (bits >= m are 1)
 *     R_a[0]  = (R_{a+1}[0] << 1) | PM[t[a]]
 *     R_a[d]  = ((R_{a+1}[d] << 1) | PM[t[a]])                     match
 *             & (R_{a+1}[d-1] << 1)                                substitution
 *             &  R_{a+1}[d-1]                                      text-only edit   ('I': consumes t[a])
 *             & (R_a[d-1] << 1)                                    pattern-only edit ('D': consumes p[b])
 * d0 = the smallest d <= 63 with bit (m-1) of R_0[d] clear.
 * Traceback from (a, b, d) = (0, 0, d0); "ok(a, b, d)" = (b == m) or bit (m-1-b) of R_a[d] is clear.  Per step, first
 * rule that applies [spec: this priority order]:
 *     b == m                                   -> the window's pattern is consumed, stop
 *     a == n                                   -> 'D' (pattern-only), b+1, d-1
 *     p[b] == t[a] and ok(a+1, b+1, d)         -> 'M'
 *     d > 0 and ok(a+1, b+1, d-1)              -> 'X'
 *     d > 0 and ok(a,   b+1, d-1)              -> 'D'
 *     d > 0 and ok(a+1, b,   d-1)              -> 'I'
 * A window that is not the last one of the pair stops as soon as it has consumed W - O = 40 pattern characters or 40
 * text characters [spec]; the next window starts at the characters after the committed ones.  The last window (it covers
 * the rest of BOTH sequences) runs to the end of its pattern.  If no d <= 63 exists (64 pattern characters without a
 * single usable match) the window is committed as min(m, n, 40) diagonal steps, 'M' where the characters are equal and 'X'
 * where they differ [spec].  When one sequence is exhausted the rest of the other is emitted as 'D' (pattern) / 'I' (text).
 * Letters as in AIM's CIGARs: M match, X mismatch, D consumes a pattern character, I consumes a text character.
 * score = number of X + I + D operations (the edit distance of the reported alignment).  ops are written FORWARD:
 * begin_offset = 0, end_offset = number of operations (<= plen + tlen).
 */
#include "aim_oracle.h"

#include <string.h>

#define GA_W 64
#define GA_COMMIT 40

static inline int ga_ok(uint64_t r, int m, int b) { return b >= m || !((r >> (m - 1 - b)) & 1ull); }

int orc_genasm_pair(const orc_params_t *p, const char *pattern, int plen, const char *text, int tlen, char *ops, orc_result_t *res)
{
    static const uint64_t ONES = ~0ull;
    uint64_t R[GA_W + 1][GA_W];          /* [a][d] */
    int pi = 0, ti = 0, nops = 0, dist = 0;
    const int cap = 2 * p->read_size;
    res->max_operations = plen + tlen;
    res->status = ORC_OK;
#define GA_EMIT(ch) do { if (ops && nops < cap) ops[nops] = (ch); ++nops; } while (0)
    while (pi < plen && ti < tlen) {
        const int m = plen - pi < GA_W ? plen - pi : GA_W, n = tlen - ti < GA_W ? tlen - ti : GA_W;
        const unsigned char *wp = (const unsigned char *)pattern + pi, *wt = (const unsigned char *)text + ti;
        const int last = (m == plen - pi) && (n == tlen - ti);
        for (int d = 0; d < GA_W; ++d) R[n][d] = ONES << d;
        for (int a = n - 1; a >= 0; --a) {
            uint64_t pm = ONES;
            for (int j = 0; j < m; ++j)
                if (wp[m - 1 - j] == wt[a]) pm &= ~(1ull << j);
            R[a][0] = (R[a + 1][0] << 1) | pm;
            for (int d = 1; d < GA_W; ++d)
                R[a][d] = ((R[a + 1][d] << 1) | pm) & (R[a + 1][d - 1] << 1) & R[a + 1][d - 1] & (R[a][d - 1] << 1);
        }
        int d = -1;
        for (int k = 0; k < GA_W; ++k)
            if (!((R[0][k] >> (m - 1)) & 1ull)) { d = k; break; }
        int a = 0, b = 0;
        if (d < 0) {   /* [spec] no alignment of this window within 63 edits */
            int steps = m < n ? m : n;
            if (steps > GA_COMMIT) steps = GA_COMMIT;
            for (; a < steps; ++a, ++b) {
                const int eq = wp[b] == wt[a];
                GA_EMIT(eq ? 'M' : 'X');
                dist += !eq;
            }
        } else {
            for (;;) {
                if (b == m) break;
                if (!last && (a >= GA_COMMIT || b >= GA_COMMIT)) break;
                if (a == n) { GA_EMIT('D'); ++b; --d; ++dist; continue; }
                if (wp[b] == wt[a] && ga_ok(R[a + 1][d], m, b + 1)) { GA_EMIT('M'); ++a; ++b; continue; }
                if (d > 0 && ga_ok(R[a + 1][d - 1], m, b + 1)) { GA_EMIT('X'); ++a; ++b; --d; ++dist; continue; }
                if (d > 0 && ga_ok(R[a][d - 1], m, b + 1)) { GA_EMIT('D'); ++b; --d; ++dist; continue; }
                if (d > 0 && ga_ok(R[a + 1][d - 1], m, b)) { GA_EMIT('I'); ++a; --d; ++dist; continue; }
                res->status = ORC_ERR_WFA_NO_LINK;   /* cannot happen: the recurrence guarantees one rule applies */
                break;
            }
            if (res->status != ORC_OK) break;
        }
        pi += b;
        ti += a;
    }
    for (; pi < plen; ++pi) { GA_EMIT('D'); ++dist; }
    for (; ti < tlen; ++ti) { GA_EMIT('I'); ++dist; }
#undef GA_EMIT
    res->begin_offset = 0;
    res->end_offset = nops;
    res->score = dist;
    return res->status;
}
