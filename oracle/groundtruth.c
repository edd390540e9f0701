/* GROUND TRUTH — TEST INFRASTRUCTURE ONLY (tests/test_ground_truth.py).
 *
 * NOT a restatement of the reference: textbook dynamic programming that knows nothing about search schemes, FM
 * indexes, bit-parallel matrices or the oracle.  It answers two questions about a reported occurrence list that
 * neither the oracle (oracle_search.hpp, the builder's reading of the reference) nor the HIP path can answer about
 * themselves:
 *   soundness     is the distance reported for text[begin, end) an edit / Hamming distance that this text window
 *                 really has to the read?                                          -> gt_edit_distance
 *   completeness  which end positions of the text are within k edits of the read at all (Sellers' semi-global
 *                 alignment, free start in the text)?                              -> gt_semiglobal_ends
 * Alphabet handling follows the matcher's contract (reads.h:43-58, alphabet.h:52-64 of the reference): the caller
 * hands over read characters already upper-cased with everything outside ACGT replaced by 'N'; an 'N' in the read
 * matches nothing, and neither does a text character outside ACGT.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline int eq(uint8_t p, uint8_t t) {
    return p == t && (p == 'A' || p == 'C' || p == 'G' || p == 'T');
}

/* plain O(la * lb) edit distance of pattern a against text window b (unit costs) */
uint32_t gt_edit_distance(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb) {
    uint32_t* prev = (uint32_t*)malloc((size_t)(lb + 1) * sizeof(uint32_t));
    uint32_t* cur = (uint32_t*)malloc((size_t)(lb + 1) * sizeof(uint32_t));
    for (uint32_t j = 0; j <= lb; j++) prev[j] = j;
    for (uint32_t i = 1; i <= la; i++) {
        cur[0] = i;
        for (uint32_t j = 1; j <= lb; j++) {
            uint32_t d = prev[j - 1] + (eq(a[i - 1], b[j - 1]) ? 0u : 1u);
            const uint32_t u = prev[j] + 1, l = cur[j - 1] + 1;
            if (u < d) d = u;
            if (l < d) d = l;
            cur[j] = d;
        }
        uint32_t* t = prev;
        prev = cur;
        cur = t;
    }
    const uint32_t r = prev[lb];
    free(prev);
    free(cur);
    return r;
}

uint32_t gt_hamming_distance(const uint8_t* a, const uint8_t* b, uint32_t l) {
    uint32_t d = 0;
    for (uint32_t i = 0; i < l; i++) d += eq(a[i], b[i]) ? 0u : 1u;
    return d;
}

/* Sellers: best[j] = min over b <= j of the edit distance between the pattern and text[b, j), for every end position
 * j = 0 .. n, capped at k + 1.  Column-wise with Ukkonen's cut-off (only the cells <= k of a column are kept), which
 * is exact for all values <= k.  Also returns, per end position with best[j] <= k, the LARGEST begin position of an
 * optimal alignment ending there (begin[j]; the shortest optimal window), else 0. */
void gt_semiglobal_ends(const uint8_t* text, uint64_t n, const uint8_t* pat, uint32_t m, uint32_t k, uint8_t* best,
                        uint64_t* begin) {
    const uint32_t INF = k + 1;
    uint32_t* col = (uint32_t*)malloc((size_t)(m + 1) * sizeof(uint32_t));
    uint64_t* org = (uint64_t*)malloc((size_t)(m + 1) * sizeof(uint64_t)); /* begin of the best alignment of the cell */
    for (uint32_t i = 0; i <= m; i++) {
        col[i] = i <= k ? i : INF;
        org[i] = 0;
    }
    uint32_t last = k < m ? k : m; /* last row whose value is <= k */
    best[0] = (uint8_t)(m <= k ? m : INF);
    if (begin) begin[0] = 0;
    for (uint64_t j = 1; j <= n; j++) {
        const uint8_t t = text[j - 1];
        uint32_t diag = col[0];
        uint64_t diagOrg = org[0];
        col[0] = 0;
        org[0] = j; /* an alignment may start right here */
        const uint32_t upto = last + 1 < m ? last + 1 : m;
        for (uint32_t i = 1; i <= upto; i++) {
            const uint32_t oldV = col[i];
            const uint64_t oldO = org[i];
            /* diagonal (match / substitution), vertical in this column (pattern char unmatched), horizontal (text char
             * unmatched: from the previous column's same row = oldV) */
            uint32_t d = diag + (eq(pat[i - 1], t) ? 0u : 1u);
            uint64_t o = diagOrg;
            const uint32_t v = col[i - 1] + 1;
            if (v < d || (v == d && org[i - 1] > o)) {
                d = v;
                o = org[i - 1];
            }
            const uint32_t h = (i <= last ? oldV : INF) + 1;
            if (h < d || (h == d && oldO > o && i <= last)) {
                d = h;
                o = oldO;
            }
            if (d > INF) d = INF;
            col[i] = d;
            org[i] = o;
            diag = i <= last ? oldV : INF;
            diagOrg = oldO;
        }
        last = upto;
        while (last > 0 && col[last] > k) last--;
        if (last == m) {
            best[j] = (uint8_t)col[m];
            if (begin) begin[j] = org[m];
        } else {
            best[j] = (uint8_t)INF;
            if (begin) begin[j] = 0;
        }
    }
    free(col);
    free(org);
}
