/* TEST INFRASTRUCTURE ONLY (the checker of gnnome_overlap_edit_distance; never imported by gnnome_amd/).
 *
 * What it restates: graph_parser.py:110  `edlib.align(read_src[-ol_length:], read_dst[:ol_length])['editDistance']`.
 * edlib is a third-party dependency that is NOT under /root/reference (requirements.txt:27 / requirements_cpu.txt:27:
 * edlib==1.3.9).  With its defaults (mode="NW", task="distance", k=-1) edlib.align returns the global-alignment edit
 * distance of the two strings under unit costs - the Levenshtein distance, computed there with Myers' / Hyyro's banded
 * bit-vector algorithm (Sosic & Sikic 2017).  The value is defined by the Wagner-Fischer recurrence, which this file
 * evaluates literally:
 *     D[i][0] = i, D[0][j] = j, D[i][j] = min(D[i-1][j] + 1, D[i][j-1] + 1, D[i-1][j-1] + (a[i-1] != b[j-1]))
 * Pinning: tests/golden/g10_gfa.pt holds the similarities the reference's own calculate_similarities produced with this
 * recurrence standing in for the edlib wheel (make_golden_gfa.py); edlib itself cannot be run here - parity of the
 * third-party arithmetic is pinned through its documented result only. */
#include <stdint.h>
#include <stdlib.h>

int gnnome_oracle_edit_distance(const uint8_t* a, int64_t m, const uint8_t* b, int64_t n, int64_t* out) {
    int32_t* prev = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    int32_t* cur = (int32_t*)malloc((size_t)(n + 1) * sizeof(int32_t));
    if (!prev || !cur) {
        free(prev);
        free(cur);
        return -1;
    }
    for (int64_t j = 0; j <= n; ++j) prev[j] = (int32_t)j;
    for (int64_t i = 1; i <= m; ++i) {
        cur[0] = (int32_t)i;
        const uint8_t ca = a[i - 1];
        for (int64_t j = 1; j <= n; ++j) {
            int32_t best = prev[j] + 1;
            const int32_t left = cur[j - 1] + 1, diag = prev[j - 1] + (ca != b[j - 1]);
            if (left < best) best = left;
            if (diag < best) best = diag;
            cur[j] = best;
        }
        int32_t* t = prev;
        prev = cur;
        cur = t;
    }
    *out = prev[n];
    free(prev);
    free(cur);
    return 0;
}
