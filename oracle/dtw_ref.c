/* CPU restatement of dtw.dtw(cost, step_pattern=symmetric1) + the boundary extraction that follows it.
 * TEST INFRASTRUCTURE ONLY: called from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
 *
 * Follows urro_whisper/align.py:818-853.  dtw-python 1.5.3 is not installed here; its published
 * algorithm is restated: candidates D[i-1][j-1]+c, D[i][j-1]+c, D[i-1][j]+c are formed as sums first and
 * scanned in that order with a strict '<' (earliest pattern wins ties), NaN marks unreachable cells,
 * the path is backtracked from (n-1, m-1).  PARITY UNPINNED against dtw-python itself.
 *
 * Build: gcc -O2 -ffp-contract=off -shared -fPIC -o oracle/_build/libdtw_ref.so oracle/dtw_ref.c
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

/* cost: [n][m] f64 row-major.  path_i/path_j: capacity n+m.  boundaries: capacity n+1.
 * Returns path length, or -1 when no path exists, -2 on allocation failure. */
int dtw_symmetric1_ref(const double* cost, int n, int m, int32_t* path_i, int32_t* path_j, int32_t* boundaries,
                       int n_frames)
{
    double* D = (double*)malloc(sizeof(double) * (size_t)n * m);
    uint8_t* S = (uint8_t*)calloc((size_t)n * m, 1);
    if (!D || !S) { free(D); free(S); return -2; }
    for (size_t k = 0; k < (size_t)n * m; ++k) D[k] = NAN;
    D[0] = cost[0];
    for (int j = 0; j < m; ++j) {
        for (int i = 0; i < n; ++i) {
            if (i == 0 && j == 0) continue;
            const double c = cost[(size_t)i * m + j];
            double best = INFINITY;
            int arg = 0;
            if (i > 0 && j > 0) { double v = D[(size_t)(i - 1) * m + j - 1] + c; if (v < best) { best = v; arg = 1; } }
            if (j > 0)          { double v = D[(size_t)i * m + j - 1] + c;       if (v < best) { best = v; arg = 2; } }
            if (i > 0)          { double v = D[(size_t)(i - 1) * m + j] + c;     if (v < best) { best = v; arg = 3; } }
            if (arg) { D[(size_t)i * m + j] = best; S[(size_t)i * m + j] = (uint8_t)arg; }
        }
    }
    if (isnan(D[(size_t)n * m - 1])) { free(D); free(S); return -1; }
    /* backtrack into the tail of the buffers, then shift to the front */
    int len = 0, i = n - 1, j = m - 1;
    const int cap = n + m;
    path_i[cap - 1] = i; path_j[cap - 1] = j; len = 1;
    while (!(i == 0 && j == 0)) {
        const int s = S[(size_t)i * m + j];
        if (s == 1) { --i; --j; } else if (s == 2) { --j; } else { --i; }
        ++len;
        path_i[cap - len] = i; path_j[cap - len] = j;
    }
    for (int k = 0; k < len; ++k) { path_i[k] = path_i[cap - len + k]; path_j[k] = path_j[cap - len + k]; }
    /* align.py:835-853: frame at each jump of the token index, 0 in front, pad to n+1 with the last value,
     * clamp the final entry to n_frames */
    int nb = 0;
    boundaries[nb++] = 0;
    for (int k = 1; k < len; ++k)
        if (path_i[k] > path_i[k - 1]) boundaries[nb++] = path_j[k];
    if (nb != n + 1) {
        if (nb < n + 1) { const int last = boundaries[nb - 1]; while (nb < n + 1) boundaries[nb++] = last; }
        else nb = n + 1;
        if (boundaries[nb - 1] > n_frames) boundaries[nb - 1] = n_frames;
    }
    free(D); free(S);
    return len;
}
