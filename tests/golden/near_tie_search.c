/* near_tie_search.c -- how close can two DIFFERENT hypergeometric weights on opposite sides of
 * the mode get?  Spec S3 treats weights within a relative TIE = 1e-10 of the observed one as
 * ties (SciPy 1.15: 1e-14); a pair of support points whose weights differ by less than that
 * without being equal would be classified differently by the two rules.  This tool walks ALL
 * margins (n1, n) of 2x2 tables over N isolates -- up to the row/column symmetries, n1 <= N/2
 * and n <= N/2 -- and, for every support point left of the mode, the one or two points right
 * of it with the closest weight (long double log-weights from the exact ratio recurrence), and
 * prints every pair with 1e-15 < |log w(y) - log w(x)| < BAND (the exact ties of symmetric
 * margins are skipped by construction; any other exact tie is weeded out by the exact check).
 *   gcc -O2 -fopenmp -o near_tie_search near_tie_search.c -lm ; ./near_tie_search N [band]
 * output lines: n1 n2 n x y logdiff.   tests/golden/make_near_ties.py drives it, verifies the
 * closest pairs with exact rational arithmetic and writes tests/golden/near_ties.json.        */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int main(int argc, char **argv) {
    const int N = argc > 1 ? atoi(argv[1]) : 2000;
    const long double band = argc > 2 ? strtold(argv[2], 0) : 1e-8L;
#pragma omp parallel
    {
        long double *lw = malloc(sizeof(long double) * (size_t)(N + 2));
#pragma omp for schedule(dynamic, 4)
        for (int n1 = 1; n1 <= N / 2; ++n1) {
            const int n2 = N - n1;
            for (int n = 1; n <= N / 2; ++n) {
                const int lo = n - n2 > 0 ? n - n2 : 0, hi = n < n1 ? n : n1;
                if (hi - lo < 3) continue;
                lw[lo] = 0.0L;
                for (int x = lo; x < hi; ++x)
                    lw[x + 1] = lw[x] + logl((long double)(n1 - x) * (long double)(n - x)) -
                                logl((long double)(x + 1) * (long double)(n2 - n + x + 1));
                int mode = lo;
                for (int x = lo; x <= hi; ++x) if (lw[x] > lw[mode]) mode = x;
                int y = hi;
                for (int x = lo; x < mode; ++x) {
                    while (y > mode && lw[y] < lw[x]) --y;         /* first y with lw[y] >= lw[x] */
                    for (int k = 0; k < 2; ++k) {
                        const int yy = y + k;
                        if (yy <= mode || yy > hi) continue;
                        /* structural exact ties: symmetric margins mirror the support */
                        if ((n1 == n2 && yy == n - x) || (2 * n == N && yy == n1 - x)) continue;
                        const long double d = fabsl(lw[yy] - lw[x]);
                        if (d > 1e-15L && d < band)
#pragma omp critical
                            printf("%d %d %d %d %d %.6Le\n", n1, n2, n, x, yy, lw[yy] - lw[x]);
                    }
                }
            }
        }
        free(lw);
    }
    return 0;
}
