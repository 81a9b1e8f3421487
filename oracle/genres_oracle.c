/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reservoir construction's index bookkeeping:
 * makesparse (src/mod_linalg.f90:180-218) and shuffle (src/mod_utilities.f90:1569-1596).  See sml_oracle.h for scope.
 *
 * The reference draws from the Fortran compiler's RANDOM_NUMBER (not reproducible across compilers), so the
 * uniform deviates are an INPUT here, consumed in the reference's order: RANDOM_NUMBER(vals) first (k deviates), then one deviate
 * per iteration of every shuffle call, rows before cols inside each block of n.  With -r8 / -fdefault-real-8 (src/makefile:6,12)
 * shuffle's `real :: a` is a 64-bit real, and `this = a * (n - n_chosen) + 1` truncates on assignment to the integer `this`.
 * PARITY UNPINNED by a runnable reference (mod_linalg.f90 imports MKL_SPBLAS and mod_utilities); integer bookkeeping, restated
 * statement by statement and cross-checked in tests/test_genres.py against an independent numpy evaluation.
 * sparse_eigen (ARPACK dnaupd/dneupd, :220-514) is not restated: ARPACK-NG is absent from the image; the value it is meant to
 * return -- the largest-magnitude eigenvalue -- is checked against scipy.sparse.linalg.eigs instead (quirk Q4: the reference's
 * `eigs = maxval(d)` also scans the imaginary-part and residual columns of a partly uninitialised d(30,3); not reproducible). */
#include "sml_oracle.h"
#include <stdlib.h>

/* shuffle(n, returnsize, shufflereturn): src/mod_utilities.f90:1569-1596.  Returns the number of deviates consumed (n). */
static int go_shuffle(int n, int returnsize, const double *a, int32_t *shufflereturn)
{
    int *choices = (int *)malloc(sizeof(int) * (size_t)n), *choiceshuffle = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 1; i <= n; ++i) choices[i - 1] = i;
    int n_chosen = 0;
    for (int i = 1; i <= n; ++i) {
        const int this_ = (int)(a[i - 1] * (double)(n - n_chosen) + 1.0);      /* this = a*(n - n_chosen) + 1 */
        const int tmp = choices[this_ - 1];
        choiceshuffle[i - 1] = tmp;
        choices[this_ - 1] = choices[n - n_chosen - 1];
        choices[n - n_chosen - 1] = tmp;
        n_chosen = n_chosen + 1;
    }
    for (int i = 0; i < returnsize; ++i) shufflereturn[i] = choiceshuffle[i];
    free(choices); free(choiceshuffle);
    return n;
}

/* makesparse: src/mod_linalg.f90:180-218.  draws: k + (number of shuffle calls) * n deviates.  Returns the count consumed. */
int go_makesparse(int n, int k, const double *draws, int32_t *rows, int32_t *cols, double *vals)
{
    int used = 0;
    for (int e = 0; e < k; ++e) vals[e] = draws[used++];
    if (k > n) {
        const int counter = k / n, leftover = k % n;
        int i;
        for (i = 1; i <= counter; ++i) {
            used += go_shuffle(n, n, draws + used, rows + (size_t)(i - 1) * n);
            used += go_shuffle(n, n, draws + used, cols + (size_t)(i - 1) * n);
        }
        if (leftover != 0) {                                   /* (i = counter + 1 after the loop, as in Fortran) */
            used += go_shuffle(n, leftover, draws + used, rows + (size_t)(i - 1) * n);
            used += go_shuffle(n, leftover, draws + used, cols + (size_t)(i - 1) * n);
        }
    } else {
        used += go_shuffle(n, k, draws + used, rows);
        used += go_shuffle(n, k, draws + used, cols);
    }
    return used;
}
