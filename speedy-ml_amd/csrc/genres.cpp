// Reservoir construction (host, set-up time only): gen_res = makesparse + spectral radius + rescale.
//
// Replaces gen_res (src/mod_reservoir.f90:182-212), makesparse (src/mod_linalg.f90:180-218), shuffle
// (src/mod_utilities.f90:1569-1596) and the ARPACK driver sparse_eigen (src/mod_linalg.f90:220-514).
//   * makesparse: vals ~ U(0,1); the row list is floor(k/n) full random permutations of 1..n followed by a partial one
//     (k mod n entries), the column list likewise and independently -- so every row and every column of A carries
//     floor(k/n) or floor(k/n)+1 entries.  The reference draws from the Fortran compiler's RANDOM_NUMBER, which is not
//     reproducible across compilers; here the stream is SplitMix64 -> U(0,1), same construction.
//   * sparse_eigen asks ARPACK (dnaupd/dneupd, 'LM', nev=4) for the largest-magnitude eigenvalue.  A is entrywise
//     non-negative, so its spectral radius is its Perron root, which plain power iteration finds; ARPACK is not in the
//     image.  (The reference's `eigs = maxval(d)` also scans residual columns of a partly uninitialised array -- quirk
//     Q4 -- which is not reproduced: the Perron root is what that code intends.)
//   * vals <- vals / eigs * radius  (:196-198)
#include <cmath>
#include <cstdint>
#include <vector>

#include "common.h"

namespace {

struct SplitMix64 {
    uint64_t s;
    uint64_t next() { uint64_t z = (s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
    double uniform() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }   // [0,1)
};

// the source of uniform deviates: the library's own stream, or an array the caller supplies (sml_makesparse_from_draws)
struct Draws {
    SplitMix64 *rng;
    const double *given;
    long used;
    double next() { ++used; return given ? given[used - 1] : rng->uniform(); }
};

// K-shuffle "random choice without repeat" exactly as shuffle(n, returnsize, out): 1-based values
void kshuffle(Draws &rng, int n, int returnsize, int32_t *out, std::vector<int> &choices)
{
    choices.resize(n);
    for (int i = 0; i < n; ++i) choices[i] = i + 1;
    int n_chosen = 0;
    for (int i = 0; i < n; ++i) {
        const int pick = (int)(rng.next() * (n - n_chosen));           // `this` - 1
        const int tmp = choices[pick];
        if (i < returnsize) out[i] = tmp;
        choices[pick] = choices[n - n_chosen - 1];
        choices[n - n_chosen - 1] = tmp;
        ++n_chosen;
    }
}

}  // namespace

extern "C" {

static long makesparse_draw_count(int n, int k)
{
    const long calls = k > n ? 2L * (k / n) + (k % n ? 2 : 0) : 2;
    return (long)k + calls * n;
}

static int makesparse_impl(int n, int k, Draws &rng, int32_t *rows, int32_t *cols, double *vals)
{
    for (int e = 0; e < k; ++e) vals[e] = rng.next();
    std::vector<int> scratch;
    if (k > n) {
        const int counter = k / n, leftover = k % n;
        for (int i = 0; i < counter; ++i) {
            kshuffle(rng, n, n, rows + (size_t)i * n, scratch);
            kshuffle(rng, n, n, cols + (size_t)i * n, scratch);
        }
        if (leftover) {
            kshuffle(rng, n, leftover, rows + (size_t)counter * n, scratch);
            kshuffle(rng, n, leftover, cols + (size_t)counter * n, scratch);
        }
    } else {
        kshuffle(rng, n, k, rows, scratch);
        kshuffle(rng, n, k, cols, scratch);
    }
    return SML_OK;
}

int sml_makesparse(int n, int k, uint64_t seed, int32_t *rows, int32_t *cols, double *vals)
{
    SML_REQUIRE(n > 0 && k >= 0 && rows && cols && vals, "sml_makesparse: bad arguments");
    SplitMix64 sm{seed};
    Draws rng{&sm, nullptr, 0};
    return makesparse_impl(n, k, rng, rows, cols, vals);
}

/* makesparse on uniform deviates the caller supplies, consumed in the reference's order (RANDOM_NUMBER(vals), then one per
 * iteration of every shuffle call): a host that owns a Fortran RANDOM_NUMBER stream reproduces its own matrices, and the index
 * construction can be compared bit for bit.  ndraws must be at least sml_makesparse_draws(n, k). */
long sml_makesparse_draws(int n, int k) { return (n > 0 && k >= 0) ? makesparse_draw_count(n, k) : -1; }

int sml_makesparse_from_draws(int n, int k, const double *draws, long ndraws, int32_t *rows, int32_t *cols, double *vals)
{
    SML_REQUIRE(n > 0 && k >= 0 && draws && rows && cols && vals, "sml_makesparse_from_draws: bad arguments");
    const long need = makesparse_draw_count(n, k);
    SML_REQUIRE(ndraws >= need, "sml_makesparse_from_draws: %ld deviates supplied, %ld needed", ndraws, need);
    // the deviates index the shuffle's choice list (pick = int(a * remaining)): one outside [0, 1) -- or a NaN -- would read past it
    for (long i = 0; i < need; ++i)
        SML_REQUIRE(draws[i] >= 0.0 && draws[i] < 1.0, "sml_makesparse_from_draws: deviate %ld = %g is not in [0, 1)", i, draws[i]);
    Draws rng{nullptr, draws, 0};
    return makesparse_impl(n, k, rng, rows, cols, vals);
}

int sml_spectral_radius(int n, int k, const int32_t *rows, const int32_t *cols, const double *vals, double tol, int maxit,
                        double *lambda, int *iterations)
{
    SML_REQUIRE(n > 0 && k >= 0 && rows && cols && vals && lambda, "sml_spectral_radius: bad arguments");
    std::vector<double> x(n, 1.0 / std::sqrt((double)n)), y(n);
    double lam = 0.0;
    int it = 0;
    for (; it < maxit; ++it) {
        std::fill(y.begin(), y.end(), 0.0);
        for (int e = 0; e < k; ++e) y[rows[e] - 1] += vals[e] * x[cols[e] - 1];
        double nrm = 0.0;
        for (int i = 0; i < n; ++i) nrm += y[i] * y[i];
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) { lam = 0.0; break; }
        double dot = 0.0;
        for (int i = 0; i < n; ++i) dot += x[i] * y[i];         // Rayleigh quotient (x has unit norm)
        for (int i = 0; i < n; ++i) x[i] = y[i] / nrm;
        if (it > 0 && std::fabs(dot - lam) <= tol * std::fabs(dot)) { lam = dot; ++it; break; }
        lam = dot;
    }
    *lambda = lam;
    if (iterations) *iterations = it;
    return SML_OK;
}

int sml_gen_res(int n, int k, double radius, uint64_t seed, int32_t *rows, int32_t *cols, double *vals, double *eigs)
{
    int rc = sml_makesparse(n, k, seed, rows, cols, vals);
    if (rc) return rc;
    double lam = 0.0;
    if ((rc = sml_spectral_radius(n, k, rows, cols, vals, 1e-13, 2000, &lam, nullptr))) return rc;
    SML_REQUIRE(lam > 0.0, "sml_gen_res: spectral radius is zero");
    for (int e = 0; e < k; ++e) vals[e] = (vals[e] / lam) * radius;
    if (eigs) *eigs = lam;
    return SML_OK;
}

}  // extern "C"
