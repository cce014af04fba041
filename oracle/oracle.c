/*
 * oracle.c -- CPU restatement of Scoary's association hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under scoary_amd/ may import, link or
 * execute this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, and only as the checker / the reported baseline.
 *
 * What it restates (reference = AdmiralenOla/Scoary v1.6.16, paths relative
 * to the reference checkout):
 *   orc_counts_dense    scoary/methods.py:930-982  Perform_statistics
 *   orc_counts_packed   same arithmetic on the bit-packed layout
 *   orc_fisher          scoary/methods.py:854 -> scipy.stats.fisher_exact
 *                       (third-party, SciPy 1.15.3 two-sided rule; see below)
 *   orc_perm_labels     scoary/methods.py:1371-1384 PermuteGTC (label shuffle
 *                       among valid isolates; counter-based RNG, see S4)
 *   orc_permute_r       scoary/methods.py:1314-1369 Permute, with the Fisher
 *                       statistic the north_star prescribes
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks these functions
 * against tests/golden/ -- vectors captured from the real reference and from
 * SciPy 1.15.3 by tests/golden/make_golden.py.
 *
 * Third-party arithmetic: scipy.stats.fisher_exact is not part of the
 * reference tree.  Its published two-sided rule is restated here in "set"
 * form:   p = sum{ pmf(x) : pmf(x) <= pmf(a_obs) * (1 + 1e-14) } over the
 * hypergeometric support, min(p, 1); (nan, 1.0) when a margin is zero; sample
 * odds ratio a*d/(b*c), inf when b*c == 0.  pmf ratios come from the exact
 * recurrence  w(x+1)/w(x) = (n1-x)(n-x) / ((x+1)(n2-n+x+1))  normalised at the
 * mode (an lgamma table loses ~ulp(lgamma(N)) and misses 1e-12 at N >= 2000,
 * SURVEY finding 4).
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* Spec S3 tie rule = SciPy's: a support point x is "as extreme as observed" iff
 * w(x) <= w(a_obs) * (1 + 1e-14) -- scipy/stats/_stats_py.py, fisher_exact: epsilon = 1e-14,
 * gamma = 1 + epsilon.  The fp64 recurrence carries ~2e-16 per step, so a comparison that
 * comes out within ORC_AMBIG of equality is settled EXACTLY, on big integers (hg_leq_exact):
 * exact ties (symmetric margins) are ties, and the closest non-equal weights that exist
 * (1.7e-12 apart at N = 1972, tests/golden/near_ties.json) are told apart. */
#define ORC_AMBIG 1e-9

/* ------------------------------------------------------------------ S1 -- */
/* Row-major bit packing: bit i of word w of a row = isolate 64*w + i.      */
void orc_pack_rows(const uint8_t *dense, int64_t G, int64_t N, uint64_t *out)
{
    int64_t W = (N + 63) / 64;
    memset(out, 0, (size_t)(G * W) * sizeof(uint64_t));
    for (int64_t g = 0; g < G; ++g)
        for (int64_t i = 0; i < N; ++i)
            if (dense[g * N + i])
                out[g * W + (i >> 6)] |= (uint64_t)1 << (i & 63);
}

/* ------------------------------------------------------------------ S2 -- */
/* Perform_statistics, isolate by isolate (methods.py:940-965).
 * trait[i]: 0 / 1, or 2 = missing for this trait (the reference drops such
 * isolates from the traits dict, methods.py:591-598).  out: G x 4 =
 * tpgp, tpgn, tngp, tngn. */
void orc_counts_dense(const uint8_t *genes, const uint8_t *trait, int64_t G,
                      int64_t N, int32_t *out)
{
    for (int64_t g = 0; g < G; ++g) {
        int32_t tpgp = 0, tpgn = 0, tngp = 0, tngn = 0;
        for (int64_t i = 0; i < N; ++i) {
            uint8_t t = trait[i], p = genes[g * N + i];
            if (t > 1)
                continue;
            if (t == 1 && p)
                ++tpgp;
            else if (t == 1)
                ++tpgn;
            else if (p)
                ++tngp;
            else
                ++tngn;
        }
        out[g * 4 + 0] = tpgp;
        out[g * 4 + 1] = tpgn;
        out[g * 4 + 2] = tngp;
        out[g * 4 + 3] = tngn;
    }
}

/* Same tallies on packed rows: a = popc(g&t), gm = popc(g&m), npos = popc(t),
 * nval = popc(m)  =>  tpgp = a, tpgn = npos-a, tngp = gm-a,
 * tngn = nval-npos-gm+a.  out: G x T x 4. */
void orc_counts_packed(const uint64_t *genes, const uint64_t *traits,
                       const uint64_t *masks, int64_t G, int64_t T, int64_t W,
                       int32_t *out)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int64_t g = 0; g < G; ++g) {
        const uint64_t *gr = genes + g * W;
        for (int64_t t = 0; t < T; ++t) {
            const uint64_t *tr = traits + t * W, *mr = masks + t * W;
            int32_t a = 0, gm = 0, npos = 0, nval = 0;
            for (int64_t w = 0; w < W; ++w) {
                a += __builtin_popcountll(gr[w] & tr[w]);
                gm += __builtin_popcountll(gr[w] & mr[w]);
                npos += __builtin_popcountll(tr[w]);
                nval += __builtin_popcountll(mr[w]);
            }
            int32_t *o = out + (g * T + t) * 4;
            o[0] = a;
            o[1] = npos - a;
            o[2] = gm - a;
            o[3] = nval - npos - gm + a;
        }
    }
}

/* ------------------------------------------------------------------ S3 -- */
/* Unnormalised hypergeometric weights over the support [lo, hi] of
 * x = tpgp given margins (n1 = trait positives, n2 = trait negatives,
 * n = gene margin), mode weight = 1.  Returns lo; *len = hi-lo+1. */
static int64_t hg_weights(int64_t n1, int64_t n2, int64_t n, double *w,
                          int64_t *len)
{
    int64_t lo = n - n2 > 0 ? n - n2 : 0;
    int64_t hi = n < n1 ? n : n1;
    int64_t mode = (int64_t)(((double)(n + 1) * (double)(n1 + 1)) /
                             (double)(n1 + n2 + 2));
    if (mode < lo)
        mode = lo;
    if (mode > hi)
        mode = hi;
    w[mode - lo] = 1.0;
    for (int64_t x = mode; x < hi; ++x)
        w[x + 1 - lo] = w[x - lo] * ((double)(n1 - x) * (double)(n - x)) /
                        ((double)(x + 1) * (double)(n2 - n + x + 1));
    for (int64_t x = mode; x > lo; --x)
        w[x - 1 - lo] = w[x - lo] * ((double)x * (double)(n2 - n + x)) /
                        ((double)(n1 - x + 1) * (double)(n - x + 1));
    *len = hi - lo + 1;
    return lo;
}

/* ---- exact comparison of two hypergeometric weights -------------------------
 * Little-endian arrays of 32-bit limbs, multiplied in place by small factors. */
typedef struct { uint32_t *v; int64_t n, cap; } orc_big;
static void big_init(orc_big *b, int64_t cap)
{
    b->v = (uint32_t *)calloc((size_t)cap, sizeof(uint32_t));
    b->cap = cap;
    b->n = 1;
    b->v[0] = 1;
}
static void big_mul(orc_big *b, uint32_t f)
{
    uint64_t carry = 0;
    for (int64_t i = 0; i < b->n; ++i) {
        uint64_t t = (uint64_t)b->v[i] * f + carry;
        b->v[i] = (uint32_t)t;
        carry = t >> 32;
    }
    if (carry) {
        if (b->n >= b->cap) abort();             /* sized by the caller: cannot happen */
        b->v[b->n++] = (uint32_t)carry;
    }
}
static int big_cmp(const orc_big *a, const orc_big *b)
{
    int64_t na = a->n, nb = b->n;
    while (na > 1 && a->v[na - 1] == 0) --na;
    while (nb > 1 && b->v[nb - 1] == 0) --nb;
    if (na != nb) return na < nb ? -1 : 1;
    for (int64_t i = na - 1; i >= 0; --i)
        if (a->v[i] != b->v[i]) return a->v[i] < b->v[i] ? -1 : 1;
    return 0;
}
/* w(x) <= w(a) * (1 + 1e-14), exactly.  For x > a:  w(x)/w(a) = NUM/DEN with
 * NUM = prod_{j=a}^{x-1} (n1-j)(n-j),  DEN = prod (j+1)(n2-n+j+1)  (the recurrence of
 * hg_weights); for x < a the roles of NUM and DEN swap. */
static int hg_leq_exact(int64_t n1, int64_t n2, int64_t n, int64_t x, int64_t a)
{
    if (x == a) return 1;
    int64_t from = x < a ? x : a, to = x < a ? a : x, k = to - from;
    orc_big num, den;
    big_init(&num, 2 * k + 16);                  /* two factors < 2^32 per step, + the constant */
    big_init(&den, 2 * k + 16);
    for (int64_t j = from; j < to; ++j) {        /* every factor < 2^32 (N < 2^32) */
        big_mul(&num, (uint32_t)(n1 - j));
        big_mul(&num, (uint32_t)(n - j));
        big_mul(&den, (uint32_t)(j + 1));
        big_mul(&den, (uint32_t)(n2 - n + j + 1));
    }
    /* x > a:  NUM/DEN <= (1e14+1)/1e14  <=>  NUM * 1e14 <= DEN * (1e14+1)
     * x < a:  DEN/NUM <= (1e14+1)/1e14  <=>  DEN * 1e14 <= NUM * (1e14+1) */
    orc_big *left = x > a ? &num : &den, *right = x > a ? &den : &num;
    big_mul(left, 10000000u);                    /* 1e14 = 1e7 * 1e7 */
    big_mul(left, 10000000u);
    big_mul(right, 29u);                         /* 1e14 + 1 = 29 * 101 * 281 * 121499449 */
    big_mul(right, 101u);
    big_mul(right, 281u);
    big_mul(right, 121499449u);
    int r = big_cmp(left, right) <= 0;
    free(num.v);
    free(den.v);
    return r;
}
/* w(x) <= w(a) (1 + 1e-14) given the fp64 weights: decided in fp64 when it is clear, exactly
 * when the two weights agree to ORC_AMBIG. */
static int hg_leq(const double *w, int64_t lo, int64_t n1, int64_t n2, int64_t n, int64_t x,
                  int64_t a)
{
    double wx = w[x - lo], wa = w[a - lo];
    if (wx <= wa * (1.0 - ORC_AMBIG)) return 1;
    if (wx > wa * (1.0 + ORC_AMBIG)) return 0;
    return hg_leq_exact(n1, n2, n, x, a);
}

/* ---- S3, N <= 170: scipy.stats.fisher_exact's own double ---------------------------------------
 * What the reference calls at scoary/methods.py:854 is third-party arithmetic (SciPy 1.15.3 over
 * Boost.Math).  For populations up to 170 -- max_factorial<double> -- Boost evaluates the hypergeometric
 * pmf from its table of factorials (hypergeometric_pdf_factorial_imp) and the tails by the term recurrence
 * of hypergeometric_cdf_imp, in plain double; scipy.stats.hypergeom clips them to [0, 1] and handles the
 * ends of the support; fisher_exact adds the tail on the observed side to the tail beyond the point its
 * _binary_search finds on the other side (scipy/stats/_stats_py.py).  Restated here operation by operation
 * (published algorithms; none of that code is in /root/reference) and PINNED: equal to SciPy bit for bit on
 * the 5189 tables with N <= 170 of tests/golden/fisher_grid.npz and on every p-value of the reference's
 * exampledata run (tests/test_oracle_golden.py).  The reference's example data has N = 100.  Above 170 Boost
 * factorises into primes and calls pow(): not restated, the set rule below is the p there (1e-12). */
#define ORC_SCIPY_SMALL_N 170
static const double orc_factorial[ORC_SCIPY_SMALL_N + 1] = {
    1.0, 1.0, 2.0, 6.0, 24.0, 120.0, 720.0, 5040.0, 40320.0, 362880.0, 3628800.0, 39916800.0, 479001600.0,
    6227020800.0, 87178291200.0, 1307674368000.0, 20922789888000.0, 355687428096000.0, 6402373705728000.0,
    1.21645100408832e+17, 2.43290200817664e+18, 5.109094217170944e+19, 1.1240007277776077e+21, 2.585201673888498e+22,
    6.204484017332394e+23, 1.5511210043330986e+25, 4.0329146112660565e+26, 1.0888869450418352e+28,
    3.0488834461171387e+29, 8.841761993739702e+30, 2.6525285981219107e+32, 8.222838654177922e+33, 2.631308369336935e+35,
    8.683317618811886e+36, 2.9523279903960416e+38, 1.0333147966386145e+40, 3.7199332678990125e+41,
    1.3763753091226346e+43, 5.230226174666011e+44, 2.0397882081197444e+46, 8.159152832478977e+47, 3.345252661316381e+49,
    1.40500611775288e+51, 6.041526306337383e+52, 2.658271574788449e+54, 1.1962222086548019e+56, 5.502622159812089e+57,
    2.5862324151116818e+59, 1.2413915592536073e+61, 6.082818640342675e+62, 3.0414093201713376e+64,
    1.5511187532873822e+66, 8.065817517094388e+67, 4.2748832840600255e+69, 2.308436973392414e+71, 1.2696403353658276e+73,
    7.109985878048635e+74, 4.0526919504877214e+76, 2.3505613312828785e+78, 1.3868311854568984e+80,
    8.32098711274139e+81, 5.075802138772248e+83, 3.146997326038794e+85, 1.98260831540444e+87, 1.2688693218588417e+89,
    8.247650592082472e+90, 5.443449390774431e+92, 3.647111091818868e+94, 2.4800355424368305e+96, 1.711224524281413e+98,
    1.1978571669969892e+100, 8.504785885678623e+101, 6.1234458376886085e+103, 4.4701154615126844e+105,
    3.307885441519386e+107, 2.48091408113954e+109, 1.8854947016660504e+111, 1.4518309202828587e+113,
    1.1324281178206297e+115, 8.946182130782976e+116, 7.156945704626381e+118, 5.797126020747368e+120,
    4.753643337012842e+122, 3.945523969720659e+124, 3.314240134565353e+126, 2.81710411438055e+128,
    2.4227095383672734e+130, 2.107757298379528e+132, 1.8548264225739844e+134, 1.650795516090846e+136,
    1.4857159644817615e+138, 1.352001527678403e+140, 1.2438414054641308e+142, 1.1567725070816416e+144,
    1.087366156656743e+146, 1.032997848823906e+148, 9.916779348709496e+149, 9.619275968248212e+151,
    9.426890448883248e+153, 9.332621544394415e+155, 9.332621544394415e+157, 9.42594775983836e+159,
    9.614466715035127e+161, 9.90290071648618e+163, 1.0299016745145628e+166, 1.081396758240291e+168,
    1.1462805637347084e+170, 1.226520203196138e+172, 1.324641819451829e+174, 1.4438595832024937e+176,
    1.588245541522743e+178, 1.7629525510902446e+180, 1.974506857221074e+182, 2.2311927486598138e+184,
    2.5435597334721877e+186, 2.925093693493016e+188, 3.393108684451898e+190, 3.969937160808721e+192,
    4.684525849754291e+194, 5.574585761207606e+196, 6.689502913449127e+198, 8.094298525273444e+200,
    9.875044200833601e+202, 1.214630436702533e+205, 1.506141741511141e+207, 1.882677176888926e+209,
    2.372173242880047e+211, 3.0126600184576594e+213, 3.856204823625804e+215, 4.974504222477287e+217,
    6.466855489220474e+219, 8.47158069087882e+221, 1.1182486511960043e+224, 1.4872707060906857e+226,
    1.9929427461615188e+228, 2.6904727073180504e+230, 3.659042881952549e+232, 5.012888748274992e+234,
    6.917786472619489e+236, 9.615723196941089e+238, 1.3462012475717526e+241, 1.898143759076171e+243,
    2.695364137888163e+245, 3.854370717180073e+247, 5.5502938327393044e+249, 8.047926057471992e+251,
    1.1749972043909107e+254, 1.727245890454639e+256, 2.5563239178728654e+258, 3.80892263763057e+260,
    5.713383956445855e+262, 8.62720977423324e+264, 1.3113358856834524e+267, 2.0063439050956823e+269,
    3.0897696138473508e+271, 4.789142901463394e+273, 7.471062926282894e+275, 1.1729568794264145e+278,
    1.853271869493735e+280, 2.9467022724950384e+282, 4.7147236359920616e+284, 7.590705053947219e+286,
    1.2296942187394494e+289, 2.0044015765453026e+291, 3.287218585534296e+293, 5.423910666131589e+295,
    9.003691705778438e+297, 1.503616514864999e+300, 2.5260757449731984e+302, 4.269068009004705e+304,
    7.257415615307999e+306};

/* Above N = 170 (and up to the 10 000th prime, 104 729) Boost evaluates the same quotient of factorials through its
 * PRIME FACTORISATION (boost/math/distributions/detail/hypergeometric_pdf.hpp, hypergeometric_pdf_prime_loop_imp):
 * for every prime p <= N the exponent of p in n! r! (N-n)! (N-r)! / (N! x! (n-x)! (r-x)! (N-n-r+x)!) by Legendre's
 * formula, the running product multiplied by p^e in ascending order of p -- a partial product that would overflow or
 * underflow is set aside and a new one started --, and at the end the partial products multiplied together, taking
 * one >= 1 while the running value is <= 1 and one < 1 otherwise.  p^|e| is an exact double for every exponent that
 * can occur (|e| <= 2 log_p N), so only the order of the multiplications matters; a negative exponent is 1 / p^|e|
 * (one rounding).  Restated from the published algorithm and PINNED bit for bit against scipy.stats.hypergeom.pmf
 * and scipy.stats.fisher_exact (tests/test_oracle_golden.py, fisher_grid.npz: every table above N = 170). */
#define ORC_SCIPY_MAX_N 104723
static uint32_t *orc_primes = NULL;
static int64_t orc_nprimes = 0;
static void orc_primes_init(void)
{
    if (orc_primes)
        return;
    const int64_t top = 104730;
    uint8_t *sieve = (uint8_t *)calloc((size_t)top + 1, 1);
    uint32_t *pr = (uint32_t *)malloc(10001 * sizeof(uint32_t));
    int64_t np = 0;
    for (int64_t i = 2; i <= top; ++i) {
        if (sieve[i])
            continue;
        pr[np++] = (uint32_t)i;
        for (int64_t j = i * i; j <= top; j += i)
            sieve[j] = 1;
    }
    free(sieve);
    orc_nprimes = np;
    orc_primes = pr;
}
static double bm_hypergeometric_pdf_prime(int64_t x, int64_t r, int64_t n, int64_t N)
{
    /* partial products, part[np - 1] the current one (Boost: a linked list on its call stack); N = 10^5 can need
     * more than a thousand of them (the >= 1 factors alone reach 10^450000) */
    int cap = 64, np = 1;
    double *part = (double *)malloc((size_t)cap * sizeof(double));
    part[0] = 1.0;
    for (int64_t k = 0; k < orc_nprimes && (int64_t)orc_primes[k] <= N; ++k) {
        const int64_t p = orc_primes[k];
        int64_t e = 0;
        for (int64_t base = p; base <= N; base *= p) {
            e += n / base + r / base + (N - n) / base + (N - r) / base;
            e -= N / base + x / base + (n - x) / base + (r - x) / base + (N - n - r + x) / base;
        }
        if (!e)
            continue;
        double v = 1.0;
        for (int64_t i = 0; i < (e < 0 ? -e : e); ++i)
            v *= (double)p;                /* exact */
        if (e < 0)
            v = 1.0 / v;
        if ((v > 1 && DBL_MAX / v < part[np - 1]) || (v < 1 && DBL_MIN / v > part[np - 1])) {
            if (np == cap)
                part = (double *)realloc(part, (size_t)(cap *= 2) * sizeof(double));
            part[np++] = v;                /* the next product would overflow / underflow: set aside, start anew */
            continue;
        }
        part[np - 1] *= v;
    }
    /* newest first, as Boost walks its list: i over the entries >= 1, j over those < 1 */
    int i = np - 1, j = np - 1;
    while (i >= 0 && part[i] < 1) --i;
    while (j >= 0 && part[j] >= 1) --j;
    double prod = 1.0;
    while (i >= 0 || j >= 0) {
        while (i >= 0 && (prod <= 1 || j < 0)) {
            prod *= part[i--];
            while (i >= 0 && part[i] < 1) --i;
        }
        while (j >= 0 && (prod >= 1 || i < 0)) {
            prod *= part[j--];
            while (j >= 0 && part[j] >= 1) --j;
        }
    }
    free(part);
    return prod;
}

/* pmf of x successes in n draws, r successes among N items */
static double bm_hypergeometric_pdf(int64_t x, int64_t r, int64_t n, int64_t N)
{
    if (N > ORC_SCIPY_SMALL_N)
        return bm_hypergeometric_pdf_prime(x, r, n, N);
    const double up[3] = {orc_factorial[r], orc_factorial[N - n], orc_factorial[N - r]};
    const double down[5] = {orc_factorial[N], orc_factorial[x], orc_factorial[n - x],
                            orc_factorial[r - x], orc_factorial[N - n - r + x]};
    double value = orc_factorial[n];
    int iu = 0, id = 0;
    while (iu < 3 || id < 5) {
        for (; id < 5 && (value >= 1 || iu >= 3); ++id)
            value /= down[id];
        for (; iu < 3 && (value <= 1 || id >= 5); ++iu)
            value *= up[iu];
    }
    return value;
}
/* P(X <= x) (complement == 0) or P(X > x) (complement == 1) */
static double bm_hypergeometric_cdf(int64_t x, int64_t r, int64_t n, int64_t N, int complement)
{
    const double eps = 2.220446049250313e-16;
    double mode = floor((double)(r + 1) * (double)(n + 1) / (double)(N + 2));
    double acc = 0, term;
    if ((double)x < mode) {
        int64_t floor_x = n + r - N > 0 ? n + r - N : 0;
        acc = term = bm_hypergeometric_pdf(x, r, n, N);
        while (term > (complement ? 1.0 : acc) * eps) {
            term = (double)x * (double)((N + x) - n - r) * term /
                   ((double)(1 + n - x) * (double)(1 + r - x));
            acc += term;
            if (x == floor_x)
                break;
            --x;
        }
    } else {
        int64_t cap = r < n ? r : n;
        complement = !complement;
        if (x != cap) {
            ++x;
            acc = term = bm_hypergeometric_pdf(x, r, n, N);
            while (x <= cap && term > (complement ? 1.0 : acc) * eps) {
                term = (double)(n - x) * (double)(r - x) * term /
                       ((double)(x + 1) * (double)((N + x + 1) - n - r));
                acc += term;
                ++x;
            }
        }
    }
    return complement ? 1 - acc : acc;
}
static double clip01(double v) { return v < 0 ? 0 : (v > 1 ? 1 : v); }
/* scipy.stats.hypergeom.pmf / cdf / sf (k, M, n, N): M items, n good, N drawn */
static double sp_pmf(int64_t k, int64_t M, int64_t n, int64_t N)
{
    int64_t a = N - (M - n) > 0 ? N - (M - n) : 0, b = n < N ? n : N;
    return (k < a || k > b) ? 0.0 : clip01(bm_hypergeometric_pdf(k, n, N, M));
}
static double sp_cdf(int64_t k, int64_t M, int64_t n, int64_t N)
{
    int64_t a = N - (M - n) > 0 ? N - (M - n) : 0, b = n < N ? n : N;
    if (k < a) return 0.0;
    if (k >= b) return 1.0;
    return clip01(bm_hypergeometric_cdf(k, n, N, M, 0));
}
static double sp_sf(int64_t k, int64_t M, int64_t n, int64_t N)
{
    int64_t a = N - (M - n) > 0 ? N - (M - n) : 0, b = n < N ? n : N;
    if (k < a) return 1.0;
    if (k >= b) return 0.0;
    return clip01(bm_hypergeometric_cdf(k, n, N, M, 1));
}
/* fisher_exact(..., alternative='two-sided') of [[a, b], [c, d]], non-degenerate margins */
static double scipy_fisher_two_sided(int64_t a, int64_t b, int64_t c, int64_t d)
{
    const int64_t n1 = a + b, n2 = c + d, n = a + c, M = n1 + n2;
    const int64_t mode = (int64_t)((double)((n + 1) * (n1 + 1)) / (double)(n1 + n2 + 2));
    const double pexact = sp_pmf(a, M, n1, n), pmode = sp_pmf(mode, M, n1, n);
    const double gamma = 1 + 1e-14;
    if (fabs(pexact - pmode) / fmax(pexact, pmode) <= 1e-14)
        return 1.0;
    /* _binary_search(f, d, lo, hi): f ascending on [lo, hi] */
    const int lower_side = a < mode;
    const double s = lower_side ? -1.0 : 1.0, dval = s * (pexact * gamma);
    double first;
    int64_t lo, hi, guess = 0;
    int found = 0;
    if (lower_side) {
        first = sp_cdf(a, M, n1, n);
        if (sp_pmf(n, M, n1, n) > pexact * gamma)
            return first;
        lo = mode;
        hi = n;
    } else {
        first = sp_sf(a - 1, M, n1, n);
        if (sp_pmf(0, M, n1, n) > pexact * gamma)
            return first;
        lo = 0;
        hi = mode;
    }
    while (lo < hi && !found) {
        int64_t mid = lo + (hi - lo) / 2;
        double midval = s * sp_pmf(mid, M, n1, n);
        if (midval < dval)
            lo = mid + 1;
        else if (midval > dval)
            hi = mid - 1;
        else {
            guess = mid;
            found = 1;
        }
    }
    if (!found)
        guess = s * sp_pmf(lo, M, n1, n) <= dval ? lo : lo - 1;
    double p = first + (lower_side ? sp_sf(guess, M, n1, n) : sp_cdf(guess, M, n1, n));
    return p < 1.0 ? p : 1.0;
}

/* Two-sided Fisher exact p and sample odds ratio of [[a, b], [c, d]]
 * (= [[tpgp, tpgn], [tngp, tngn]], methods.py:842-845). */
void orc_fisher(int64_t a, int64_t b, int64_t c, int64_t d, double *p_out,
                double *or_out)
{
    int64_t n1 = a + b, n2 = c + d, n = a + c;
    if (n1 == 0 || n2 == 0 || n == 0 || b + d == 0) {
        *p_out = 1.0;
        *or_out = NAN;
        return;
    }
    *or_out = (c > 0 && b > 0) ? ((double)a * (double)d) / ((double)c * (double)b)
                               : INFINITY;
    if (n1 + n2 <= ORC_SCIPY_SMALL_N) {          /* SciPy's own double (above) */
        *p_out = scipy_fisher_two_sided(a, b, c, d);
        return;
    }
    int64_t cap = (n < n1 ? n : n1) + 2;
    double *w = (double *)malloc((size_t)cap * sizeof(double));
    int64_t len, lo = hg_weights(n1, n2, n, w, &len);
    double tot = 0.0, inc = 0.0;
    int all = 1;
    for (int64_t i = 0; i < len; ++i) {
        tot += w[i];
        if (hg_leq(w, lo, n1, n2, n, lo + i, a))
            inc += w[i];
        else
            all = 0;
    }
    double p = all ? 1.0 : inc / tot;
    *p_out = p < 1.0 ? p : 1.0;
    free(w);
}

/* scipy.stats.fisher_exact(...)[1] itself, to the last bit, for tables of up to ORC_SCIPY_MAX_N isolates (what the
 * command line's result files print: scoary_fisher_scipy, k_fisher_scipy); NaN beyond or with an empty margin. */
double orc_fisher_scipy(int64_t a, int64_t b, int64_t c, int64_t d)
{
    if (a + b == 0 || c + d == 0 || a + c == 0 || b + d == 0)
        return 1.0;
    if (a + b + c + d > ORC_SCIPY_MAX_N)
        return NAN;
    orc_primes_init();
    return scipy_fisher_two_sided(a, b, c, d);
}
void orc_fisher_scipy_many(const int32_t *counts, int64_t M, double *p)
{
    orc_primes_init();
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16)
#endif
    for (int64_t i = 0; i < M; ++i)
        p[i] = orc_fisher_scipy(counts[i * 4], counts[i * 4 + 1], counts[i * 4 + 2], counts[i * 4 + 3]);
}

void orc_fisher_many(const int32_t *counts, int64_t M, double *p, double *orr)
{
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
    for (int64_t i = 0; i < M; ++i)
        orc_fisher(counts[i * 4], counts[i * 4 + 1], counts[i * 4 + 2],
                   counts[i * 4 + 3], p + i, orr + i);
}

/* ------------------------------------------------------------------ S4 -- */
/* Philox4x32-10 (Salmon et al., SC'11), the counter-based generator both
 * sides use so CPU and GPU regenerate identical permutations. */
void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2],
                       uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0;
        c1 = n1;
        c2 = n2;
        c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0;
    out[1] = c1;
    out[2] = c2;
    out[3] = c3;
}

/* Spec S4 (round 5): one label permutation = a uniformly random subset of size npos of
 * the valid isolates (== shuffling the 0/1 labels among the non-missing isolates,
 * PermuteGTC methods.py:1377-1383).  The reference's shuffle is unseeded, so the law is
 * the contract, not the bits; this build's bits are defined so that NO step depends on
 * the isolate before it (rounds 1-4 used sequential selection sampling: a chain of N
 * dependent steps per permutation):
 *
 *   m     = min(npos, nval - npos) marks are placed (the complement is taken at the end
 *           when npos > nval / 2), so the mark fraction is <= 1/2;
 *   q/256 = an 8-bit mark probability a little BELOW m/nval (orc_perm_plan);
 *   round 0: every valid isolate i is marked independently with probability q/256.  The
 *           32 permutations B*32 .. B*32+31 share eight 32-bit words R_0..R_7 per isolate,
 *           R_0..3 = Philox(key = seed, ctr = (i, B, t, "SCOB")), R_4..7 = the same with
 *           "SCOC"; bit b of  X = fold_{j=0..7} (q>>j & 1 ? X | R_j : X & R_j), X = 0 at
 *           the start, is the mark of permutation B*32+b (bit-sliced comparison of an 8-bit
 *           uniform number with q: probability exactly q/256, independent across b and i);
 *   fix-up: K = marks placed; conditional on K the marked set is uniform among the K-subsets
 *           of the valid isolates, so adding (K < m) or removing (K > m) uniformly chosen
 *           isolates keeps it uniform.  Draws c = 0, 1, 2, ...: u = word c&3 of
 *           Philox(key, ctr = (c>>2, pi, t, "SCOD")); prod = u * N; rejected if the low
 *           32 bits of prod are < 2^32 mod N (Lemire: pos = prod >> 32 is then EXACTLY
 *           uniform on 0..N-1); pos is toggled iff it is a candidate (add: valid and
 *           unmarked; remove: marked); stop when K = m.
 * Under ideal random words the result is exactly uniform (no 2^-32 rounding term). */
#define ORC_DOM_BERN_LO 0x53434F42u /* "SCOB": counter word 3, words R_0..R_3 */
#define ORC_DOM_BERN_HI 0x53434F43u /* "SCOC": words R_4..R_7 */
#define ORC_DOM_FIX 0x53434F44u     /* "SCOD": fix-up position draws */
#define ORC_FIX_MAX_CALLS (1u << 20) /* safety stop of the fix-up loop (never reached on consistent input) */

static uint64_t isqrt_u64(uint64_t v)
{
    uint64_t r = 0;
    for (int s = 31; s >= 0; --s) {
        uint64_t c = r | ((uint64_t)1 << s);
        if (c * c <= v) r = c;
    }
    return r;
}

/* marks to place, whether the complement is taken, and the 8-bit round-0 probability.
 * The target mean of round 0 sits b = 2.4 sigma (1 - 2 m/nval) below m: removing a mark
 * costs ~nval/m position draws, adding one ~nval/(nval-m), and with this b the longest
 * fix-up among many permutations costs about the same on either side (4.8 sigma draws). */
void orc_perm_plan(int64_t npos, int64_t nval, int64_t *m_out, int *flip_out, uint32_t *q_out)
{
    int flip = 2 * npos > nval;
    int64_t m = flip ? nval - npos : npos;
    uint32_t q = 0;
    if (m > 0 && nval > 0) {
        uint64_t s = isqrt_u64((uint64_t)m * (uint64_t)(nval - m) / (uint64_t)nval);
        uint64_t b = 12u * s * (uint64_t)(nval - 2 * m) / (5u * (uint64_t)nval);
        uint64_t target = (uint64_t)m > b ? (uint64_t)m - b : 0;
        q = (uint32_t)(256u * target / (uint64_t)nval);
    }
    *m_out = m < 0 ? 0 : m;
    *flip_out = flip;
    *q_out = q;
}

/* The 32 permutations pi = 32 B + b, b = 0..31, of trait t: out[b * W + w] = rows64 label
 * bits of permutation 32 B + b (W = ceil(N/64)). */
void orc_perm_block(uint64_t seed, uint32_t t, uint32_t B, const uint64_t *mask,
                    int64_t npos, int64_t N, uint64_t *out)
{
    int64_t W = (N + 63) / 64, nval = 0, m;
    int flip;
    uint32_t q;
    uint32_t key[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    memset(out, 0, (size_t)(32 * W) * sizeof(uint64_t));
    if (N < 1) return;
    for (int64_t w = 0; w < W; ++w) nval += __builtin_popcountll(mask[w]);
    orc_perm_plan(npos, nval, &m, &flip, &q);
    uint32_t *X = (uint32_t *)calloc((size_t)N, sizeof(uint32_t));
    int64_t K[32];
    memset(K, 0, sizeof K);
#define VALID(i) ((mask[(i) >> 6] >> ((i) & 63)) & 1)
    if (q)
        for (int64_t i = 0; i < N; ++i) {
            if (!VALID(i)) continue;
            uint32_t R[8], x = 0;
            uint32_t ctr[4] = {(uint32_t)i, B, t, ORC_DOM_BERN_LO};
            orc_philox4x32_10(ctr, key, R);
            ctr[3] = ORC_DOM_BERN_HI;
            orc_philox4x32_10(ctr, key, R + 4);
            for (int j = 0; j < 8; ++j) x = ((q >> j) & 1u) ? (x | R[j]) : (x & R[j]);
            X[i] = x;
            for (int b = 0; b < 32; ++b) K[b] += (x >> b) & 1u;
        }
    const uint32_t reject_below = (uint32_t)(((uint64_t)1 << 32) % (uint64_t)N);
    for (int b = 0; b < 32; ++b) {
        const uint32_t pi = B * 32u + (uint32_t)b, bit = 1u << b;
        int64_t d = m - K[b]; /* > 0: add marks, < 0: remove marks */
        uint32_t rnd[4] = {0, 0, 0, 0};
        for (uint32_t c = 0; d != 0 && (c >> 2) < ORC_FIX_MAX_CALLS; ++c) {
            if (!(c & 3)) {
                uint32_t ctr[4] = {c >> 2, pi, t, ORC_DOM_FIX};
                orc_philox4x32_10(ctr, key, rnd);
            }
            uint64_t prod = (uint64_t)rnd[c & 3] * (uint64_t)N;
            if ((uint32_t)prod < reject_below) continue;
            int64_t pos = (int64_t)(prod >> 32);
            if (d > 0) {
                if (VALID(pos) && !(X[pos] & bit)) { X[pos] |= bit; --d; }
            } else if (X[pos] & bit) {
                X[pos] &= ~bit;
                ++d;
            }
        }
    }
    for (int64_t i = 0; i < N; ++i) {
        if (!VALID(i)) continue;
        uint32_t x = flip ? ~X[i] : X[i];
        while (x) {
            int b = __builtin_ctz(x);
            x &= x - 1;
            out[(int64_t)b * W + (i >> 6)] |= (uint64_t)1 << (i & 63);
        }
    }
#undef VALID
    free(X);
}

/* One permutation (index pi) of the block it belongs to. */
void orc_perm_labels(uint64_t seed, uint32_t t, uint32_t pi,
                     const uint64_t *mask, int64_t npos, int64_t N,
                     uint64_t *out)
{
    int64_t W = (N + 63) / 64;
    uint64_t *blk = (uint64_t *)malloc((size_t)(32 * W) * sizeof(uint64_t));
    orc_perm_block(seed, t, pi >> 5, mask, npos, N, blk);
    memcpy(out, blk + (int64_t)(pi & 31u) * W, (size_t)W * sizeof(uint64_t));
    free(blk);
}

/* ------------------------------------------------------------------ S5 -- */
/* r[g][t] = #{ pi < P : the permuted table is as or less probable than the
 * observed one }, i.e. w(a_pi) <= w(a_obs) * (1 + 1e-14) under the (gene, trait)
 * margins -- the rejection region of the two-sided Fisher test, so
 * "p_pi <= p_obs".  Empirical p = (r+1)/(P+1) (methods.py:1365).  Genes the
 * reference skips (all-absent / all-present among valid isolates,
 * methods.py:804-814) and degenerate traits get r = P.
 * perm_base: permutation indices are perm_base .. perm_base+P-1. */
void orc_permute_r(const uint64_t *genes, const uint64_t *traits,
                   const uint64_t *masks, int64_t G, int64_t T, int64_t N,
                   int64_t P, uint64_t seed, int64_t perm_base, uint32_t *r_out)
{
    enum { PB = 2048 };   /* 64 blocks of 32 permutations per batch (spec S4) */
    int64_t W = (N + 63) / 64;
    uint64_t *labs = (uint64_t *)malloc((size_t)(PB * W) * sizeof(uint64_t));
    memset(r_out, 0, (size_t)(G * T) * sizeof(uint32_t));
    for (int64_t t = 0; t < T; ++t) {
        const uint64_t *tr = traits + t * W, *mr = masks + t * W;
        int64_t npos = 0, nval = 0;
        for (int64_t w = 0; w < W; ++w) {
            npos += __builtin_popcountll(tr[w]);
            nval += __builtin_popcountll(mr[w]);
        }
        int64_t nneg = nval - npos;
        /* per gene: margin, observed a, and the exceedance table over a */
        int32_t *gm = (int32_t *)malloc((size_t)G * sizeof(int32_t));
        int32_t *aobs = (int32_t *)malloc((size_t)G * sizeof(int32_t));
        /* memo over gene margin: weights array + support offset */
        double **wtab = (double **)calloc((size_t)(nval + 2), sizeof(double *));
        int64_t *wlo = (int64_t *)calloc((size_t)(nval + 2), sizeof(int64_t));
        for (int64_t g = 0; g < G; ++g) {
            int32_t a = 0, m = 0;
            for (int64_t w = 0; w < W; ++w) {
                a += __builtin_popcountll(genes[g * W + w] & tr[w]);
                m += __builtin_popcountll(genes[g * W + w] & mr[w]);
            }
            gm[g] = m;
            aobs[g] = a;
            if (npos > 0 && nneg > 0 && m > 0 && m < nval && !wtab[m]) {
                int64_t cap = (m < npos ? m : npos) + 2, len;
                wtab[m] = (double *)malloc((size_t)cap * sizeof(double));
                wlo[m] = hg_weights(npos, nneg, m, wtab[m], &len);
            }
        }
        /* rejection region per gene: first point at or above the mode / last point at or
         * below it whose weight is <= w(a_obs) (1 + 1e-14) (hg_leq) */
        int32_t *regL = (int32_t *)malloc((size_t)G * sizeof(int32_t));
        int32_t *regH = (int32_t *)malloc((size_t)G * sizeof(int32_t));
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 64)
#endif
        for (int64_t g = 0; g < G; ++g) {
            int32_t m = gm[g];
            regL[g] = regH[g] = 0;
            if (!wtab[m]) continue;
            const double *w = wtab[m];
            int64_t lo = wlo[m], hi = m < npos ? m : npos, mode = lo;
            for (int64_t x = lo; x <= hi; ++x)
                if (w[x - lo] == 1.0) { mode = x; break; }
            int64_t H = hi + 1, L = lo - 1;
            for (int64_t x = mode; x <= hi; ++x)
                if (hg_leq(w, lo, npos, nneg, m, x, aobs[g])) { H = x; break; }
            for (int64_t x = mode; x >= lo; --x)
                if (hg_leq(w, lo, npos, nneg, m, x, aobs[g])) { L = x; break; }
            regL[g] = (int32_t)L;
            regH[g] = (int32_t)H;
        }
        for (int64_t p0 = 0; p0 < P; p0 += PB) {
            int64_t nb = P - p0 < PB ? P - p0 : PB;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
            for (int64_t jb = 0; jb < (nb + 31) / 32; ++jb) {
                /* block-wise when the batch is aligned to the 32-permutation blocks of spec S4 */
                int64_t first = perm_base + p0 + jb * 32, cnt = nb - jb * 32 < 32 ? nb - jb * 32 : 32;
                if (!(first & 31)) {
                    uint64_t *blk = (uint64_t *)malloc((size_t)(32 * W) * sizeof(uint64_t));
                    orc_perm_block(seed, (uint32_t)t, (uint32_t)(first >> 5), mr, npos, N, blk);
                    memcpy(labs + jb * 32 * W, blk, (size_t)(cnt * W) * sizeof(uint64_t));
                    free(blk);
                } else {
                    for (int64_t j = 0; j < cnt; ++j)
                        orc_perm_labels(seed, (uint32_t)t, (uint32_t)(first + j), mr, npos, N,
                                        labs + (jb * 32 + j) * W);
                }
            }
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
            for (int64_t g = 0; g < G; ++g) {
                const uint64_t *gr = genes + g * W;
                int32_t m = gm[g];
                uint32_t cnt = 0;
                if (!wtab[m]) {
                    cnt = (uint32_t)nb;
                } else {
                    /* the acceptance interval (Lb, Hb) of a under spec S3, then: in the
                     * region <=> a <= Lb or a >= Hb (weights are unimodal) */
                    for (int64_t j = 0; j < nb; ++j) {
                        const uint64_t *lr = labs + j * W;
                        int32_t a = 0;
                        for (int64_t k = 0; k < W; ++k)
                            a += __builtin_popcountll(gr[k] & lr[k]);
                        cnt += (a <= regL[g] || a >= regH[g]);
                    }
                }
                r_out[g * T + t] += cnt;
            }
        }
        for (int64_t m = 0; m < nval + 2; ++m)
            free(wtab[m]);
        free(wtab);
        free(wlo);
        free(gm);
        free(aobs);
        free(regL);
        free(regH);
    }
    free(labs);
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------ S6 -- */
/* Maximum number of contrasting pairs on a binary tree: restatement of
 * PhyloTree / Tip (scoary/classes.py:199-592).  State index: 0 = "AB",
 * 1 = "Ab", 2 = "aB", 3 = "ab", 4 = "0" (no free path).  Every node carries,
 * per state, (total pairs, supporting pairs, opposing pairs); -1 = the state
 * cannot be reached.
 *
 * The tree is given as a stack program over its tips (children order does not
 * matter: the candidate list of classes.py:320-405 is symmetric in left/right):
 *   ops[k] >= 0  : push Tip(tipstate[ops[k]])          (classes.py:575-592)
 *   ops[k] == -1 : pop right, pop left, push the merged node
 * out = (max_contrasting_pairs, max_contrasting_propairs,
 *        max_contrasting_antipairs) of the root (classes.py:246-249: three
 * independent maxima over the five states). */
typedef struct { int32_t tot[5], pro[5], anti[5]; } orc_node;

static void tree_merge(const orc_node *L, const orc_node *R, orc_node *out)
{
    for (int c = 0; c < 4; ++c) { /* calculate_max_condition, classes.py:268-457 */
        int ls[9], rs[9], n = 0;
        ls[n] = c; rs[n++] = 4;
        for (int o = 0; o < 4; ++o) if (o != c) { ls[n] = c; rs[n++] = o; }
        ls[n] = c; rs[n++] = c;
        ls[n] = 4; rs[n++] = c;
        for (int o = 0; o < 4; ++o) if (o != c) { ls[n] = o; rs[n++] = c; }
        int32_t tot[9], pro[9], anti[9], best = -1;
        for (int k = 0; k < 9; ++k) {
            tot[k] = pro[k] = anti[k] = -1;
            if (L->tot[ls[k]] > -1 && R->tot[rs[k]] > -1) {
                tot[k] = L->tot[ls[k]] + R->tot[rs[k]];
                pro[k] = L->pro[ls[k]] + R->pro[rs[k]];
                anti[k] = L->anti[ls[k]] + R->anti[rs[k]];
            }
            if (tot[k] > best) best = tot[k];
        }
        int32_t bp = -1, ba = -1;
        for (int k = 0; k < 9; ++k)
            if (tot[k] == best) {
                if (pro[k] > bp) bp = pro[k];
                if (anti[k] > ba) ba = anti[k];
            }
        out->tot[c] = best; out->pro[c] = bp; out->anti[c] = ba;
    }
    { /* calculate_max_nofree, classes.py:459-572 */
        static const int ls[5] = {4, 0, 3, 1, 2}, rs[5] = {4, 3, 0, 2, 1};
        static const int dpro[5] = {0, 1, 1, 0, 0}, danti[5] = {0, 0, 0, 1, 1};
        int32_t tot[5], pro[5], anti[5], best = -1;
        for (int k = 0; k < 5; ++k) {
            tot[k] = pro[k] = anti[k] = -1;
            if (L->tot[ls[k]] > -1 && R->tot[rs[k]] > -1) {
                tot[k] = L->tot[ls[k]] + R->tot[rs[k]] + (k > 0);
                pro[k] = L->pro[ls[k]] + R->pro[rs[k]] + dpro[k];
                anti[k] = L->anti[ls[k]] + R->anti[rs[k]] + danti[k];
            }
            if (tot[k] > best) best = tot[k];
        }
        int32_t bp = -1, ba = -1;
        for (int k = 0; k < 5; ++k)
            if (tot[k] == best) {
                if (pro[k] > bp) bp = pro[k];
                if (anti[k] > ba) ba = anti[k];
            }
        out->tot[4] = best; out->pro[4] = bp; out->anti[4] = ba;
    }
}

int orc_tree_dp(const int32_t *ops, int64_t nops, const uint8_t *tipstate,
                int32_t out[3])
{
    orc_node *st = (orc_node *)malloc((size_t)(nops + 1) * sizeof(orc_node));
    int64_t sp = 0;
    for (int64_t k = 0; k < nops; ++k) {
        if (ops[k] >= 0) {
            orc_node *t = &st[sp++];
            for (int c = 0; c < 5; ++c)
                t->tot[c] = t->pro[c] = t->anti[c] = (c == tipstate[ops[k]]) ? 0 : -1;
        } else {
            if (sp < 2) { free(st); return -1; }
            orc_node m;
            tree_merge(&st[sp - 2], &st[sp - 1], &m);
            st[sp - 2] = m;
            --sp;
        }
    }
    if (sp != 1) { free(st); return -1; }
    out[0] = out[1] = out[2] = -1;
    for (int c = 0; c < 5; ++c) {
        if (st[0].tot[c] > out[0]) out[0] = st[0].tot[c];
        if (st[0].pro[c] > out[1]) out[1] = st[0].pro[c];
        if (st[0].anti[c] > out[2]) out[2] = st[0].anti[c];
    }
    free(st);
    return 0;
}

/* Tree-statistic permutations (Permute, scoary/methods.py:1314-1369) for one
 * gene: gbits / the trait's label+validity bits are rows64 over isolates;
 * tips[k] = isolate index of tip k of the (pruned) tree program.  For each
 * permutation pi (labels by spec S4) writes exceed[pi] = 1 iff
 * float(New[key])/New["Total"] >= observed estimator, key = Pro if the
 * observed tree has Pro >= Anti else Anti (methods.py:1333-1355).  A permuted
 * tree with Total == 0 (ZeroDivisionError in the reference) counts as 0.
 * out_obs = observed (Total, Pro, Anti). */
int orc_tree_permute(const int32_t *ops, int64_t nops, const int32_t *tips,
                     int64_t ntips, const uint64_t *gbits, const uint64_t *tbits,
                     const uint64_t *mbits, int64_t N, uint32_t t, int64_t P,
                     uint64_t seed, int32_t out_obs[3], uint8_t *exceed)
{
    int64_t W = (N + 63) / 64, npos = 0;
    for (int64_t w = 0; w < W; ++w) npos += __builtin_popcountll(tbits[w]);
    uint8_t *ts = (uint8_t *)malloc((size_t)ntips);
    uint64_t *lab = (uint64_t *)malloc((size_t)W * sizeof(uint64_t));
#define BIT(a, i) (((a)[(i) >> 6] >> ((i) & 63)) & 1)
    for (int64_t k = 0; k < ntips; ++k) {
        int64_t i = tips[k];
        ts[k] = (uint8_t)((BIT(gbits, i) ? 0 : 2) + (BIT(tbits, i) ? 0 : 1));
    }
    if (orc_tree_dp(ops, nops, ts, out_obs)) { free(ts); free(lab); return -1; }
    int key = out_obs[1] >= out_obs[2] ? 1 : 2;
    double est = (double)out_obs[key] / (double)out_obs[0];
    uint64_t *blk = (uint64_t *)malloc((size_t)(32 * W) * sizeof(uint64_t));
    for (int64_t pi = 0; pi < P; ++pi) {
        if (!(pi & 31)) orc_perm_block(seed, t, (uint32_t)(pi >> 5), mbits, npos, N, blk);
        memcpy(lab, blk + (pi & 31) * W, (size_t)W * sizeof(uint64_t));
        for (int64_t k = 0; k < ntips; ++k) {
            int64_t i = tips[k];
            ts[k] = (uint8_t)((BIT(gbits, i) ? 0 : 2) + (BIT(lab, i) ? 0 : 1));
        }
        int32_t o[3];
        orc_tree_dp(ops, nops, ts, o);
        exceed[pi] = (o[0] > 0 && (double)o[key] / (double)o[0] >= est) ? 1 : 0;
    }
#undef BIT
    free(ts);
    free(lab);
    free(blk);
    return 0;
}
