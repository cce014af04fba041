// scoary_scipy.hip -- scoary_fisher_scipy: the two-sided Fisher p of a 2x2 table AS scipy.stats.fisher_exact
// PRINTS IT, to the last bit, for tables of 171 ... 104 723 isolates.  (Up to 170 isolates k_fisher already returns
// SciPy's double, scoary_assoc.hip; this kernel leaves such tables alone.)
//
// Why it exists: above 170 isolates k_fisher's p is the exact value of SciPy's rule to ~3e-15 -- inside the path's
// 1e-12, but not the digits the reference writes into its result files.  The command line runs this pass over the
// tables it is going to print, so that its CSVs are the reference's bytes at any realistic size.  It is NOT part of
// the benchmarked step (bench.py reports its time separately): ~50x the work of k_fisher per table.
//
// What SciPy 1.15.3 computes (scipy/stats/_stats_py.py fisher_exact -> hypergeom.pmf / cdf / sf -> Boost.Math 1.83):
//   * pmf: beyond its table of factorials Boost evaluates the quotient of factorials through its PRIME FACTORISATION
//     (boost/math/distributions/detail/hypergeometric_pdf.hpp, hypergeometric_pdf_prime_loop_imp): for every prime
//     q <= N the exponent of q by Legendre's formula, the running product multiplied by q^e in ascending order of q --
//     a partial product that would overflow or underflow is set aside and a new one started -- and at the end the
//     partial products multiplied together, one >= 1 while the running value is <= 1, one < 1 otherwise.  q^|e| is an
//     exact double for every exponent that can occur, so only the order of the multiplications matters; a negative
//     exponent is 1 / q^|e|, one rounding.
//   * cdf / sf: the pmf at the point next to x on the side of the nearer end of the support, then the term
//     recurrence towards that end until a term no longer counts; the far side as 1 - sum.
//   * fisher_exact: pmf at the observed table and at the mode (equal to 1e-14: p = 1), a binary search over the pmf
//     for the point on the other side of the mode, the two tails added, min(p, 1).
// Restated operation by operation in plain fp64 (-ffp-contract=off); the CPU checker of the tests has its own
// restatement, pinned against SciPy bit for bit, and this kernel is held to the checker
// (tests/test_gpu_parity.py) and to the SciPy of the GPU box directly.
#include "scoary_common.hpp"

namespace {

constexpr int kScipyMinN = 171;          // below: k_fisher's own SciPy path (factorial table)
constexpr int kScipyMaxN = 104723;       // Boost's prime table ends at the 10 000th prime, 104 729
constexpr int kMaxPrimes = 10000;
constexpr int kMaxParts = 192;           // partial products of one pmf (127 seen at N = 104 723)
constexpr double kDblMax = 1.7976931348623157e308, kDblMin = 2.2250738585072014e-308;
constexpr double kEps = 2.220446049250313e-16;
constexpr double kGammaS = 1.0 + 1e-14;  // fisher_exact: gamma = 1 + epsilon

struct Primes { const uint32_t* q; const float* inv; int n; };   // inv[k] = 1.0f / q[k]

// floor(x / base) in fp32 for 0 <= x <= 104 723, 1 <= base <= 104 729: xh = x + 0.5 (exact in fp32), inv = 1 / base to
// one ulp.  (x + 0.5) / base lies at least 0.5 / base away from every integer and the product's error is below
// (x / base) 2^-22 < 0.5 / base (x base < 2^21 ... x < 2^21), so truncation gives the quotient.  fp32 because
// v_cvt_i32_f64 runs at a quarter of the rate: the nine quotients per prime are the kernel's inner loop.
__device__ __forceinline__ int fdiv(float xh, float inv) { return (int)(xh * inv); }

// Boost's prime-factorised pmf of x successes in n draws, r successes among N items.  `part` = this lane's scratch.
// Returns a negative value if more than kMaxParts partial products were needed (the caller leaves the table alone).
__device__ __noinline__ double pdf_prime(int x, int r, int n, int N, Primes pr, double* part) {
  int np = 1;
  part[0] = 1.0;
  const float t0 = (float)n + 0.5f, t1 = (float)r + 0.5f, t2 = (float)(N - n) + 0.5f, t3 = (float)(N - r) + 0.5f,
              u0 = (float)N + 0.5f, u1 = (float)x + 0.5f, u2 = (float)(n - x) + 0.5f, u3 = (float)(r - x) + 0.5f,
              u4 = (float)(N - n - r + x) + 0.5f;
  for (int k = 0; k < pr.n; ++k) {
    const int q = (int)pr.q[k];
    if (q > N) break;
    int e = 0;
    float inv = pr.inv[k];
    for (int64_t base = q; base <= N; base *= q) {
      if (base != q) inv = 1.0f / (float)base;
      e += fdiv(t0, inv) + fdiv(t1, inv) + fdiv(t2, inv) + fdiv(t3, inv);
      e -= fdiv(u0, inv) + fdiv(u1, inv) + fdiv(u2, inv) + fdiv(u3, inv) + fdiv(u4, inv);
    }
    if (e == 0) continue;
    double v = 1.0;
    const double dq = (double)q;
    for (int i = e < 0 ? -e : e; i > 0; --i) v *= dq;      // exact
    if (e < 0) v = 1.0 / v;
    const double cur = part[np - 1];
    // Boost's tests "would the product overflow / underflow" each cost a division; with both factors far from the
    // ends of the range neither can be true (max / 1e100 > 1e200, min / 1e-100 < 1e-200) and they are not evaluated
    const bool far = cur < 1e200 && cur > 1e-200 && v < 1e100 && v > 1e-100;
    if (!far && ((v > 1.0 && kDblMax / v < cur) || (v < 1.0 && kDblMin / v > cur))) {
      if (np == kMaxParts) return -1.0;
      part[np++] = v;
      continue;
    }
    part[np - 1] = cur * v;
  }
  // newest first, as Boost walks its list: i over the entries >= 1, j over those < 1
  int i = np - 1, j = np - 1;
  while (i >= 0 && part[i] < 1.0) --i;
  while (j >= 0 && part[j] >= 1.0) --j;
  double prod = 1.0;
  while (i >= 0 || j >= 0) {
    while (i >= 0 && (prod <= 1.0 || j < 0)) {
      prod *= part[i--];
      while (i >= 0 && part[i] < 1.0) --i;
    }
    while (j >= 0 && (prod >= 1.0 || i < 0)) {
      prod *= part[j--];
      while (j >= 0 && part[j] >= 1.0) --j;
    }
  }
  return prod;
}

struct Hg {             // scipy.stats.hypergeom(M, good, draws)
  int M, good, draws, lo, hi;
  Primes pr;
  double* part;
  bool bad;             // a pmf ran out of partial-product slots
  __device__ Hg(int M_, int good_, int draws_, Primes pr_, double* part_)
      : M(M_), good(good_), draws(draws_), lo(max(draws_ - (M_ - good_), 0)), hi(min(good_, draws_)), pr(pr_),
        part(part_), bad(false) {}
  __device__ static double clip(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
  __device__ double raw_pdf(int k) {
    const double v = pdf_prime(k, good, draws, M, pr, part);
    if (v < 0.0) bad = true;
    return v;
  }
  __device__ double pmf(int k) { return (k < lo || k > hi) ? 0.0 : clip(raw_pdf(k)); }
  // Boost's lower tail P(X <= x) (upper == false) or upper tail P(X > x)
  __device__ double tail(int x, bool upper) {
    const int r = good, n = draws, N = M;
    const double mode = floor((double)(r + 1) * (double)(n + 1) / (double)(N + 2));
    double sum = 0.0;
    bool invert = upper;
    if ((double)x < mode) {
      sum = raw_pdf(x);
      double term = sum;
      const int lower = max(0, n + r - N);
      while (term > (invert ? 1.0 : sum) * kEps) {
        term = (double)x * (double)(N + x - n - r) * term / ((double)(1 + n - x) * (double)(1 + r - x));
        sum += term;
        if (x == lower) break;
        --x;
      }
    } else {
      invert = !invert;
      const int top = min(r, n);
      if (x != top) {
        ++x;
        sum = raw_pdf(x);
        double term = sum;
        while (x <= top && term > (invert ? 1.0 : sum) * kEps) {
          term = (double)(n - x) * (double)(r - x) * term / ((double)(x + 1) * (double)(N + x + 1 - n - r));
          sum += term;
          ++x;
        }
      }
    }
    return invert ? 1.0 - sum : sum;
  }
  __device__ double cdf(int k) { return k < lo ? 0.0 : (k >= hi ? 1.0 : clip(tail(k, false))); }
  __device__ double sf(int k) { return k < lo ? 1.0 : (k >= hi ? 0.0 : clip(tail(k, true))); }
};

__device__ double fisher_two_sided(int a, int b, int c, int d, Primes pr, double* part, bool* bad) {
  const int n1 = a + b, n2 = c + d, n = a + c;
  Hg h(n1 + n2, n1, n, pr, part);
  const int mode = (int)(((double)(n + 1) * (double)(n1 + 1)) / (double)(n1 + n2 + 2));
  const double pexact = h.pmf(a), pmode = h.pmf(mode);
  double p;
  if (fabs(pexact - pmode) / fmax(pexact, pmode) <= 1e-14) {
    p = 1.0;
  } else {
    const double target = pexact * kGammaS;
    const bool below = a < mode;
    if (below ? h.pmf(n) > target : h.pmf(0) > target) {
      p = below ? h.cdf(a) : h.sf(a - 1);
    } else {
      // _binary_search(f, target, lo, hi) on f = -pmf (a < mode: the descending side) or pmf (ascending side)
      const double sign = below ? -1.0 : 1.0, want = sign * target;
      int lo = below ? mode : 0, hi = below ? n : mode, guess = 0;
      bool hit = false;
      while (lo < hi) {
        const int mid = lo + (hi - lo) / 2;
        const double f = sign * h.pmf(mid);
        if (f < want) {
          lo = mid + 1;
        } else if (f > want) {
          hi = mid - 1;
        } else {
          guess = mid;
          hit = true;
          break;
        }
      }
      if (!hit) guess = sign * h.pmf(lo) <= want ? lo : lo - 1;
      p = below ? h.cdf(a) + h.sf(guess) : h.sf(a - 1) + h.cdf(guess);
      p = p < 1.0 ? p : 1.0;
    }
  }
  *bad = h.bad;
  return p;
}

// One lane per table.  Tables with an empty margin or outside [kScipyMinN, kScipyMaxN] are left as they are.
__global__ __launch_bounds__(64) void k_fisher_scipy(const int4* __restrict__ tables, int64_t M,
                                                     double* __restrict__ p_out, const uint32_t* __restrict__ primes,
                                                     const float* __restrict__ inv, int nprimes,
                                                     unsigned long long* __restrict__ skipped) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= M) return;
  const int4 t = tables[i];
  const int a = t.x, b = t.y, c = t.z, d = t.w;
  if (a + b == 0 || c + d == 0 || a + c == 0 || b + d == 0) return;
  const int N = a + b + c + d;
  if (N < kScipyMinN) return;
  if (N > kScipyMaxN) {
    if (skipped) atomicAdd(skipped, 1ull);
    return;
  }
  double part[kMaxParts];
  bool bad = false;
  const double p = fisher_two_sided(a, b, c, d, Primes{primes, inv, nprimes}, part, &bad);
  if (bad) {
    if (skipped) atomicAdd(skipped, 1ull);
    return;
  }
  p_out[i] = p;
}

// the primes up to 104 729, once per device (40 KB; lives until the process ends)
constexpr int kMaxDevices = 64;
uint32_t* g_primes[kMaxDevices] = {};
float* g_inv[kMaxDevices] = {};
int g_nprimes = 0;

int primes_on_device(scoary_handle h, const uint32_t** out, const float** inv_out, int* n) {
  if (h->device < 0 || h->device >= kMaxDevices) return fail(h, SCOARY_ERR_ARG, "scoary_fisher_scipy: device index");
  if (!g_primes[h->device]) {
    const int top = 104730;
    std::vector<uint8_t> sieve((size_t)top + 1, 0);
    std::vector<uint32_t> pr;
    pr.reserve(kMaxPrimes);
    for (int i = 2; i <= top; ++i) {
      if (sieve[i]) continue;
      pr.push_back((uint32_t)i);
      for (int64_t j = (int64_t)i * i; j <= top; j += i) sieve[(size_t)j] = 1;
    }
    std::vector<float> inv(pr.size());
    for (size_t i = 0; i < pr.size(); ++i) inv[i] = 1.0f / (float)pr[i];
    uint32_t* dev = nullptr;
    float* dinv = nullptr;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&dev), pr.size() * sizeof(uint32_t)));
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&dinv), pr.size() * sizeof(float)));
    HIP_TRY(h, hipMemcpy(dev, pr.data(), pr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(h, hipMemcpy(dinv, inv.data(), inv.size() * sizeof(float), hipMemcpyHostToDevice));
    g_nprimes = (int)pr.size();
    g_inv[h->device] = dinv;
    g_primes[h->device] = dev;
  }
  *out = g_primes[h->device];
  *inv_out = g_inv[h->device];
  *n = g_nprimes;
  return SCOARY_OK;
}

}  // namespace

extern "C" {

int64_t scoary_fisher_scipy_max_isolates(void) { return kScipyMaxN; }

int scoary_fisher_scipy(scoary_handle h, const int32_t* d_tables, int64_t M, double* d_p,
                        uint64_t* d_skipped, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tables || !d_p || M < 1) return fail(h, SCOARY_ERR_ARG, "scoary_fisher_scipy: bad argument");
  if ((M + kWave - 1) / kWave > 0x7fffffffLL) return fail(h, SCOARY_ERR_SIZE, "scoary_fisher_scipy: M too large");
  DeviceGuard guard(h->device);
  const uint32_t* primes = nullptr;
  const float* inv = nullptr;
  int nprimes = 0;
  const int rc = primes_on_device(h, &primes, &inv, &nprimes);
  if (rc != SCOARY_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_fisher_scipy");
  hipLaunchKernelGGL(k_fisher_scipy, dim3((unsigned)((M + kWave - 1) / kWave)), dim3(kWave), 0, s,
                     reinterpret_cast<const int4*>(d_tables), M, d_p, primes, inv, nprimes,
                     reinterpret_cast<unsigned long long*>(d_skipped));
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

}  // extern "C"
