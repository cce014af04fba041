// scoary_assoc.hip -- gfx950 (MI355X / CDNA4) kernels + C-ABI for Scoary's
// association hot path.  See include/scoary_hip.h for the contract and
// DESIGN.md for layouts, byte models and the specs S1-S5 shared with the CPU
// oracle.
//
// Design in one paragraph: the gene presence/absence matrix lives in HBM as
// 32-bit words in "word-quad-major" order, so ONE LANE OWNS ONE GENE: a
// wavefront's gene loads are 1 KiB coalesced dwordx4, while the other operand
// of every AND -- a trait / mask / permuted-label word -- is wave-uniform and
// is fetched through the SCALAR cache into SGPRs.  The inner loop is therefore
// exactly two VALU ops per 32 isolates per (gene, vector) pair:
//     v_and_b32  tmp, s_vec, v_gene ;  v_bcnt_u32_b32  acc, tmp, acc
// with no LDS traffic and no cross-lane reduction.  There is no MFMA: CDNA4
// has no AND-popcount matrix mode, and this is integer/bit work.
#include "scoary_common.hpp"

namespace {

// ----------------------------------------------------------------------------
// a1: packing
// ----------------------------------------------------------------------------
// One thread per (gene, 32-bit word): 32 presence bytes -> one word.
__global__ __launch_bounds__(256) void k_pack_dense(const uint8_t* __restrict__ dense,
                                                    int64_t G, int64_t N, int64_t Gp,
                                                    int64_t Qp, uint32_t* __restrict__ tiled) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t k = blockIdx.y;  // 32-bit word index, < 4*Qp
  if (g >= Gp) return;
  uint32_t word = 0;
  if (g < G) {
    const int64_t i0 = k * 32;
    const uint8_t* row = dense + g * N;
    for (int b = 0; b < 32; ++b) {
      const int64_t i = i0 + b;
      if (i < N && row[i] != 0) word |= 1u << b;
    }
  }
  tiled[((k >> 2) * Gp + g) * 4 + (k & 3)] = word;
}

// One thread per (gene, quad): 16 bytes of a row-major bit row -> its tile slot.
__global__ __launch_bounds__(256) void k_tile_rows(const uint32_t* __restrict__ rows32,
                                                   int64_t G, int64_t W32, int64_t Gp,
                                                   uint4* __restrict__ tiled) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t q = blockIdx.y;
  if (g >= Gp) return;
  uint32_t w[4] = {0, 0, 0, 0};
  if (g < G) {
    const uint32_t* row = rows32 + g * W32;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (q * 4 + j < W32) w[j] = row[q * 4 + j];
  }
  tiled[q * Gp + g] = make_uint4(w[0], w[1], w[2], w[3]);
}

// ----------------------------------------------------------------------------
// a3: contingency counts
// ----------------------------------------------------------------------------
__device__ __forceinline__ int popc4(const uint4 a, const uint4 b) {
  return __popc(a.x & b.x) + __popc(a.y & b.y) + __popc(a.z & b.z) + __popc(a.w & b.w);
}

__device__ __forceinline__ int popc4(const uint4 a) {
  return __popc(a.x) + __popc(a.y) + __popc(a.z) + __popc(a.w);
}

// Trait plan: everything the counts need that depends on the traits alone -- built once per trait
// set (like the index lists once per gene matrix), not once per step.
//   margins[t] = (positives, valid isolates) = popcounts of the label / validity rows;
//   cls[t]     = the smallest t' <= t whose validity row equals trait t's: traits of one class
//                share popc(gene & valid), which k_counts computes once per class and pass (most
//                traits of a real file have no missing values at all: one class);
//   per pass of TB traits (counts_traits_per_pass): the classes of the pass as dense slots, and
//   the pass's label and validity rows GATHERED quad-major -- vecq[pass][q][2 TB] 16-byte quads,
//   slots 0 .. TB - 1 the label rows (the last one repeated past the pass's traits), TB .. the
//   validity rows of the pass's classes (the last one repeated) -- so that k_counts fetches the
//   operands of one gene quad with a few wide scalar loads from one contiguous 32 TB-byte row.
// Plan buffer (int32 words): header {T, TB, passes, Qp}, cls[T], slot[T], nM[passes],
// mrow[passes][TB], padding to 64 bytes, vecq.
struct PlanLayout {
  int64_t cls, slot, nm, mrow, vecq_words, words;
  int tb, passes;
};
constexpr int kPlanHdr = 4;
__host__ __device__ inline int plan_traits_per_pass(int64_t T) {
  const int64_t passes = (T + 31) / 32;
  const int64_t tpp = (T + passes - 1) / passes;
  return tpp <= 2 ? (int)tpp : (int)((tpp + 3) / 4 * 4);    // kernel instances: 1, 2, 4, 8, ... 32
}
__host__ __device__ inline PlanLayout plan_layout(int64_t T, int64_t Qp) {
  PlanLayout L{};
  L.tb = plan_traits_per_pass(T);
  L.passes = (int)((T + L.tb - 1) / L.tb);
  L.cls = kPlanHdr;
  L.slot = L.cls + T;
  L.nm = L.slot + T;
  L.mrow = L.nm + L.passes;
  L.vecq_words = (L.mrow + (int64_t)L.passes * L.tb + 15) / 16 * 16;
  L.words = L.vecq_words + (int64_t)L.passes * Qp * 2 * L.tb * 4;
  return L;
}
// One wavefront per trait; the class search compares rows front to back with an early exit
// (different masks differ within the first words, identical ones are found at t' = cls).  Classes
// are shared WITHIN a pass of tb <= 32 traits only -- that is all k_counts can use -- so the search
// looks at the traits of the own pass: at most 31 row compares per trait, whatever T is (it ran
// over all earlier traits before round 5: O(T^2) row compares for T distinct masks, T <= 524 280).
__global__ __launch_bounds__(64) void k_trait_plan(const uint32_t* __restrict__ traits,
                                                   const uint32_t* __restrict__ masks, int Wp, int T, int tb,
                                                   int32_t* __restrict__ margins,
                                                   int32_t* __restrict__ cls,
                                                   int32_t* __restrict__ cls_user) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const uint32_t* trow = traits + (int64_t)t * Wp;
  const uint32_t* mrow = masks + (int64_t)t * Wp;
  int npos = 0, nval = 0;
  for (int w = lane; w < Wp; w += 64) {
    npos += __popc(trow[w]);
    nval += __popc(mrow[w]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    npos += __shfl_xor(npos, off);
    nval += __shfl_xor(nval, off);
  }
  if (lane == 0) {
    margins[2 * t] = npos;
    margins[2 * t + 1] = nval;
  }
  int found = t;
  for (int u = t / tb * tb; u < t; ++u) {
    const uint32_t* urow = masks + (int64_t)u * Wp;
    bool same = true;
    for (int w0 = 0; w0 < Wp && same; w0 += 64) {
      const int w = w0 + lane;
      const bool eq = w >= Wp || urow[w] == mrow[w];
      same = __all(eq);
    }
    if (same) {
      found = u;
      break;
    }
  }
  if (lane == 0) {
    cls[t] = found;
    if (cls_user) cls_user[t] = found;
  }
}
// One wavefront per pass: the classes of the pass's traits -> dense slots 0 .. nM - 1
__global__ __launch_bounds__(64) void k_trait_slots(int32_t* __restrict__ plan, int T, int Qp) {
  const PlanLayout L = plan_layout(T, Qp);
  const int pass = blockIdx.x, lane = threadIdx.x, t0 = pass * L.tb;
  const int nT = min(L.tb, T - t0);
  const int32_t* cls = plan + L.cls;
  int rep = lane;                                         // first trait of the pass with this trait's class
  if (lane < nT)
    for (int u = 0; u < lane; ++u)
      if (cls[t0 + u] == cls[t0 + lane]) {
        rep = u;
        break;
      }
  const uint64_t owners = __ballot(lane < nT && rep == lane);
  if (lane < nT) plan[L.slot + t0 + lane] = __popcll(owners & (((uint64_t)1 << rep) - 1));
  if (lane < nT && rep == lane)
    plan[L.mrow + (int64_t)pass * L.tb + __popcll(owners & (((uint64_t)1 << lane) - 1))] = lane;
  if (lane == 0) {
    plan[L.nm + pass] = __popcll(owners);
    if (pass == 0) {
      plan[0] = T;
      plan[1] = L.tb;
      plan[2] = L.passes;
      plan[3] = Qp;
    }
  }
}
// block (q, pass), thread = slot: gather the pass's rows quad-major
__global__ __launch_bounds__(64) void k_trait_vecq(const uint4* __restrict__ traits,
                                                   const uint4* __restrict__ masks,
                                                   int32_t* __restrict__ plan, int T, int Qp) {
  const PlanLayout L = plan_layout(T, Qp);
  const int q = blockIdx.x, pass = blockIdx.y, sl = threadIdx.x, t0 = pass * L.tb;
  if (sl >= 2 * L.tb) return;
  const int nT = min(L.tb, T - t0), nM = plan[L.nm + pass];
  uint4 v;
  if (sl < L.tb) {
    v = traits[(int64_t)(t0 + min(sl, nT - 1)) * Qp + q];
  } else {
    const int row = plan[L.mrow + (int64_t)pass * L.tb + min(sl - L.tb, nM - 1)];
    v = masks[(int64_t)(t0 + row) * Qp + q];
  }
  reinterpret_cast<uint4*>(plan + L.vecq_words)[((int64_t)pass * Qp + q) * (2 * L.tb) + sl] = v;
}

// K1.  Block = 64 genes x QS wavefronts; lane = gene, wavefront w = the quads q = w, w + QS, ... of
// the rows: the matrix is streamed ONCE per pass, 1 KiB per wavefront load, and QS (1 ... 16, chosen by
// the host) makes enough wavefronts to hide the latency of a stream that has little arithmetic per
// byte (round 3's lane-per-gene kernel took 4 traits per pass: ceil(T / 4) passes over the matrix,
// 13 at cfg5, 8.5x the algorithmic bytes).  A pass takes TB traits (instances 1, 2, 4, 8, ... 32;
// grid.y passes; TB <= 32 keeps 2 TB accumulators in VGPRs):
//   a[j] += popc(gene & label_j)      for every trait slot j of the pass
//   m[i] += popc(gene & valid_i)      for the i-th CLASS of identical validity rows of the pass,
//                                     four classes per (wave-uniform) branch
// The operands of a gene quad are one contiguous row of the plan's vecq (scalar loads of 64 bytes,
// the next group requested before the current one is used; SGPR operands of the AND).  The partial
// sums of a gene's quad slices meet in LDS (ds_add), where the tables are put together (a of trait
// j, m of its class, the margins from the plan) and leave as one 16-byte store per (trait, gene).
template <int TB>
__global__ __launch_bounds__(1024) void k_counts(const uint4* __restrict__ tiled,
                                                 const int32_t* __restrict__ plan,
                                                 const int32_t* __restrict__ margins, int G, int Gp,
                                                 int Qp, int T, int4* __restrict__ counts) {
  constexpr int GS = TB < 4 ? TB : 4;                     // vectors per scalar load / mask branch
  constexpr int NGR = TB / GS;                            // groups of trait slots
  __shared__ int32_t s_tot[2 * TB][kWave];               // 512 B per accumulator: 16 KB at TB = 32
  const PlanLayout L = plan_layout(T, Qp);
  const int tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = blockIdx.x * kWave + lane;               // < Gp: the tiled matrix is padded
  const int pass = blockIdx.y, t0 = pass * TB;
  const int nT = min(TB, T - t0);                         // traits of this pass, 1 ... TB
  const int nM = __builtin_amdgcn_readfirstlane(plan[L.nm + pass]);     // classes of the pass, >= 1
  for (int i = tid; i < 2 * TB * kWave; i += blockDim.x) (&s_tot[0][0])[i] = 0;
  __syncthreads();
  uint32_t a[TB], m[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j) a[j] = m[j] = 0u;
  const uint4* vecq = reinterpret_cast<const uint4*>(plan + L.vecq_words) + (int64_t)pass * Qp * (2 * TB);
  auto load_group = [&](const uint4* row, int gi, uint4 (&dst)[GS]) {
#pragma unroll
    for (int k = 0; k < GS; ++k) dst[k] = row[gi * GS + k];          // wave-uniform: one s_load_dwordx(4 GS)
  };
  auto add_group = [&](const uint4& gw, const uint4 (&src)[GS], uint32_t* acc) {
#pragma unroll
    for (int k = 0; k < GS; ++k) {
      bcnt_acc(acc[k], gw.x & src[k].x);
      bcnt_acc(acc[k], gw.y & src[k].y);
      bcnt_acc(acc[k], gw.z & src[k].z);
      bcnt_acc(acc[k], gw.w & src[k].w);
    }
  };
  // U gene quads are requested before any of them is used: with few traits per pass there is
  // little arithmetic per load and the stream lives on loads in flight (cfg4: T = 1)
  // (round 6, cold stream on 125 000 x 10 000: two quads in flight per wavefront beat four at T = 4 -- 34.2 against
  // 35.1 us -- and are within 1 % at T = 1 / 2; eight, or a double buffer, are slower: profiles/r06_k1_*.txt)
  constexpr int U = TB <= 16 ? 2 : 1;
  for (int q0 = wave; q0 < Qp; q0 += nw * U) {
    uint4 gws[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = q0 + nw * u;                          // wave-uniform
      gws[u] = q < Qp ? tiled[(int64_t)q * Gp + g] : make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int q = min(q0 + nw * u, Qp - 1);             // past the end: a zero gene quad adds nothing
      const uint4 gw = gws[u];
      const uint4* row = vecq + (int64_t)q * (2 * TB);
      uint4 cur[GS], nxt[GS];
      load_group(row, 0, cur);
#pragma unroll
      for (int gi = 0; gi < NGR; ++gi) {                  // label rows; then the first group of classes
        load_group(row, gi + 1, nxt);                     // gi + 1 == NGR: slots TB .. = validity rows
        add_group(gw, cur, a + gi * GS);
#pragma unroll
        for (int k = 0; k < GS; ++k) cur[k] = nxt[k];
      }
      add_group(gw, cur, m);                              // classes 0 .. GS - 1 (nM >= 1; repeats are not read)
#pragma unroll
      for (int gi = 1; gi < NGR; ++gi) {                  // further classes: rare (many distinct masks)
        if (gi * GS < nM) {                               // wave-uniform
          load_group(row, NGR + gi, cur);
          add_group(gw, cur, m + gi * GS);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < TB; ++j) {                          // the quad slices of a gene meet in LDS
    if (j < nT) atomicAdd(&s_tot[j][lane], (int32_t)a[j]);
    if (j < nM) atomicAdd(&s_tot[TB + j][lane], (int32_t)m[j]);
  }
  __syncthreads();
  if (g >= G) return;
  for (int j = wave; j < nT; j += nw) {                   // wavefront w puts together traits w, w + nw, ...
    const int aa = s_tot[j][lane];
    const int mm = s_tot[TB + plan[L.slot + t0 + j]][lane];
    const int npos = margins[2 * (t0 + j)], nval = margins[2 * (t0 + j) + 1];
    counts[(int64_t)(t0 + j) * G + g] = make_int4(aa, npos - aa, mm - aa, nval - npos - mm + aa);
  }
}

// ----------------------------------------------------------------------------
// a5: two-sided Fisher exact test (spec S3)
// ----------------------------------------------------------------------------
// Hypergeometric weights by the exact ratio recurrence, normalised at the
// mode; same operation order as the oracle's hg_weights so the weights (and
// hence the rejection regions) are bit-identical on both sides.
// One step: w(x + 1) = w(x) * ((n1 - x)(n - x)) / ((x + 1)(n2 - n + x + 1)), the four factors
// kept as doubles that move by +-1 (exact; no int -> double conversions in the loop).
struct HgWalk {
  double fa, fb, fc, fd;
  __device__ __forceinline__ HgWalk(int x, int n1, int n2, int n)
      : fa((double)(n1 - x)), fb((double)(n - x)), fc((double)(x + 1)), fd((double)(n2 - n + x + 1)) {}
  __device__ __forceinline__ double step(double w) {
    const double r = w * (fa * fb) / (fc * fd);
    fa -= 1.0; fb -= 1.0; fc += 1.0; fd += 1.0;
    return r;
  }
};

// ---- spec S3 tie rule: w(x) <= w(a) * (1 + 1e-14), SciPy's gamma ------------------------
// The fp64 recurrence carries ~2e-16 of error per step, so a comparison that comes out within
// kAmbig of equality is redone in double-double arithmetic (~1e-31 per step): the ratio
// w(x)/w(a) is the product of the recurrence's step ratios between the two points; they sit on
// opposite sides of the mode, so the product is taken from both ends -- the next factor > 1
// while the running value is <= 1, the next factor < 1 otherwise -- and never leaves the
// range of the factors.  Exact ties (symmetric margins) come out as 1 to ~1e-28 and are ties;
// the closest non-equal weights there are (1.7e-12 apart at N = 1972; 6.1e-12 at N = 10 000,
// tests/golden/near_ties.json) are told apart.  The oracle settles the same comparisons on
// big integers.
struct DD { double hi, lo; };
__device__ __forceinline__ DD dd_mul_d(DD a, double b) {
  const double p = a.hi * b;
  double e = fma(a.hi, b, -p);          // a.hi * b = p + e exactly
  e += a.lo * b;
  const double s = p + e;
  return {s, e - (s - p)};
}
__device__ __forceinline__ DD dd_div_d(DD a, double b) {
  const double q1 = a.hi / b;
  const double p = q1 * b;
  const double e = fma(q1, b, -p);      // q1 * b = p + e exactly
  const double r = ((a.hi - p) - e) + a.lo;
  const double q2 = r / b;
  const double s = q1 + q2;
  return {s, q2 - (s - q1)};
}
constexpr double kAmbig = 1e-9;
constexpr double kGamma = 1.0 + 1e-14;  // scipy/stats/_stats_py.py fisher_exact: gamma = 1 + epsilon
__device__ __noinline__ bool hg_leq_dd(int n1, int n2, int n, int x, int a) {
  if (x == a) return true;
  int i = min(x, a), j = max(x, a) - 1;  // step ratios i .. j still to be multiplied in
  DD R = {1.0, 0.0};                     // -> w(max) / w(min)
  while (i <= j) {
    const int t = R.hi <= 1.0 ? i++ : j--;
    R = dd_mul_d(R, (double)(n1 - t) * (double)(n - t));            // both products are exact
    R = dd_div_d(R, (double)(t + 1) * (double)(n2 - n + t + 1));
  }
  if (x > a) return R.hi < kGamma || (R.hi == kGamma && R.lo <= 0.0);   // w(x)/w(a) = R <= gamma
  const DD G = dd_mul_d(R, kGamma);                                      // w(x)/w(a) = 1/R <= gamma
  return G.hi > 1.0 || (G.hi == 1.0 && G.lo >= 0.0);                     //   <=>  gamma * R >= 1
}
// w(x) <= w(a) (1 + 1e-14) for fp64 weights that agree to kAmbig
__device__ __forceinline__ bool hg_tie(int n1, int n2, int n, int x, int a) {
  // Exact ties that need no arithmetic: the observed table itself, and its mirror image when
  // a margin pair is symmetric (n1 == n2: w(x) = w(n - x); n == N - n: w(x) = w(n1 - x)).
  // These are nearly all the comparisons that get here; sending them through the
  // double-double product (|x - a| steps, run by the whole wavefront for one lane) made a
  // wavefront of balanced genes -- which list-slot order puts side by side -- the long pole.
  if (x == a || (n1 == n2 && x == n - a) || (2 * n == n1 + n2 && x == n1 - a)) return true;
  return hg_leq_dd(n1, n2, n, x, a);
}

// ---- spec S3, N <= 170: SciPy's own double ------------------------------------------------
// Up to 170 valid isolates scipy.stats.fisher_exact (1.15.3; the arithmetic behind scoary/methods.py:854)
// evaluates the hypergeometric pmf from a table of factorials and the tails by a term recurrence
// (Boost.Math: hypergeometric_pdf_factorial_imp / hypergeometric_cdf_imp), all in plain fp64 -- few enough
// operations to be restated one by one, so for such tables the p-value written to the CSV is the reference's
// to the last bit (checked against SciPy itself: every table up to N = 18 and 30 000 random ones up to 170,
// tests/golden/fisher_grid.npz and the exampledata goldens; N = 100 is the reference's own example).  Above
// 170 Boost switches to a prime factorisation with pow() -- not restated; there the walk above is the p,
// within 1e-12 (measured 3e-15) of SciPy's.  The rejection region always comes from the walk (exact rule).
constexpr int kSciPySmallN = 170;
__constant__ double kFactorial[kSciPySmallN + 1] = {
    1.0, 1.0, 2.0, 6.0, 24.0, 120.0, 720.0, 5040.0, 40320.0, 362880.0, 3628800.0, 39916800.0, 479001600.0,
    6227020800.0, 87178291200.0, 1307674368000.0, 20922789888000.0, 355687428096000.0, 6402373705728000.0,
    1.21645100408832e+17, 2.43290200817664e+18, 5.109094217170944e+19, 1.1240007277776077e+21, 2.585201673888498e+22,
    6.204484017332394e+23, 1.5511210043330986e+25, 4.0329146112660565e+26, 1.0888869450418352e+28, 3.0488834461171387e+29,
    8.841761993739702e+30, 2.6525285981219107e+32, 8.222838654177922e+33, 2.631308369336935e+35, 8.683317618811886e+36,
    2.9523279903960416e+38, 1.0333147966386145e+40, 3.7199332678990125e+41, 1.3763753091226346e+43, 5.230226174666011e+44,
    2.0397882081197444e+46, 8.159152832478977e+47, 3.345252661316381e+49, 1.40500611775288e+51, 6.041526306337383e+52,
    2.658271574788449e+54, 1.1962222086548019e+56, 5.502622159812089e+57, 2.5862324151116818e+59, 1.2413915592536073e+61,
    6.082818640342675e+62, 3.0414093201713376e+64, 1.5511187532873822e+66, 8.065817517094388e+67, 4.2748832840600255e+69,
    2.308436973392414e+71, 1.2696403353658276e+73, 7.109985878048635e+74, 4.0526919504877214e+76, 2.3505613312828785e+78,
    1.3868311854568984e+80, 8.32098711274139e+81, 5.075802138772248e+83, 3.146997326038794e+85, 1.98260831540444e+87,
    1.2688693218588417e+89, 8.247650592082472e+90, 5.443449390774431e+92, 3.647111091818868e+94, 2.4800355424368305e+96,
    1.711224524281413e+98, 1.1978571669969892e+100, 8.504785885678623e+101, 6.1234458376886085e+103, 4.4701154615126844e+105,
    3.307885441519386e+107, 2.48091408113954e+109, 1.8854947016660504e+111, 1.4518309202828587e+113, 1.1324281178206297e+115,
    8.946182130782976e+116, 7.156945704626381e+118, 5.797126020747368e+120, 4.753643337012842e+122, 3.945523969720659e+124,
    3.314240134565353e+126, 2.81710411438055e+128, 2.4227095383672734e+130, 2.107757298379528e+132, 1.8548264225739844e+134,
    1.650795516090846e+136, 1.4857159644817615e+138, 1.352001527678403e+140, 1.2438414054641308e+142, 1.1567725070816416e+144,
    1.087366156656743e+146, 1.032997848823906e+148, 9.916779348709496e+149, 9.619275968248212e+151, 9.426890448883248e+153,
    9.332621544394415e+155, 9.332621544394415e+157, 9.42594775983836e+159, 9.614466715035127e+161, 9.90290071648618e+163,
    1.0299016745145628e+166, 1.081396758240291e+168, 1.1462805637347084e+170, 1.226520203196138e+172, 1.324641819451829e+174,
    1.4438595832024937e+176, 1.588245541522743e+178, 1.7629525510902446e+180, 1.974506857221074e+182, 2.2311927486598138e+184,
    2.5435597334721877e+186, 2.925093693493016e+188, 3.393108684451898e+190, 3.969937160808721e+192, 4.684525849754291e+194,
    5.574585761207606e+196, 6.689502913449127e+198, 8.094298525273444e+200, 9.875044200833601e+202, 1.214630436702533e+205,
    1.506141741511141e+207, 1.882677176888926e+209, 2.372173242880047e+211, 3.0126600184576594e+213, 3.856204823625804e+215,
    4.974504222477287e+217, 6.466855489220474e+219, 8.47158069087882e+221, 1.1182486511960043e+224, 1.4872707060906857e+226,
    1.9929427461615188e+228, 2.6904727073180504e+230, 3.659042881952549e+232, 5.012888748274992e+234, 6.917786472619489e+236,
    9.615723196941089e+238, 1.3462012475717526e+241, 1.898143759076171e+243, 2.695364137888163e+245, 3.854370717180073e+247,
    5.5502938327393044e+249, 8.047926057471992e+251, 1.1749972043909107e+254, 1.727245890454639e+256, 2.5563239178728654e+258,
    3.80892263763057e+260, 5.713383956445855e+262, 8.62720977423324e+264, 1.3113358856834524e+267, 2.0063439050956823e+269,
    3.0897696138473508e+271, 4.789142901463394e+273, 7.471062926282894e+275, 1.1729568794264145e+278, 1.853271869493735e+280,
    2.9467022724950384e+282, 4.7147236359920616e+284, 7.590705053947219e+286, 1.2296942187394494e+289,
    2.0044015765453026e+291, 3.287218585534296e+293, 5.423910666131589e+295, 9.003691705778438e+297, 1.503616514864999e+300,
    2.5260757449731984e+302, 4.269068009004705e+304, 7.257415615307999e+306};
// Boost's pmf of x successes among n draws from N items of which r are successes: the quotient of
// factorials multiplied up and divided down so that the running value stays near 1
__device__ __noinline__ double bm_pdf(int x, int r, int n, int N) {
  double v = kFactorial[n];
  int i = 0, j = 0;
  while (i < 3 || j < 5) {
    while (j < 5 && (v >= 1.0 || i >= 3)) {
      v /= kFactorial[j == 0 ? N : j == 1 ? x : j == 2 ? n - x : j == 3 ? r - x : N - n - r + x];
      ++j;
    }
    while (i < 3 && (v <= 1.0 || j >= 5)) {
      v *= kFactorial[i == 0 ? r : i == 1 ? N - n : N - r];
      ++i;
    }
  }
  return v;
}
// Boost's lower tail P(X <= x) (upper == false) or upper tail P(X > x) (true): terms added from x
// towards the nearer end of the support until they no longer count, the far side as 1 - sum
__device__ __noinline__ double bm_tail(int x, int r, int n, int N, bool upper) {
  constexpr double kEps = 2.220446049250313e-16;
  const double mode = floor((double)(r + 1) * (double)(n + 1) / (double)(N + 2));
  double sum = 0.0;
  bool invert = upper;
  if ((double)x < mode) {
    sum = bm_pdf(x, r, n, N);
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
      sum = bm_pdf(x, r, n, N);
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
// scipy.stats.hypergeom's pmf / cdf / sf around them (support checks, clip to [0, 1]) and fisher_exact's
// two-sided rule (scipy/stats/_stats_py.py): the tail on the observed side plus the tail beyond the point
// a binary search over the pmf finds on the other side
struct SciPyHypergeom {
  int M, good, draws, lo, hi;
  __device__ SciPyHypergeom(int M_, int good_, int draws_)
      : M(M_), good(good_), draws(draws_), lo(max(draws_ - (M_ - good_), 0)), hi(min(good_, draws_)) {}
  __device__ static double clip(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
  __device__ double pmf(int k) const { return (k < lo || k > hi) ? 0.0 : clip(bm_pdf(k, good, draws, M)); }
  __device__ double cdf(int k) const { return k < lo ? 0.0 : (k >= hi ? 1.0 : clip(bm_tail(k, good, draws, M, false))); }
  __device__ double sf(int k) const { return k < lo ? 1.0 : (k >= hi ? 0.0 : clip(bm_tail(k, good, draws, M, true))); }
};
__device__ __noinline__ double scipy_small_p(int a, int b, int c, int d) {
  const int n1 = a + b, n2 = c + d, n = a + c;
  const SciPyHypergeom h(n1 + n2, n1, n);
  const int mode = (int)(((double)(n + 1) * (double)(n1 + 1)) / (double)(n1 + n2 + 2));
  const double pexact = h.pmf(a), pmode = h.pmf(mode);
  if (fabs(pexact - pmode) / fmax(pexact, pmode) <= 1e-14) return 1.0;
  const double target = pexact * kGamma;
  // _binary_search(f, target, lo, hi) on f = -pmf (a < mode: the descending side) or pmf (ascending side)
  const bool below = a < mode;
  if (below ? h.pmf(n) > target : h.pmf(0) > target) return below ? h.cdf(a) : h.sf(a - 1);
  const double sign = below ? -1.0 : 1.0, want = sign * target;
  int lo = below ? mode : 0, hi = below ? n : mode, guess;
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
  const double p = below ? h.cdf(a) + h.sf(guess) : h.sf(a - 1) + h.cdf(guess);
  return p < 1.0 ? p : 1.0;
}

// A lane PAIR per table: the even lane sums the weights from the mode upwards, the odd lane
// those below the mode.  A downward walk is an upward walk in the mirrored coordinates
// x' = n - x with the two margins swapped -- the oracle's downward step
// w(x-1) = w(x) * (x * (n2-n+x)) / ((n1-x+1) * (n-x+1)) and the upward step at n - x on (n2, n1, n) multiply
// and divide by the same two exact integer products, so the weights are bit-identical --
// which lets every lane run the same loop whatever its
// direction (no divergence between the lanes of a pair or between tables on either side of
// their mode) and halves the serial length of the kernel.
//
// SLOTS = false: tables in the caller's order, pair id = table.  SLOTS = true (the list-driven
// path, scoary_fisher_lists): grid.y = trait, pair id = list slot k, the table is that of gene
// order[k].  Slots are sorted by list length, i.e. by the gene's minority count, so the lanes
// of a wavefront walk supports of similar length (the trip count of a wavefront is that of
// its longest lane), and the rejection region can be written a second time in the form the
// list kernel wants -- (lo, hi1) of the LIST count u, in slot order -- which saves the
// separate conversion launch (k_lists_crit):
//   ones-list : u in [base, base + span);  zeros-list: u in [npos - base - span + 1, npos - base + 1)
template <bool SLOTS>
__global__ __launch_bounds__(64) void k_fisher(const int4* __restrict__ tables, int64_t M,
                                               double* __restrict__ p_out,
                                               double* __restrict__ or_out,
                                               uint2* __restrict__ crit,
                                               const int32_t* __restrict__ order,
                                               const uint8_t* __restrict__ flipped,
                                               uint2* __restrict__ lcrit) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t pid = tid >> 1;
  const bool down = tid & 1;
  if (pid >= M) return;            // M = tables (SLOTS = false) or genes per trait (true)
  int64_t idx = pid, lidx = 0;
  bool zeros_list = false;
  if constexpr (SLOTS) {
    const int g = order[pid];
    idx = (int64_t)blockIdx.y * M + g;
    lidx = (int64_t)blockIdx.y * M + pid;
    zeros_list = flipped[g] != 0;
  }
  const int4 c = tables[idx];
  const int b = c.y, cc = c.z, d = c.w;
  const int n0 = c.x + cc;
  if (c.x + b == 0 || cc + d == 0 || n0 == 0 || b + d == 0) {
    if (down) {
      or_out[idx] = __longlong_as_double(0x7ff8000000000000LL);
    } else {
      p_out[idx] = 1.0;
      if (crit) crit[idx] = make_uint2(0u, 0u);
      if constexpr (SLOTS) lcrit[lidx] = make_uint2(0u, 0u);
    }
    return;
  }
  // Canonical orientation (round 6): a gene and the gene with the complementary pattern -- the tables
  // [[a, b], [c, d]] and [[b, a], [d, c]] -- have the same p, and SciPy returns the same double for both
  // whenever the table is larger than its factorial table (N > 170: 400 of 400 random pairs at N = 200,
  // 500, 2000; about half below), so the reference's stable sort leaves such rows in file order.  Walking
  // each table from its own mode gave the pair p-values one ulp apart (the mode's term joins the other
  // lane's sum; with two equal modes the weights are normalised at the other one) and the rows changed
  // places in the CSV (tests/test_gpu_fuzz.py).  So a gene carried by more than half of the valid isolates
  // (at exactly half: the larger a) is evaluated as its complement, in the coordinate x' = n1 - x, and only
  // the region bounds are mapped back.
  const int n1 = c.x + b, n2 = cc + d;
  const bool mirrored = n0 > b + d || (n0 == b + d && c.x > b);
  const int a = mirrored ? b : c.x, n = mirrored ? b + d : n0;
  const int lo = max(0, n - n2), hi = min(n, n1);
  int mode = (int)(((double)(n + 1) * (double)(n1 + 1)) / (double)(n1 + n2 + 2));
  mode = min(max(mode, lo), hi);

  // weight of the observed table, walking from the mode towards a
  double w = 1.0;
  {
    const bool below = a < mode;
    const int xs = below ? n - mode : mode, xe = below ? n - a : a;
    HgWalk wk(xs, below ? n2 : n1, below ? n1 : n2, n);
    for (int x = xs; x < xe; ++x) w = wk.step(w);
  }
  const double wobs = w;

  // The walk stops once a term is inside the rejection region AND below
  // 2^-64 of the observed table's weight: what is left of the (super-geometrically
  // decaying) tail is < 1e-18 of the included sum -- six orders inside the 1e-12 the
  // path promises, also RELATIVE to a p of 1e-200 -- and the region boundary has already
  // been passed, so (L, H) are exact.  (2^-90 until round 6: 14 % more steps per table for
  // digits nothing reads; k_fisher 0.126 -> 0.11 ms at cfg3, and it is on the critical path of
  // a gene-sharded rank.)
  // The body is predicated rather than branched (a lone wavefront issues an instruction
  // every ~6 cycles, so the instruction count is the latency of a small problem): only the
  // comparisons inside the 1e-9 band around a tie -- rare -- leave the straight line.
  const double tiny = 5.421010862427522e-20 * (wobs < 1.0 ? wobs : 1.0);   // 2^-64 * w_obs
  const double sure_in = wobs * (1.0 - kAmbig), sure_out = wobs * (1.0 + kAmbig);
  const int xend = down ? n - lo : hi;
  int x = down ? n - mode : mode;          // own (mirrored, for the odd lane) coordinate
  int first = -1;                          // first term inside the region, own coordinate
  bool take = !down;                       // the mode's term belongs to the upward lane
  double tot = 0.0, inc = 0.0;
  HgWalk wk(x, down ? n2 : n1, down ? n1 : n2, n);
  w = 1.0;
  for (;;) {
    bool in = w <= sure_in;
    if (!in && !(w > sure_out)) in = hg_tie(n1, n2, n, down ? n - x : x, a);
    in = in && take;
    tot += take ? w : 0.0;
    inc += in ? w : 0.0;
    first = (in && first < 0) ? x : first;
    if ((in && w < tiny) || x >= xend) break;
    take = true;
    w = wk.step(w);
    ++x;
  }
  tot += __shfl_xor(tot, 1);
  inc += __shfl_xor(inc, 1);
  const int other = __shfl_xor(first, 1);
  if (down) {
    or_out[idx] = (cc > 0 && b > 0) ? ((double)c.x * (double)d) / ((double)cc * (double)b)
                                    : __longlong_as_double(0x7ff0000000000000LL);
    return;
  }
  const int Hc = first >= 0 ? first : hi + 1;          // region bounds in the canonical coordinate
  const int Lc = other >= 0 ? n - other : lo - 1;
  const bool all = (Hc == mode);
  double p = all ? 1.0 : inc / tot;
  if (n1 + n2 <= kSciPySmallN) p = scipy_small_p(c.x, b, cc, d);   // the reference's own double (above)
  p_out[idx] = p < 1.0 ? p : 1.0;
  const int L = mirrored ? n1 - Hc : Lc, H = mirrored ? n1 - Lc : Hc;   // ... and in the table's own a
  const uint32_t base = (uint32_t)(L + 1), span = all ? 0u : (uint32_t)(H - L - 1);
  if (crit) crit[idx] = all ? make_uint2(0u, 0u) : make_uint2(base, span);
  if constexpr (SLOTS) {
    const uint32_t npos = (uint32_t)n1;                // the trait's positives among the valid isolates
    uint2 o = make_uint2(0u, 0u);                      // span 0: every permutation is in the region
    if (span != 0u)
      o = zeros_list ? make_uint2(npos - base - span + 1u, npos - base + 1u)
                     : make_uint2(base, base + span);
    lcrit[lidx] = o;
  }
}

// ----------------------------------------------------------------------------
// a7: permutation exceedance counts
// ----------------------------------------------------------------------------
// Register-resident variant: a lane keeps GL whole gene rows (RQ quads each)
// in VGPRs and walks a chunk of permuted label rows, which arrive as scalar
// loads.  Work per (gene, permutation): 8*RQ VALU ops + 3 for the region test.
// grid = (Gp / (64*GL), perm chunks, T), block = one wavefront.
template <int RQ, int GL>
__global__ __launch_bounds__(64) void k_permute_reg(const uint4* __restrict__ tiled,
                                                    const uint32_t* __restrict__ perms,
                                                    const uint2* __restrict__ crit, int G, int Gp,
                                                    int64_t P, int pchunk,
                                                    uint32_t* __restrict__ r) {
  const int t = blockIdx.z;
  const int lane = threadIdx.x;
  const int g0 = blockIdx.x * (kWave * GL) + lane;
  const int64_t p0 = (int64_t)blockIdx.y * pchunk;
  const int np = (int)min((int64_t)pchunk, P - p0);

  uint4 gw[GL][RQ];
  uint32_t base[GL], span[GL], cnt[GL];
#pragma unroll
  for (int gl = 0; gl < GL; ++gl) {
    const int g = g0 + gl * kWave;
#pragma unroll
    for (int q = 0; q < RQ; ++q) gw[gl][q] = tiled[(int64_t)q * Gp + g];
    const uint2 cr = (g < G) ? crit[(int64_t)t * G + g] : make_uint2(0u, 0u);
    base[gl] = cr.x;
    span[gl] = cr.y;
    cnt[gl] = 0;
  }

  const uint4* prow = reinterpret_cast<const uint4*>(perms + ((int64_t)t * P + p0) * (RQ * 4));
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
  for (int i = 0; i < np; ++i, prow += RQ) {
    uint32_t acc[GL][4];
#pragma unroll
    for (int gl = 0; gl < GL; ++gl) acc[gl][0] = acc[gl][1] = acc[gl][2] = acc[gl][3] = 0;
#pragma unroll
    for (int q = 0; q < RQ; ++q) {
      const uint4 s = prow[q];  // wave-uniform address -> s_load
#pragma unroll
      for (int gl = 0; gl < GL; ++gl) {
        bcnt_acc(acc[gl][0], gw[gl][q].x & s.x);
        bcnt_acc(acc[gl][1], gw[gl][q].y & s.y);
        bcnt_acc(acc[gl][2], gw[gl][q].z & s.z);
        bcnt_acc(acc[gl][3], gw[gl][q].w & s.w);
      }
    }
#pragma unroll
    for (int gl = 0; gl < GL; ++gl) {
      const uint32_t a = (acc[gl][0] + acc[gl][1]) + (acc[gl][2] + acc[gl][3]);
      cnt[gl] += ((a - base[gl]) >= span[gl]) ? 1u : 0u;
    }
  }
#pragma unroll
  for (int gl = 0; gl < GL; ++gl) {
    const int g = g0 + gl * kWave;
    if (g < G && cnt[gl]) atomicAdd(&r[(int64_t)t * G + g], cnt[gl]);
  }
}

// General variant for rows too long to keep in registers: CQ-quad register
// chunks of the gene row, PB permutations accumulated per pass.
template <int CQ, int PB>
__global__ __launch_bounds__(64) void k_permute_chunked(const uint4* __restrict__ tiled,
                                                        const uint32_t* __restrict__ perms,
                                                        const uint2* __restrict__ crit, int G,
                                                        int Gp, int Qp, int64_t P, int pchunk,
                                                        uint32_t* __restrict__ r) {
  const int t = blockIdx.z;
  const int g = blockIdx.x * kWave + threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.y * pchunk;
  const int np = (int)min((int64_t)pchunk, P - p0);
  const uint2 cr = (g < G) ? crit[(int64_t)t * G + g] : make_uint2(0u, 0u);
  const int nchunks = Qp / CQ;
  uint32_t cnt = 0;
  const uint4* pbase = reinterpret_cast<const uint4*>(perms + ((int64_t)t * P + p0) * ((int64_t)Qp * 4));
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
  for (int i0 = 0; i0 < np; i0 += PB) {
    uint32_t acc[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[j] = 0;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int c = 0; c < nchunks; ++c) {
      uint4 gw[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) gw[q] = tiled[(int64_t)(c * CQ + q) * Gp + g];
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const int i = min(i0 + j, np - 1);
        const uint4* pr = pbase + (int64_t)i * Qp + c * CQ;
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
          const uint4 s = pr[q];  // wave-uniform -> s_load
          bcnt_acc(acc[j], gw[q].x & s.x);
          bcnt_acc(acc[j], gw[q].y & s.y);
          bcnt_acc(acc[j], gw[q].z & s.z);
          bcnt_acc(acc[j], gw[q].w & s.w);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j)
      if (i0 + j < np) cnt += ((acc[j] - cr.x) >= cr.y) ? 1u : 0u;
  }
  if (g < G && cnt) atomicAdd(&r[(int64_t)t * G + g], cnt);
}

// The reference's SEQUENTIAL estimator with early abort (scoary/methods.py:1348-1365), for the
// Fisher statistic (opt-in, --permute-early-abort): a lane = one (gene, trait) walks ALL the
// permutations in index order -- no permutation chunks across blocks -- with the running
// count r; from permutation index i >= 30 on, the first i with r >= thr[i] (thr[i] = smallest
// r with 1 - binom.cdf(r, i, 0.1) < 0.05, built on the host exactly as the reference
// evaluates it) freezes the lane: nstop = i + 1, Empirical_p = (r + 1) / (i + 2).  Lanes that
// never stop end with nstop = 0 and (r + 1) / (P + 1).  The state (r, nstop) lives in HBM
// between batches of permutations; a wavefront leaves a batch as soon as all its lanes have
// stopped.  Same CQ-quad register chunks x PB permutations per pass as k_permute_chunked.
template <int CQ, int PB>
__global__ __launch_bounds__(64) void k_permute_seq(const uint4* __restrict__ tiled,
                                                    const uint32_t* __restrict__ perms,
                                                    const uint2* __restrict__ crit,
                                                    const uint32_t* __restrict__ thr, int G, int Gp,
                                                    int Qp, int64_t P, int64_t perm_base,
                                                    uint32_t* __restrict__ r,
                                                    uint32_t* __restrict__ nstop) {
  const int t = blockIdx.y;
  const int g = blockIdx.x * kWave + threadIdx.x;
  const bool have = g < G;
  const uint2 cr = have ? crit[(int64_t)t * G + g] : make_uint2(0u, 0u);
  uint32_t run = have ? r[(int64_t)t * G + g] : 0u;
  uint32_t stop = have ? nstop[(int64_t)t * G + g] : 1u;       // padding lanes: stopped
  const int nchunks = Qp / CQ;
  const int np = (int)P;
  const uint4* pbase = reinterpret_cast<const uint4*>(perms + (int64_t)t * P * ((int64_t)Qp * 4));
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
  for (int i0 = 0; i0 < np; i0 += PB) {
    if (__ballot(stop == 0u) == 0) break;                      // every lane of the wavefront is done
    uint32_t acc[PB];
#pragma unroll
    for (int j = 0; j < PB; ++j) acc[j] = 0;
#pragma clang loop unroll(disable) vectorize(disable) interleave(disable)
    for (int c = 0; c < nchunks; ++c) {
      uint4 gw[CQ];
#pragma unroll
      for (int q = 0; q < CQ; ++q) gw[q] = tiled[(int64_t)(c * CQ + q) * Gp + min(g, Gp - 1)];
#pragma unroll
      for (int j = 0; j < PB; ++j) {
        const int i = min(i0 + j, np - 1);
        const uint4* pr = pbase + (int64_t)i * Qp + c * CQ;
#pragma unroll
        for (int q = 0; q < CQ; ++q) {
          const uint4 s = pr[q];  // wave-uniform -> s_load
          bcnt_acc(acc[j], gw[q].x & s.x);
          bcnt_acc(acc[j], gw[q].y & s.y);
          bcnt_acc(acc[j], gw[q].z & s.z);
          bcnt_acc(acc[j], gw[q].w & s.w);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      if (i0 + j < np && stop == 0u) {
        const int64_t i = perm_base + i0 + j;                  // global permutation index
        run += ((acc[j] - cr.x) >= cr.y) ? 1u : 0u;
        if (i >= 30 && run >= thr[i]) stop = (uint32_t)(i + 1);
      }
    }
  }
  if (have) {
    r[(int64_t)t * G + g] = run;
    nstop[(int64_t)t * G + g] = stop;
  }
}

template <int RQ, int GL>
void launch_permute_reg(dim3 grid, hipStream_t s, const uint32_t* tiled, const uint32_t* perms,
                        const uint32_t* crit, int G, int Gp, int64_t P, int pchunk, uint32_t* r) {
  hipLaunchKernelGGL((k_permute_reg<RQ, GL>), grid, dim3(kWave), 0, s,
                     reinterpret_cast<const uint4*>(tiled), perms,
                     reinterpret_cast<const uint2*>(crit), G, Gp, P, pchunk, r);
}

}  // namespace

extern "C" {

int scoary_pack_dense(scoary_handle h, const uint8_t* d_dense, int64_t G, int64_t N,
                      uint32_t* d_tiled, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_dense || !d_tiled || G < 1 || N < 1) return fail(h, SCOARY_ERR_ARG, "scoary_pack_dense: bad argument");
  DeviceGuard guard(h->device);
  const int64_t Gp = scoary_tiled_genes(G), Qp = scoary_tiled_quads(N);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_pack_dense");
  hipLaunchKernelGGL(k_pack_dense, dim3((unsigned)(Gp / 256), (unsigned)(Qp * 4)), dim3(256), 0, s,
                     d_dense, G, N, Gp, Qp, d_tiled);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int scoary_tile_rows(scoary_handle h, const uint64_t* d_rows64, int64_t G, int64_t N,
                     uint32_t* d_tiled, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_rows64 || !d_tiled || G < 1 || N < 1) return fail(h, SCOARY_ERR_ARG, "scoary_tile_rows: bad argument");
  DeviceGuard guard(h->device);
  const int64_t Gp = scoary_tiled_genes(G), Qp = scoary_tiled_quads(N);
  const int64_t W32 = 2 * ((N + 63) / 64);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_tile_rows");
  hipLaunchKernelGGL(k_tile_rows, dim3((unsigned)(Gp / 256), (unsigned)Qp), dim3(256), 0, s,
                     reinterpret_cast<const uint32_t*>(d_rows64), G, W32, Gp,
                     reinterpret_cast<uint4*>(d_tiled));
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

extern "C++" {
static int launch_trait_plan(scoary_handle h, hipStream_t s, const uint32_t* d_traits, const uint32_t* d_masks,
                             int64_t T, int64_t N, int32_t* d_margins, int32_t* d_mask_class, int32_t* d_plan) {
  const int64_t Qp = scoary_tiled_quads(N);
  const PlanLayout L = plan_layout(T, Qp);
  KernelTimer kt(h, s, "k_trait_plan");
  hipLaunchKernelGGL(k_trait_plan, dim3((unsigned)T), dim3(64), 0, s, d_traits, d_masks,
                     (int)scoary_row_words(N), (int)T, (int)L.tb, d_margins, d_plan + L.cls, d_mask_class);
  hipLaunchKernelGGL(k_trait_slots, dim3((unsigned)L.passes), dim3(64), 0, s, d_plan, (int)T, (int)Qp);
  hipLaunchKernelGGL(k_trait_vecq, dim3((unsigned)Qp, (unsigned)L.passes), dim3(64), 0, s,
                     reinterpret_cast<const uint4*>(d_traits), reinterpret_cast<const uint4*>(d_masks),
                     d_plan, (int)T, (int)Qp);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}
static int launch_counts(scoary_handle h, hipStream_t s, const uint32_t* d_tiled, const int32_t* d_plan,
                         const int32_t* d_margins, int64_t G, int64_t T, int64_t N, int32_t* d_counts) {
  const int64_t Gp = scoary_tiled_genes(G), Qp = scoary_tiled_quads(N);
  const PlanLayout L = plan_layout(T, Qp);
  // wavefronts per 64 genes: enough of them for ~8 per SIMD over the chip, at least two quads each
  const int64_t base = Gp / kWave * L.passes, want = (int64_t)h->num_cu * 4 * 8;
  int qs = 1;
  while (qs < 16 && base * qs < want && 4 * qs <= Qp) qs *= 2;
  // Round 6: on the cold stream (125 000 x 10 000, 1954 blocks: 4 wavefronts each are 7816 and just miss the target)
  // blocks of 4 wavefronts beat blocks of 8 at every T but 16 -- T = 4: 34.2 against 38.1 us = 0.60 against 0.54 of
  // 8 TB/s, T = 32: 107 against 117 us -- and blocks of 2 or 16 lose (profiles/r06_k1_launch_geometry_ab.txt): where
  // four already give 6 wavefronts per SIMD, four it is (the 12- / 16-trait instances keep the larger blocks).
  if (qs > 4 && L.tb != 12 && L.tb != 16 && base * 4 >= (int64_t)h->num_cu * 4 * 6) qs = 4;
  const dim3 grid((unsigned)(Gp / kWave), (unsigned)L.passes), block((unsigned)(kWave * qs));
  KernelTimer kt(h, s, "k_counts");
#define COUNTS(TBV)                                                                                   \
  hipLaunchKernelGGL((k_counts<TBV>), grid, block, 0, s, reinterpret_cast<const uint4*>(d_tiled),     \
                     d_plan, d_margins, (int)G, (int)Gp, (int)Qp, (int)T, reinterpret_cast<int4*>(d_counts))
  switch (L.tb) {
    case 1: COUNTS(1); break;
    case 2: COUNTS(2); break;
    case 4: COUNTS(4); break;
    case 8: COUNTS(8); break;
    case 12: COUNTS(12); break;
    case 16: COUNTS(16); break;
    case 20: COUNTS(20); break;
    case 24: COUNTS(24); break;
    case 28: COUNTS(28); break;
    default: COUNTS(32); break;
  }
#undef COUNTS
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}
}  // extern "C++"

int64_t scoary_counts_traits_per_pass(int64_t T) { return T < 1 ? 0 : plan_traits_per_pass(T); }
int64_t scoary_trait_plan_bytes(int64_t T, int64_t N) {
  return (T < 1 || N < 1) ? 0 : plan_layout(T, scoary_tiled_quads(N)).words * (int64_t)sizeof(int32_t);
}

int scoary_trait_plan(scoary_handle h, const uint32_t* d_traits, const uint32_t* d_masks, int64_t T,
                      int64_t N, int32_t* d_margins, int32_t* d_mask_class, void* d_plan,
                      scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_traits || !d_masks || !d_margins || !d_plan || T < 1 || N < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_trait_plan: bad argument");
  if (T > 65535 * 8) return fail(h, SCOARY_ERR_SIZE, "scoary_trait_plan: T too large");
  DeviceGuard guard(h->device);
  return launch_trait_plan(h, static_cast<hipStream_t>(stream), d_traits, d_masks, T, N, d_margins,
                           d_mask_class, static_cast<int32_t*>(d_plan));
}

int scoary_counts_planned(scoary_handle h, const uint32_t* d_tiled, const void* d_plan,
                          const int32_t* d_margins, int64_t G, int64_t T, int64_t N,
                          int32_t* d_counts, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_plan || !d_margins || !d_counts || G < 1 || T < 1 || N < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_counts_planned: bad argument");
  if (G > (int64_t)1 << 30 || T > 65535 * 8)
    return fail(h, SCOARY_ERR_SIZE, "scoary_counts_planned: G or T too large");
  DeviceGuard guard(h->device);
  return launch_counts(h, static_cast<hipStream_t>(stream), d_tiled, static_cast<const int32_t*>(d_plan),
                       d_margins, G, T, N, d_counts);
}

int scoary_counts(scoary_handle h, const uint32_t* d_tiled, const uint32_t* d_traits,
                  const uint32_t* d_masks, int64_t G, int64_t T, int64_t N, int32_t* d_counts,
                  int32_t* d_margins, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_traits || !d_masks || !d_counts || !d_margins || G < 1 || T < 1 || N < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_counts: bad argument");
  if (G > (int64_t)1 << 30 || T > 65535 * 8) return fail(h, SCOARY_ERR_SIZE, "scoary_counts: G or T too large");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the one-call form: a plan in stream-ordered temporary memory, then the tables.  Callers with more
  // than one step per trait set build the plan once (scoary_trait_plan + scoary_counts_planned) --
  // and so must a caller that records a hipGraph: the temporary allocation does not belong in one.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone)
    return fail(h, SCOARY_ERR_ARG, "scoary_counts: the stream is capturing a graph; use scoary_trait_plan "
                                   "(outside the capture) + scoary_counts_planned");
  void* tmp = nullptr;
  HIP_TRY(h, hipMallocAsync(&tmp, (size_t)scoary_trait_plan_bytes(T, N), s));
  int rc = launch_trait_plan(h, s, d_traits, d_masks, T, N, d_margins, nullptr, static_cast<int32_t*>(tmp));
  if (rc == SCOARY_OK)
    rc = launch_counts(h, s, d_tiled, static_cast<const int32_t*>(tmp), d_margins, G, T, N, d_counts);
  (void)hipFreeAsync(tmp, s);
  return rc;
}

int scoary_fisher(scoary_handle h, const int32_t* d_tables, int64_t M, double* d_p, double* d_or,
                  uint32_t* d_crit, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tables || !d_p || !d_or || M < 1) return fail(h, SCOARY_ERR_ARG, "scoary_fisher: bad argument");
  if ((M + kWave - 1) / kWave > 0x7fffffffLL) return fail(h, SCOARY_ERR_SIZE, "scoary_fisher: M too large");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_fisher");
  hipLaunchKernelGGL(k_fisher<false>, dim3((unsigned)((2 * M + kWave - 1) / kWave)), dim3(kWave), 0, s,
                     reinterpret_cast<const int4*>(d_tables), M, d_p, d_or,
                     reinterpret_cast<uint2*>(d_crit), nullptr, nullptr, nullptr);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int scoary_fisher_lists(scoary_handle h, const int32_t* d_tables, int64_t T, int64_t G,
                        const int32_t* d_lorder, const uint8_t* d_lflipped, double* d_p,
                        double* d_or, uint32_t* d_crit, uint32_t* d_lcrit,
                        scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tables || !d_lorder || !d_lflipped || !d_p || !d_or || !d_lcrit || T < 1 || G < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_fisher_lists: bad argument");
  if (T > 65535 || (2 * G + kWave - 1) / kWave > 0x7fffffffLL)
    return fail(h, SCOARY_ERR_SIZE, "scoary_fisher_lists: T > 65535 or G too large");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_fisher");
  hipLaunchKernelGGL(k_fisher<true>, dim3((unsigned)((2 * G + kWave - 1) / kWave), (unsigned)T),
                     dim3(kWave), 0, s, reinterpret_cast<const int4*>(d_tables), G, d_p, d_or,
                     reinterpret_cast<uint2*>(d_crit), d_lorder, d_lflipped,
                     reinterpret_cast<uint2*>(d_lcrit));
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int scoary_permute(scoary_handle h, const uint32_t* d_tiled, const uint32_t* d_perms,
                   const uint32_t* d_crit, int64_t G, int64_t T, int64_t N, int64_t P,
                   uint32_t* d_r, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_perms || !d_crit || !d_r || G < 1 || T < 1 || N < 1 || P < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_permute: bad argument");
  if (T > 65535 || G > (int64_t)1 << 30) return fail(h, SCOARY_ERR_SIZE, "scoary_permute: T > 65535 or G > 2^30");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t Gp = scoary_tiled_genes(G), Qp = scoary_tiled_quads(N);

  // Enough independent wave-tasks to fill 256 CUs x 4 SIMDs several times
  // over: split the permutation range when there are few gene-waves.
  const int GL = 1;
  const int64_t gene_waves = Gp / (kWave * GL);
  const int64_t want_tasks = (int64_t)h->num_cu * 4 * 32;
  int64_t nch = (want_tasks + gene_waves * T - 1) / (gene_waves * T);
  const int64_t max_ch = (P + 63) / 64;  // keep >= 64 permutations per task
  if (nch > max_ch) nch = max_ch;
  if (nch < 1) nch = 1;
  if (nch > 65535) nch = 65535;
  const int pchunk = (int)((P + nch - 1) / nch);
  nch = (P + pchunk - 1) / pchunk;
  dim3 grid((unsigned)gene_waves, (unsigned)nch, (unsigned)T);

  // Tuning knob (experiments only): SCOARY_PERMUTE_VARIANT=reg|c8x8|c8x16|c4x16|c8x4
  const char* variant = std::getenv("SCOARY_PERMUTE_VARIANT");
  const bool force_chunk = variant && variant[0] == 'c';
  const bool use_reg = Qp <= kMaxRegQuads && !force_chunk && (Qp <= kAutoRegQuads || (variant && variant[0] == 'r'));
  KernelTimer kt(h, s, "k_permute");
  if (use_reg) {
    switch (Qp) {
#define CASE_RQ(RQ)                                                                          \
  case RQ:                                                                                   \
    launch_permute_reg<RQ, 1>(grid, s, d_tiled, d_perms, d_crit, (int)G, (int)Gp, P, pchunk, d_r); \
    break;
      CASE_RQ(1) CASE_RQ(2) CASE_RQ(4) CASE_RQ(6) CASE_RQ(8) CASE_RQ(12) CASE_RQ(16) CASE_RQ(20)
      CASE_RQ(24) CASE_RQ(32) CASE_RQ(40) CASE_RQ(48)
#undef CASE_RQ
      default:
        return fail(h, SCOARY_ERR_SIZE, "scoary_permute: unsupported tiled row size");
    }
  } else {
    const uint4* t4 = reinterpret_cast<const uint4*>(d_tiled);
    const uint2* c2 = reinterpret_cast<const uint2*>(d_crit);
    const std::string v = variant ? variant : "";
#define LAUNCH_CHUNK(CQ, PB)                                                                  \
  hipLaunchKernelGGL((k_permute_chunked<CQ, PB>), grid, dim3(kWave), 0, s, t4, d_perms, c2, (int)G, \
                     (int)Gp, (int)Qp, P, pchunk, d_r)
    if (v == "c8x16" && Qp % 8 == 0) LAUNCH_CHUNK(8, 16);
    else if (v == "c4x16" && Qp % 4 == 0) LAUNCH_CHUNK(4, 16);
    else if (v == "c8x4" && Qp % 8 == 0) LAUNCH_CHUNK(8, 4);
    else if (Qp % 8 == 0) LAUNCH_CHUNK(8, 8);
    else if (Qp % 4 == 0) LAUNCH_CHUNK(4, 16);
    else if (Qp % 2 == 0) LAUNCH_CHUNK(2, 16);
    else LAUNCH_CHUNK(1, 16);
#undef LAUNCH_CHUNK
  }
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int scoary_permute_seq(scoary_handle h, const uint32_t* d_tiled, const uint32_t* d_perms,
                       const uint32_t* d_crit, const uint32_t* d_thr, int64_t G, int64_t T,
                       int64_t N, int64_t P, int64_t perm_base, uint32_t* d_r, uint32_t* d_nstop,
                       scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_perms || !d_crit || !d_thr || !d_r || !d_nstop || G < 1 || T < 1 || N < 1 ||
      P < 1 || perm_base < 0)
    return fail(h, SCOARY_ERR_ARG, "scoary_permute_seq: bad argument");
  if (T > 65535 || P > 0x7fffffffLL || perm_base + P > 0xffffffffLL)
    return fail(h, SCOARY_ERR_SIZE, "scoary_permute_seq: T > 65535 or permutation index >= 2^32");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t Gp = scoary_tiled_genes(G), Qp = scoary_tiled_quads(N);
  KernelTimer kt(h, s, "k_permute_seq");
  const dim3 grid((unsigned)(Gp / kWave), (unsigned)T);
  const uint4* t4 = reinterpret_cast<const uint4*>(d_tiled);
  const uint2* c2 = reinterpret_cast<const uint2*>(d_crit);
  // tiled rows come in the register sizes of kRegQuads (any of 1, 2, 4, 6 ... quads) or in
  // multiples of kChunkQuads: chunk = the largest of 8 / 4 / 2 / 1 quads that divides Qp
  if (Qp % 8 == 0)
    hipLaunchKernelGGL((k_permute_seq<8, 8>), grid, dim3(kWave), 0, s, t4, d_perms, c2, d_thr, (int)G,
                       (int)Gp, (int)Qp, P, perm_base, d_r, d_nstop);
  else if (Qp % 4 == 0)
    hipLaunchKernelGGL((k_permute_seq<4, 8>), grid, dim3(kWave), 0, s, t4, d_perms, c2, d_thr, (int)G,
                       (int)Gp, (int)Qp, P, perm_base, d_r, d_nstop);
  else if (Qp % 2 == 0)
    hipLaunchKernelGGL((k_permute_seq<2, 8>), grid, dim3(kWave), 0, s, t4, d_perms, c2, d_thr, (int)G,
                       (int)Gp, (int)Qp, P, perm_base, d_r, d_nstop);
  else
    hipLaunchKernelGGL((k_permute_seq<1, 8>), grid, dim3(kWave), 0, s, t4, d_perms, c2, d_thr, (int)G,
                       (int)Gp, (int)Qp, P, perm_base, d_r, d_nstop);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

}  // extern "C"
