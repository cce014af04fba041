// scoary_hip.hip -- gfx950 (MI355X / CDNA4) kernels + C-ABI for Scoary's
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
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "scoary_hip.h"

namespace {

constexpr int kWave = 64;
constexpr int kGeneAlign = 256;
constexpr double kTie = 1e-10;             // spec S3: relative tie window
constexpr uint32_t kPermDomain = 0x53434F41u;  // "SCOA", spec S4

// Row sizes (in quads of four 32-bit words) for which a gene row is held
// entirely in VGPRs by k_permute_reg.
constexpr int kRegQuads[] = {1, 2, 4, 6, 8, 12, 16, 20, 24, 32, 40, 48};
constexpr int kMaxRegQuads = 48;
constexpr int kAutoRegQuads = 24;  // longer rows: the chunked kernel wins (measured 1.39x at N=5000)
constexpr int kChunkQuads = 8;  // k_permute_chunked: quads per register chunk

inline int64_t round_up(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

inline int64_t tiled_quads(int64_t N) {
  int64_t q = (((N + 31) / 32) + 3) / 4;
  if (q < 1) q = 1;
  for (int r : kRegQuads)
    if (r >= q) return r;
  return round_up(q, kChunkQuads);
}

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
__global__ __launch_bounds__(64) void k_margins(const uint32_t* __restrict__ traits,
                                                const uint32_t* __restrict__ masks, int Wp,
                                                int32_t* __restrict__ margins) {
  const int t = blockIdx.x;
  int npos = 0, nval = 0;
  for (int k = threadIdx.x; k < Wp; k += kWave) {
    npos += __popc(traits[(int64_t)t * Wp + k]);
    nval += __popc(masks[(int64_t)t * Wp + k]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    npos += __shfl_down(npos, off);
    nval += __shfl_down(nval, off);
  }
  if (threadIdx.x == 0) {
    margins[2 * t] = npos;
    margins[2 * t + 1] = nval;
  }
}

__device__ __forceinline__ int popc4(const uint4 a, const uint4 b) {
  return __popc(a.x & b.x) + __popc(a.y & b.y) + __popc(a.z & b.z) + __popc(a.w & b.w);
}

// acc += popcount(x) as ONE v_bcnt_u32_b32 (its second operand is the
// accumulator).  Opaque to the optimiser on purpose: left to itself LLVM
// reassociates the accumulate chain into short chains joined by v_add3_u32,
// ~15 % more VALU work in the permutation inner loop.
__device__ __forceinline__ void bcnt_acc(uint32_t& acc, uint32_t x) {
  asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
}

// Lane = gene; blockIdx.y = group of TB traits.  Gene quads stream once per
// trait group (coalesced 16 B / lane); trait and mask quads are wave-uniform
// (scalar loads).
template <int TB>
__global__ __launch_bounds__(256) void k_counts(const uint4* __restrict__ tiled,
                                                const uint32_t* __restrict__ traits,
                                                const uint32_t* __restrict__ masks,
                                                const int32_t* __restrict__ margins, int G,
                                                int Gp, int Qp, int T, int4* __restrict__ counts) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int t0 = blockIdx.y * TB;
  const int Wp = Qp * 4;
  int a[TB], m[TB];
  const uint4* trow[TB];
  const uint4* mrow[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j) {
    a[j] = 0;
    m[j] = 0;
    const int t = min(t0 + j, T - 1);
    trow[j] = reinterpret_cast<const uint4*>(traits + (int64_t)t * Wp);
    mrow[j] = reinterpret_cast<const uint4*>(masks + (int64_t)t * Wp);
  }
  for (int q = 0; q < Qp; ++q) {
    const uint4 gw = tiled[(int64_t)q * Gp + g];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      a[j] += popc4(gw, trow[j][q]);
      m[j] += popc4(gw, mrow[j][q]);
    }
  }
  if (g >= G) return;
#pragma unroll
  for (int j = 0; j < TB; ++j) {
    const int t = t0 + j;
    if (t < T) {
      const int npos = margins[2 * t], nval = margins[2 * t + 1];
      counts[(int64_t)t * G + g] =
          make_int4(a[j], npos - a[j], m[j] - a[j], nval - npos - m[j] + a[j]);
    }
  }
}

// ----------------------------------------------------------------------------
// a5: two-sided Fisher exact test (spec S3)
// ----------------------------------------------------------------------------
// Hypergeometric weights by the exact ratio recurrence, normalised at the
// mode; same operation order as the oracle's hg_weights so the weights (and
// hence the rejection regions) are bit-identical on both sides.
__device__ __forceinline__ double w_up(double w, int x, int n1, int n2, int n) {
  return w * ((double)(n1 - x) * (double)(n - x)) / ((double)(x + 1) * (double)(n2 - n + x + 1));
}
__device__ __forceinline__ double w_down(double w, int x, int n1, int n2, int n) {
  return w * ((double)x * (double)(n2 - n + x)) / ((double)(n1 - x + 1) * (double)(n - x + 1));
}

__global__ __launch_bounds__(64) void k_fisher(const int4* __restrict__ tables, int64_t M,
                                               double* __restrict__ p_out,
                                               double* __restrict__ or_out,
                                               uint2* __restrict__ crit) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= M) return;
  const int4 c = tables[idx];
  const int a = c.x, b = c.y, cc = c.z, d = c.w;
  const int n1 = a + b, n2 = cc + d, n = a + cc;
  if (n1 == 0 || n2 == 0 || n == 0 || b + d == 0) {
    p_out[idx] = 1.0;
    or_out[idx] = __longlong_as_double(0x7ff8000000000000LL);
    if (crit) crit[idx] = make_uint2(0u, 0u);
    return;
  }
  or_out[idx] = (cc > 0 && b > 0) ? ((double)a * (double)d) / ((double)cc * (double)b)
                                  : __longlong_as_double(0x7ff0000000000000LL);
  const int lo = max(0, n - n2), hi = min(n, n1);
  int mode = (int)(((double)(n + 1) * (double)(n1 + 1)) / (double)(n1 + n2 + 2));
  mode = min(max(mode, lo), hi);

  double w = 1.0;
  if (a > mode)
    for (int x = mode; x < a; ++x) w = w_up(w, x, n1, n2, n);
  else
    for (int x = mode; x > a; --x) w = w_down(w, x, n1, n2, n);
  const double thr = w * (1.0 + kTie);

  // Both walks stop once a term is inside the rejection region AND below
  // 2^-90 of the observed table's weight: what is left of the
  // (super-geometrically decaying) tail is < 1e-26 of the included sum, so p
  // keeps its RELATIVE accuracy even when it is 1e-200, and the region
  // boundary has already been passed, so (L, H) are exact.
  const double tiny = 8.077935669463161e-28 * (thr < 1.0 ? thr : 1.0);   // 2^-90 * w_obs
  double tot = 0.0, inc = 0.0;
  int H = hi + 1, L = lo - 1;
  w = 1.0;
  for (int x = mode; x <= hi; ++x) {
    tot += w;
    if (w <= thr) {
      inc += w;
      if (H > hi) H = x;
      if (w < tiny) break;
    }
    w = w_up(w, x, n1, n2, n);
  }
  w = 1.0;
  for (int x = mode; x > lo; --x) {
    w = w_down(w, x, n1, n2, n);  // weight of x-1
    tot += w;
    if (w <= thr) {
      inc += w;
      if (L < lo) L = x - 1;
      if (w < tiny) break;
    }
  }
  const bool all = (H == mode);
  const double p = all ? 1.0 : inc / tot;
  p_out[idx] = p < 1.0 ? p : 1.0;
  if (crit) crit[idx] = all ? make_uint2(0u, 0u) : make_uint2((uint32_t)(L + 1), (uint32_t)(H - L - 1));
}

// ----------------------------------------------------------------------------
// a8: label permutations (spec S4)
// ----------------------------------------------------------------------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply per product (v_mad_u64_u32) instead of hi + lo
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
    const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    const uint32_t n0 = h1 ^ c1 ^ k0, n2 = h0 ^ c3 ^ k1;
    c0 = n0;
    c1 = l1;
    c2 = n2;
    c3 = l0;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0;
  out[1] = c1;
  out[2] = c2;
  out[3] = c3;
}

// One thread per (trait, permutation): sequential selection sampling over the
// isolates, 32 at a time; the validity word is wave-uniform (blockIdx.y =
// trait), the 32-bit draws come from Philox keyed by (isolate>>2, pi, trait).
__global__ __launch_bounds__(64) void k_perm_generate(const uint32_t* __restrict__ masks,
                                                      const int32_t* __restrict__ margins, int N,
                                                      int Wp, int64_t P, int64_t perm_base,
                                                      int trait_base, uint32_t k0, uint32_t k1,
                                                      uint32_t* __restrict__ perms) {
  const int t = blockIdx.y;
  const int64_t pl = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pl >= P) return;
  const uint32_t pi = (uint32_t)(perm_base + pl);
  uint32_t needed = (uint32_t)margins[2 * t], remaining = (uint32_t)margins[2 * t + 1];
  const uint32_t* mrow = masks + (int64_t)t * Wp;
  uint32_t* out = perms + ((int64_t)t * P + pl) * Wp;
  const int nw = (N + 31) / 32;
  for (int k = 0; k < nw; ++k) {
    const uint32_t mw = mrow[k];
    uint32_t word = 0;
#pragma unroll 4
    for (int jj = 0; jj < 8; ++jj) {             // one Philox call = four isolates
      uint32_t r[4];
      philox4x32_10((uint32_t)(k * 8 + jj), pi, (uint32_t)(trait_base + t), kPermDomain, k0, k1, r);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int bit = 4 * jj + q;
        if ((mw >> bit) & 1u) {
          if (__umulhi(r[q], remaining) < needed) {
            word |= 1u << bit;
            --needed;
          }
          --remaining;
        }
      }
    }
    out[k] = word;
  }
  for (int k = nw; k < Wp; ++k) out[k] = 0u;
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



// ----------------------------------------------------------------------------
// a7 (list-driven variant): permutation exceedance counts from minority lists
// ----------------------------------------------------------------------------
// The dense kernel (k_permute_reg) pays 2 VALU ops per 32 isolates whatever the
// gene looks like.  Here the roles are swapped: a gene is the ascending list of
// isolates carrying its MINORITY value (scoary_lists_build), the permuted
// labels are stored isolate-major in tiles of LG*32 permutations that live in
// LDS, and a gene's overlap count with 32 permutations at once is a
// bit-sliced ("vertical") counter: every listed isolate adds one LDS row word
// into KC counter planes through v_bitop3 full adders (sum = a^b^c,
// carry = maj(a,b,c)).  Cost ~3.5 VALU ops per listed isolate per 32
// permutations instead of 2 ops per 32 isolates per permutation, i.e.
// ~0.11 * |list| ops per test versus 0.0625 * N * 2: a gene present in 26 %
// of 2000 isolates costs 57 ops per test instead of 137, a rare variant ~20x less.
// A wavefront = 64/LG lane groups = 64/LG genes of similar list length.
// LG = lanes (32-permutation words) per gene: 16 while a tile of 512
// permutations x (N+1) rows fits in LDS (N <= 2559), 8 (tiles of 256) up to
// N <= 5119, 4 (tiles of 128) up to N <= 10239.  One wavefront processes 64/LG
// genes; the 32/LG lane groups of a 32-lane half read LDS in lockstep.
// LDS row stride in dwords.  No padding: with a 16-dword stride a row starts at
// bank 0 or 16 by the PARITY of its isolate index, and the list builder orders
// the two genes that share a 32-lane half so that one walks its even rows while
// the other walks its odd rows (scoary_lists_build) -- conflict-free except where
// their even/odd counts differ.
__host__ __device__ constexpr int list_lg(int64_t N) {
  return N <= 2559 ? 16 : (N <= 5119 ? 8 : (N <= 10239 ? 4 : 0));
}
// index entries held per lane and step: LG = 16 spreads the 32 entries of a step
// over the group's 16 lanes (row_newbcast); LG = 8 / 4 give every QUAD of the
// group its own copy, 8 entries per lane (quad_perm broadcast)
__host__ __device__ constexpr int list_epl(int LG) { return LG == 16 ? 2 : 8; }
// dwords per label tile in HBM: rows 0..N plus padding to a 16-byte multiple
__host__ __device__ constexpr int64_t list_tile_dwords(int64_t N, int LG) {
  return ((N + 1) * LG + 3) / 4 * 4;
}

__device__ __forceinline__ uint32_t bit_xor3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x96" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t bit_maj(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xe8" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// majority(~a, b, c): the borrow of a - b - c
__device__ __forceinline__ uint32_t bit_majn(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x8e" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// Bit-sliced rejection-region test (spec S5) on KC counter planes: bit j of the
// result = ((count_j - base) mod 2^KD) < span.  Three LUT ops per plane.
template <int KC, int KD>
__device__ __forceinline__ uint32_t region_lt(const uint32_t (&c)[16], uint32_t base,
                                              uint32_t span) {
  uint32_t borrow = 0u, lt = 0u;   // d = u - base ; lt = (d < span)
#pragma unroll
  for (int k = 0; k < KD; ++k) {
    const uint32_t bk = (uint32_t)__builtin_amdgcn_sbfe((int)base, k, 1);   // 0 / ~0
    const uint32_t sk = (uint32_t)__builtin_amdgcn_sbfe((int)span, k, 1);
    uint32_t dk;
    if (k < KC) {
      dk = bit_xor3(c[k], bk, borrow);
      borrow = bit_majn(c[k], bk, borrow);
    } else {
      dk = bk ^ borrow;
      borrow |= bk;
    }
    lt = bit_majn(dk, sk, lt);
  }
  return lt;
}
// c += x + y at bit-plane weight 1: returns the carry (weight 2)
__device__ __forceinline__ uint32_t full_add(uint32_t& c, uint32_t x, uint32_t y) {
  const uint32_t carry = bit_maj(c, x, y);
  c = bit_xor3(c, x, y);
  return carry;
}

// Isolate-major label tiles: tiles[t][tile][row 0..N][LG] dwords, row N all
// zero; dword j of a row = labels of permutations tile*LG*32 + 32j .. +31.
// One wavefront generates 64 consecutive permutations (spec S4, same draws as
// k_perm_generate) and transposes them with ballots.
template <int LG>
__global__ __launch_bounds__(64) void k_perm_generate_tiles(const uint32_t* __restrict__ masks,
                                                            const int32_t* __restrict__ margins,
                                                            int N, int Wp, int64_t P,
                                                            int64_t perm_base, int trait_base,
                                                            uint32_t k0, uint32_t k1, int ntiles,
                                                            uint32_t* __restrict__ tiles) {
  const int t = blockIdx.y;
  const int lane = threadIdx.x;
  const int64_t wave = blockIdx.x;                    // 64 permutations each
  const int64_t pl = wave * kWave + lane;
  const bool live = pl < P;
  const uint32_t pi = (uint32_t)(perm_base + pl);
  const int waves_per_tile = LG / 2;
  const int tile = (int)(wave / waves_per_tile);
  const int col = (int)(wave % waves_per_tile) * 2;   // two dwords of each row
  uint32_t* base = tiles + (int64_t)(t * ntiles + tile) * list_tile_dwords(N, LG) + col;
  uint32_t needed = (uint32_t)margins[2 * t], remaining = (uint32_t)margins[2 * t + 1];
  const uint32_t* mrow = masks + (int64_t)t * Wp;
  const int nw = (N + 31) / 32;
  uint64_t mine = 0;
  for (int k = 0; k < nw; ++k) {
    const uint32_t mw = mrow[k];
#pragma unroll 2
    for (int jj = 0; jj < 8; ++jj) {
      uint32_t r[4];
      philox4x32_10((uint32_t)(k * 8 + jj), pi, (uint32_t)(trait_base + t), kPermDomain, k0, k1, r);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int bit = 4 * jj + q;
        bool sel = false;
        if ((mw >> bit) & 1u) {
          if (__umulhi(r[q], remaining) < needed) {
            sel = live;
            --needed;
          }
          --remaining;
        }
        const uint64_t b = __ballot(sel);
        const int iso = k * 32 + bit;
        if ((iso & 63) == lane) mine = b;
        if ((iso & 63) == 63 || iso == nw * 32 - 1) {  // 64 isolates collected: one row per lane
          const int row = (iso & ~63) + lane;
          if (row < N) {
            base[(int64_t)row * LG] = (uint32_t)mine;
            base[(int64_t)row * LG + 1] = (uint32_t)(mine >> 32);
          }
          mine = 0;
        }
      }
    }
  }
  if (lane == 0) {  // the all-zero row that list padding points at
    base[(int64_t)N * LG] = 0u;
    base[(int64_t)N * LG + 1] = 0u;
  }
}

// The same tiles from a workgroup per 64 permutations, for launches with fewer
// wavefronts than the chip has SIMDs (few traits x permutations, long rows): the
// sampling is serial over the isolates only in `needed` (two dependent VALU ops
// per isolate) while the Philox draws are not, so kGenProducers wavefronts
// compute the draws one 64-isolate chunk ahead into LDS (lane = permutation in
// every wavefront) and ONE selection wavefront walks the chunk; its compare
// mask over the 64 lanes IS the tile row of that isolate.
constexpr int kGenProducers = 7;
constexpr int kGenSplitBelow = 1;  // workgroup variant below this many wavefronts per SIMD
constexpr int kGenChunk = 64;      // isolates per LDS buffer = 16 Philox counters
// Philox counters (of the 16 per chunk) each producer wavefront computes.  A
// workgroup's wavefronts go to the SIMDs cyclically, so wavefront 4 shares the
// selection wavefront's SIMD and is given nothing.
__constant__ const int8_t kGenWork[kGenProducers][3] = {
    {0, 6, 12}, {1, 7, 13}, {2, 8, 14}, {-1, -1, -1}, {3, 9, 15}, {4, 10, -1}, {5, 11, -1}};

// v[LANE] = value (wave-uniform); hipcc has no builtin for v_writelane_b32, and
// its lane select must be an inline constant next to an SGPR value.
template <int LANE>
__device__ __forceinline__ void write_lane(uint32_t& v, uint32_t value) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(v) : "s"(value), "n"(LANE));
}
// Spec-S4 selection for isolates II.. of a chunk: u = the lane's draws, rem0 =
// valid isolates left at the chunk's start, mw = its validity bits (both
// wave-uniform; ALLVALID: mw is all ones), livemask = lanes whose permutation exists.
template <int II, bool ALLVALID>
__device__ __forceinline__ void select_rows(const uint32_t (&u)[kGenChunk], uint32_t rem0,
                                            uint64_t mw, uint64_t livemask, uint32_t& needed,
                                            uint32_t& lo, uint32_t& hi) {
  if constexpr (II < kGenChunk) {
    uint64_t b = 0;
    if (ALLVALID || ((mw >> II) & 1u)) {               // wave-uniform
      const uint32_t rem =
          ALLVALID ? rem0 - II : rem0 - (uint32_t)__popcll(mw & (((uint64_t)1 << II) - 1));
      const bool hit = __umulhi(u[II], rem) < needed;
      needed -= hit ? 1u : 0u;
      b = __builtin_amdgcn_ballot_w64(hit) & livemask;
    }
    write_lane<II>(lo, (uint32_t)b);                 // lane l: the row of isolate l of the chunk
    write_lane<II>(hi, (uint32_t)(b >> 32));
    select_rows<II + 1, ALLVALID>(u, rem0, mw, livemask, needed, lo, hi);
  }
}

template <int LG>
__global__ __launch_bounds__(kWave*(1 + kGenProducers)) void k_perm_generate_tiles_wg(
    const uint32_t* __restrict__ masks, const int32_t* __restrict__ margins, int N, int Wp,
    int64_t P, int64_t perm_base, int trait_base, uint32_t k0, uint32_t k1, int ntiles,
    uint32_t* __restrict__ tiles) {
  __shared__ uint32_t draws[2][kGenChunk][kWave];
  const int t = blockIdx.y;
  const int lane = threadIdx.x & (kWave - 1), role = threadIdx.x / kWave;   // 0: selection
  const int64_t wave = blockIdx.x;                    // 64 permutations each
  const int64_t pl = wave * kWave + lane;
  const bool live = pl < P;
  const uint32_t pi = (uint32_t)(perm_base + pl);
  const int waves_per_tile = LG / 2;
  const int tile = (int)(wave / waves_per_tile);
  const int col = (int)(wave % waves_per_tile) * 2;   // two dwords of each row
  uint32_t* base = tiles + (int64_t)(t * ntiles + tile) * list_tile_dwords(N, LG) + col;
  uint32_t needed = (uint32_t)margins[2 * t];
  uint32_t remaining = (uint32_t)__builtin_amdgcn_readfirstlane(margins[2 * t + 1]);
  const uint32_t* mrow = masks + (int64_t)t * Wp;     // Wp >= 2*nchunks words, zero padded
  const int nchunks = (N + kGenChunk - 1) / kGenChunk;
  const uint64_t livemask = __builtin_amdgcn_ballot_w64(live);
  if (role == 0) __builtin_amdgcn_s_setprio(3);       // the serial wavefront goes first
  for (int c = 0; c <= nchunks; ++c) {
    if (role > 0) {
      if (c < nchunks) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
          const int jj = kGenWork[role - 1][k];
          if (jj < 0) break;
          uint32_t r[4];
          philox4x32_10((uint32_t)(c * (kGenChunk / 4) + jj), pi, (uint32_t)(trait_base + t),
                        kPermDomain, k0, k1, r);
#pragma unroll
          for (int q = 0; q < 4; ++q) draws[c & 1][4 * jj + q][lane] = r[q];
        }
      }
    } else if (c > 0) {
      const int cc = c - 1;
      const uint64_t mw = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(mrow[2 * cc]) |
                          (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane(mrow[2 * cc + 1]) << 32;
      const uint32_t rem0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)remaining);
      uint32_t u[kGenChunk];
#pragma unroll
      for (int ii = 0; ii < kGenChunk; ++ii) u[ii] = draws[cc & 1][ii][lane];
      uint32_t lo = 0u, hi = 0u;
      if (mw == ~(uint64_t)0)
        select_rows<0, true>(u, rem0, mw, livemask, needed, lo, hi);
      else
        select_rows<0, false>(u, rem0, mw, livemask, needed, lo, hi);
      remaining -= (uint32_t)__popcll(mw);
      const int row = cc * kGenChunk + lane;
      if (row < N) {
        base[(int64_t)row * LG] = lo;
        base[(int64_t)row * LG + 1] = hi;
      }
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {  // the all-zero row that list padding points at
    base[(int64_t)N * LG] = 0u;
    base[(int64_t)N * LG + 1] = 0u;
  }
}

// Per (trait, list slot): the rejection region in terms of the LIST count u
// (u = a for a ones-list, npos - a for a zeros-list), modulo M = 2^KD:
//   in region  <=>  always | ((((u - base) mod M) >= span) ^ invert)
// out[t][slot] = { base, span | invert << 30 | always << 31 }.
__global__ __launch_bounds__(256) void k_lists_crit(const uint2* __restrict__ crit,
                                                    const int32_t* __restrict__ margins,
                                                    const int32_t* __restrict__ order,
                                                    const uint8_t* __restrict__ flipped, int G,
                                                    int KD, uint2* __restrict__ out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int t = blockIdx.y;
  if (k >= G) return;
  const int g = order[k];
  const uint2 c = crit[(int64_t)t * G + g];
  const uint32_t M = 1u << KD;
  uint2 o;
  if (c.y == 0u) {
    o = make_uint2(0u, 1u << 31);
  } else if (!flipped[g]) {
    o = make_uint2(c.x & (M - 1), c.y);
  } else {
    const uint32_t npos = (uint32_t)margins[2 * t];
    o = make_uint2((npos - c.x + 1u) & (M - 1), (M - c.y) | (1u << 30));
  }
  out[(int64_t)t * G + k] = o;
}

// Entry J (0..31) of a gene's 32-entry index vector -> every lane of its group,
// plus the lane's column offset, as ONE v_add_u32_dpp:
//   LG = 16: a DPP row is a group; entries sit two per lane -> row_newbcast:J/2
//   LG < 16: every quad of the group holds all 32 entries, eight per lane
//            -> quad_perm:[s,s,s,s] with s = J/8
template <int LG, int J>
__device__ __forceinline__ uint32_t entry_addr(const uint32_t (&e)[list_epl(LG)], uint32_t col4) {
  constexpr int EPL = list_epl(LG);
  const int v = (int)e[J % EPL];
  if constexpr (LG == 16) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, v, 0x150 + J / EPL, 0xf, 0xf, false) + col4;
  } else {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, v, (J / EPL) * 0x55, 0xf, 0xf, false) + col4;
  }
}
__device__ __forceinline__ uint32_t lds_at(const uint32_t* lds, uint32_t byte_off) {
  return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(lds) + byte_off);
}

// Issue the 8 LDS reads of entries 8J..8J+7 into x[8J..8J+7].
template <int LG, int J>
__device__ __forceinline__ void read8(uint32_t (&x)[32], const uint32_t* __restrict__ lds,
                                      const uint32_t (&e)[list_epl(LG)], uint32_t col4) {
  x[8 * J + 0] = lds_at(lds, entry_addr<LG, 8 * J + 0>(e, col4));
  x[8 * J + 1] = lds_at(lds, entry_addr<LG, 8 * J + 1>(e, col4));
  x[8 * J + 2] = lds_at(lds, entry_addr<LG, 8 * J + 2>(e, col4));
  x[8 * J + 3] = lds_at(lds, entry_addr<LG, 8 * J + 3>(e, col4));
  x[8 * J + 4] = lds_at(lds, entry_addr<LG, 8 * J + 4>(e, col4));
  x[8 * J + 5] = lds_at(lds, entry_addr<LG, 8 * J + 5>(e, col4));
  x[8 * J + 6] = lds_at(lds, entry_addr<LG, 8 * J + 6>(e, col4));
  x[8 * J + 7] = lds_at(lds, entry_addr<LG, 8 * J + 7>(e, col4));
}
// 8 row words -> counter planes 0..2, returns the carry of weight 8
__device__ __forceinline__ uint32_t sum8(uint32_t (&c)[16], const uint32_t* x) {
  const uint32_t a1 = full_add(c[0], x[0], x[1]);
  const uint32_t a2 = full_add(c[0], x[2], x[3]);
  const uint32_t b1 = full_add(c[1], a1, a2);
  const uint32_t a3 = full_add(c[0], x[4], x[5]);
  const uint32_t a4 = full_add(c[0], x[6], x[7]);
  const uint32_t b2 = full_add(c[1], a3, a4);
  return full_add(c[2], b1, b2);
}

template <int LG, int KC, int KD>
__global__ __launch_bounds__(1024) void k_permute_lists(const uint32_t* __restrict__ tiles,
                                                        const uint32_t* __restrict__ lidx,
                                                        const int32_t* __restrict__ lstart,
                                                        const int32_t* __restrict__ lngroups,
                                                        const int32_t* __restrict__ lorder,
                                                        const uint2* __restrict__ lcrit, int G,
                                                        int N, int64_t P, int ntiles,
                                                        int quads_per_block,
                                                        uint32_t* __restrict__ r) {
  extern __shared__ __attribute__((aligned(16))) uint32_t tile_lds[];
  // blockIdx.x = (trait, tile) fastest, blockIdx.y = gene chunk: the blocks that
  // run together walk the SAME chunk of index lists against different label
  // tiles, so the lists stream from HBM once and are re-read from L2.
  const int t = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int lg = lane / LG, col = lane % LG;
  constexpr int GPW = kWave / LG;   // genes per wavefront
  constexpr int EPL = list_epl(LG); // index entries per lane and step
  constexpr int LPS = 32 / EPL;     // lanes that together hold one step's 32 entries

  // tile -> LDS (contiguous copy, 16 B per lane)
  const int tile_dwords = (N + 1) * LG;
  const uint32_t* src = tiles + (int64_t)(t * ntiles + tile) * list_tile_dwords(N, LG);
  {
    const uint4* src4 = reinterpret_cast<const uint4*>(src);   // tiles are 16-B aligned per tile
    uint4* dst4 = reinterpret_cast<uint4*>(tile_lds);
    const int n4 = tile_dwords / 4;
    for (int i = tid; i < n4; i += blockDim.x) dst4[i] = src4[i];
    for (int i = n4 * 4 + tid; i < tile_dwords; i += blockDim.x) tile_lds[i] = src[i];
  }
  __syncthreads();

  // permutations of this lane's word that exist (the last tile may be ragged)
  const int64_t p_first = ((int64_t)tile * LG + col) * 32;
  const uint32_t valid = p_first >= P ? 0u : (P - p_first >= 32 ? 0xffffffffu : ((1u << (P - p_first)) - 1u));

  const int nquads = (G + GPW - 1) / GPW;
  const int q_lo = blockIdx.y * quads_per_block;
  const int q_hi = min(nquads, q_lo + quads_per_block);
  for (int q = q_lo + wave; q < q_hi; q += nwaves) {
    const int slot = min(q * GPW + lg, G - 1);
    const bool have = q * GPW + lg < G;
    // every gene of a quad has the same (padded) number of 32-entry groups
    const int nsuper = __builtin_amdgcn_readfirstlane(lngroups[q * GPW]);
    // 32 list entries (byte offsets of LDS rows) per step, EPL per lane: the
    // group's 16 lanes (LG = 16) or each of its quads (LG < 16) hold all 32
    struct alignas(EPL == 2 ? 8 : 16) Ent { uint32_t e[EPL]; };
    const Ent* lp = reinterpret_cast<const Ent*>(lidx) + (int64_t)lstart[slot] * LPS +
                    (LG == 16 ? col : (lane & 3));
    const uint32_t col4 = (uint32_t)col * 4u;

    uint32_t c[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c[k] = 0u;
    auto read32 = [&](uint32_t (&x)[32], const Ent& ix) {
      read8<LG, 0>(x, tile_lds, ix.e, col4);
      read8<LG, 1>(x, tile_lds, ix.e, col4);
      read8<LG, 2>(x, tile_lds, ix.e, col4);
      read8<LG, 3>(x, tile_lds, ix.e, col4);
    };
    auto sum32 = [&](const uint32_t (&x)[32]) -> uint32_t {
      const uint32_t cA = sum8(c, x);
      const uint32_t cB = sum8(c, x + 8);
      const uint32_t e1 = full_add(c[3], cA, cB);               // weight 16
      const uint32_t cC = sum8(c, x + 16);
      const uint32_t cD = sum8(c, x + 24);
      const uint32_t e2 = full_add(c[3], cC, cD);
      return full_add(c[4], e1, e2);                            // weight 32
    };
    // Software pipeline: index vectors are fetched four steps ahead (2 VGPRs per
    // step), the 32 LDS row reads of step s+1 are in flight while step s is summed.
    const int last = max(nsuper - 1, 0);
    Ent b0 = lp[0], b1 = lp[(int64_t)min(1, last) * LPS], b2 = lp[(int64_t)min(2, last) * LPS],
        b3 = lp[(int64_t)min(3, last) * LPS];
    uint32_t xa[32], xb[32];
    if (nsuper > 0) read32(xa, b0);
    for (int sg = 0; sg < nsuper; sg += 4) {
      // four steps (128 rows) per trip; their weight-32 carries are paired up
      // the tree before the (short) half-adder ripple
      uint32_t f1 = 0u, f2 = 0u, f3 = 0u;
      b0 = lp[(int64_t)min(sg + 4, last) * LPS];
      if (sg + 1 < nsuper) read32(xb, b1);
      const uint32_t f0 = sum32(xa);
      b1 = lp[(int64_t)min(sg + 5, last) * LPS];
      if (sg + 2 < nsuper) read32(xa, b2);
      if (sg + 1 < nsuper) f1 = sum32(xb);
      b2 = lp[(int64_t)min(sg + 6, last) * LPS];
      if (sg + 3 < nsuper) read32(xb, b3);
      if (sg + 2 < nsuper) f2 = sum32(xa);
      b3 = lp[(int64_t)min(sg + 7, last) * LPS];
      if (sg + 4 < nsuper) read32(xa, b0);
      if (sg + 3 < nsuper) f3 = sum32(xb);
      const uint32_t g0 = full_add(c[5], f0, f1);               // weight 64
      const uint32_t g1 = full_add(c[5], f2, f3);
      uint32_t carry = full_add(c[6], g0, g1);                  // weight 128
#pragma unroll
      for (int k = 7; k < KC; ++k) {                            // ripple (half adders)
        const uint32_t nc = c[k] & carry;
        c[k] ^= carry;
        carry = nc;
      }
    }
    // region test, bit-sliced against this lane group's constants
    const uint2 cr = lcrit[(int64_t)t * G + slot];
    const uint32_t base = cr.x, span = cr.y & 0x3fffffffu;
    const uint32_t inv = (cr.y >> 30) & 1u ? 0xffffffffu : 0u;
    const uint32_t always = (cr.y >> 31) ? 0xffffffffu : 0u;
    const uint32_t lt = region_lt<KC, KD>(c, base, span);
    uint32_t ex = ((~lt) ^ inv) | always;
    ex &= valid;
    int cnt = have ? __popc(ex) : 0;
#pragma unroll
    for (int off = LG / 2; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (have && col == 0 && cnt) atomicAdd(&r[(int64_t)t * G + lorder[slot]], (uint32_t)cnt);
  }
}



// 128 permutations per lane: 4 lanes per gene read the 16-dword tile rows with
// ds_read_b128, 16 genes per wavefront.  One address add serves four words
// (~2.2 VALU ops per listed isolate and 32 permutations).  ds_read_b128 is served
// in the lane groups {0-3,12-15,20-27} ...: with slot k starting at residue
// class k mod 4 (scoary_lists_build) each group's four genes sit on distinct
// 64-byte bank slots.
__device__ __forceinline__ uint4 lds128_at(const uint32_t* lds, uint32_t byte_off) {
  return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(lds) + byte_off);
}
struct Rows4 { uint32_t w0[4], w1[4], w2[4], w3[4]; };   // 4 tile rows x 4 permutation words
// The 4 entries held by lane H of every LPG-lane gene group -> 4 ds_read_b128.
template <int LPG, int H>
__device__ __forceinline__ void read4x4(Rows4& x, const uint32_t* __restrict__ lds,
                                        const uint32_t (&e)[4], uint32_t colb) {
  // quad_perm broadcast of lane H of the group: [H,H,H,H] or [H,H,2+H,2+H]
  constexpr int kCtrl = LPG == 4 ? H * 0x55 : 0xA0 + H * 0x55;
#define RD(J)                                                                                      \
  {                                                                                                \
    uint32_t a;                                                                                    \
    if constexpr (LPG == 1)                                                                        \
      a = e[J];                                                                                    \
    else                                                                                           \
      a = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)e[J], kCtrl, 0xf, 0xf, false) + colb;      \
    const uint4 v = lds128_at(lds, a);                                                             \
    x.w0[J] = v.x;                                                                                 \
    x.w1[J] = v.y;                                                                                 \
    x.w2[J] = v.z;                                                                                 \
    x.w3[J] = v.w;                                                                                 \
  }
  RD(0) RD(1) RD(2) RD(3)
#undef RD
}
// 4 row words -> counter planes 0..1, returns the carry of weight 4
__device__ __forceinline__ uint32_t sum4(uint32_t (&c)[16], const uint32_t (&x)[4]) {
  const uint32_t a1 = full_add(c[0], x[0], x[1]);
  const uint32_t a2 = full_add(c[0], x[2], x[3]);
  return full_add(c[1], a1, a2);
}
struct Carry4 { uint32_t w[4]; };

// LPG lanes per gene (4, 2, 1 for 16-, 8-, 4-dword tile rows), 64/LPG genes per
// wavefront.  Lists are walked in sub-steps of 4 entries: lane j of a gene group
// holds entries 4j..4j+3 of each 4*LPG-entry piece.
template <int LPG, int KC, int KD>
__global__ __launch_bounds__(1024) void k_permute_lists128(const uint32_t* __restrict__ tiles,
                                                           const uint32_t* __restrict__ lidx,
                                                           const int32_t* __restrict__ lstart,
                                                           const int32_t* __restrict__ lngroups,
                                                           const int32_t* __restrict__ lorder,
                                                           const uint2* __restrict__ lcrit, int G,
                                                           int N, int64_t P, int ntiles,
                                                           int quads_per_block,
                                                           uint32_t* __restrict__ r) {
  extern __shared__ __attribute__((aligned(16))) uint32_t tile_lds[];
  constexpr int TW = 4 * LPG;        // tile row, dwords
  constexpr int GPW = kWave / LPG;   // genes per wavefront
  const int t = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nwaves = blockDim.x >> 6;
  const int lg = lane / LPG, col = lane % LPG;

  const int tile_dwords = (N + 1) * TW;
  const uint32_t* src = tiles + (int64_t)(t * ntiles + tile) * list_tile_dwords(N, TW);
  {
    const uint4* src4 = reinterpret_cast<const uint4*>(src);
    uint4* dst4 = reinterpret_cast<uint4*>(tile_lds);
    const int n4 = tile_dwords / 4;
    for (int i = tid; i < n4; i += blockDim.x) dst4[i] = src4[i];
  }
  __syncthreads();

  uint32_t valid[4];
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    const int64_t p_first = ((int64_t)tile * TW + 4 * col + w) * 32;
    valid[w] = p_first >= P ? 0u : (P - p_first >= 32 ? 0xffffffffu : ((1u << (P - p_first)) - 1u));
  }

  const int nquads = (G + GPW - 1) / GPW;
  const int q_lo = blockIdx.y * quads_per_block;
  const int q_hi = min(nquads, q_lo + quads_per_block);
  for (int q = q_lo + wave; q < q_hi; q += nwaves) {
    const int slot = min(q * GPW + lg, G - 1);
    const bool have = q * GPW + lg < G;
    const int nsuper = __builtin_amdgcn_readfirstlane(lngroups[q * GPW]);   // 32-entry steps
    // interleaved lists (piece = TW entries): the wavefront's 64 lanes read 64
    // consecutive 16-byte index vectors per piece
    struct alignas(16) Ent { uint32_t e[4]; };
    const Ent* lp = reinterpret_cast<const Ent*>(lidx) +
                    (int64_t)__builtin_amdgcn_readfirstlane(lstart[q * GPW]) * 8 + lane;
    const uint32_t colb = (uint32_t)col * 16u;

    uint32_t c0[16], c1[16], c2[16], c3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c0[k] = c1[k] = c2[k] = c3[k] = 0u;
    Rows4 xa, xb;
    auto s4 = [&](const Rows4& x) -> Carry4 {
      Carry4 b;
      b.w[0] = sum4(c0, x.w0);
      b.w[1] = sum4(c1, x.w1);
      b.w[2] = sum4(c2, x.w2);
      b.w[3] = sum4(c3, x.w3);
      return b;
    };
    auto fa = [&](int plane, const Carry4& a, const Carry4& b) -> Carry4 {
      Carry4 o;
      o.w[0] = full_add(c0[plane], a.w[0], b.w[0]);
      o.w[1] = full_add(c1[plane], a.w[1], b.w[1]);
      o.w[2] = full_add(c2[plane], a.w[2], b.w[2]);
      o.w[3] = full_add(c3[plane], a.w[3], b.w[3]);
      return o;
    };
    // pieces of 4*LPG entries; the index loads run three pieces ahead.  Reads past
    // the end of the list re-read its last piece (valid rows, never summed).
    const int last = max(nsuper * (8 / LPG) - 1, 0);
    int piece = 0;
    Ent cur = lp[0], nxt = lp[(int64_t)min(1, last) * kWave], nn = lp[(int64_t)min(2, last) * kWave];
    read4x4<LPG, 0>(xa, tile_lds, cur.e, colb);
    // sub-step S of a step: issue the reads of sub-step S+1 into `other`, sum `mine`
#define SUBSTEP(S, MINE, OTHER)                                        \
  [&]() -> Carry4 {                                                    \
    constexpr int Hn = ((S) + 1) % LPG;                                \
    if constexpr (Hn == 0) {                                           \
      cur = nxt;                                                       \
      nxt = nn;                                                        \
      ++piece;                                                         \
      nn = lp[(int64_t)min(piece + 2, last) * kWave];                  \
    }                                                                  \
    read4x4<LPG, Hn>(OTHER, tile_lds, cur.e, colb);                    \
    return s4(MINE);                                                   \
  }()
    for (int sg = 0; sg < nsuper; sg += 4) {
      const Carry4 zero = {{0u, 0u, 0u, 0u}};
      auto step = [&](int k) -> Carry4 {                    // 32 listed isolates
        if (sg + k >= nsuper) return zero;
        const Carry4 b0 = SUBSTEP(0, xa, xb);
        const Carry4 b1 = SUBSTEP(1, xb, xa);
        const Carry4 d0 = fa(2, b0, b1);
        const Carry4 b2 = SUBSTEP(2, xa, xb);
        const Carry4 b3 = SUBSTEP(3, xb, xa);
        const Carry4 d1 = fa(2, b2, b3);
        const Carry4 e0 = fa(3, d0, d1);
        const Carry4 b4 = SUBSTEP(4, xa, xb);
        const Carry4 b5 = SUBSTEP(5, xb, xa);
        const Carry4 d2 = fa(2, b4, b5);
        const Carry4 b6 = SUBSTEP(6, xa, xb);
        const Carry4 b7 = SUBSTEP(7, xb, xa);
        const Carry4 d3 = fa(2, b6, b7);
        const Carry4 e1 = fa(3, d2, d3);
        return fa(4, e0, e1);                               // weight 32
      };
      const Carry4 f0 = step(0), f1 = step(1);
      const Carry4 g0 = fa(5, f0, f1);
      const Carry4 f2 = step(2), f3 = step(3);
      const Carry4 g1 = fa(5, f2, f3);
      Carry4 carry = fa(6, g0, g1);
#pragma unroll
      for (int k = 7; k < KC; ++k) {
#define RIPPLE(C, W)                         \
  {                                          \
    const uint32_t nc = C[k] & carry.w[W];   \
    C[k] ^= carry.w[W];                      \
    carry.w[W] = nc;                         \
  }
        RIPPLE(c0, 0) RIPPLE(c1, 1) RIPPLE(c2, 2) RIPPLE(c3, 3)
#undef RIPPLE
      }
    }
#undef SUBSTEP
    const uint2 cr = lcrit[(int64_t)t * G + slot];
    const uint32_t base = cr.x, span = cr.y & 0x3fffffffu;
    const uint32_t inv = (cr.y >> 30) & 1u ? 0xffffffffu : 0u;
    const uint32_t always = (cr.y >> 31) ? 0xffffffffu : 0u;
    int cnt = 0;
    cnt += __popc((((~region_lt<KC, KD>(c0, base, span)) ^ inv) | always) & valid[0]);
    cnt += __popc((((~region_lt<KC, KD>(c1, base, span)) ^ inv) | always) & valid[1]);
    cnt += __popc((((~region_lt<KC, KD>(c2, base, span)) ^ inv) | always) & valid[2]);
    cnt += __popc((((~region_lt<KC, KD>(c3, base, span)) ^ inv) | always) & valid[3]);
    if (!have) cnt = 0;
#pragma unroll
    for (int off = LPG / 2; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (have && col == 0 && cnt) atomicAdd(&r[(int64_t)t * G + lorder[slot]], (uint32_t)cnt);
  }
}

// ----------------------------------------------------------------------------
// f-2: pairwise Hamming counts between rows (isolates x variable genes)
// ----------------------------------------------------------------------------
// Same shape as k_counts: lane = row i (coalesced 16 B loads of the tiled
// matrix), blockIdx.y = a group of TB rows j whose words are wave-uniform
// (scalar loads).  out is symmetric, so lane i stores out[j][i]: coalesced.
template <int TB>
__global__ __launch_bounds__(256) void k_hamming(const uint4* __restrict__ tiled,
                                                 const uint32_t* __restrict__ vec, int R, int Rp,
                                                 int Qp, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j0 = blockIdx.y * TB;
  const int Wp = Qp * 4;
  uint32_t acc[TB];
  const uint4* rowj[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j) {
    acc[j] = 0;
    rowj[j] = reinterpret_cast<const uint4*>(vec + (int64_t)min(j0 + j, R - 1) * Wp);
  }
  for (int q = 0; q < Qp; ++q) {
    const uint4 gw = tiled[(int64_t)q * Rp + i];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const uint4 s = rowj[j][q];
      bcnt_acc(acc[j], gw.x ^ s.x);
      bcnt_acc(acc[j], gw.y ^ s.y);
      bcnt_acc(acc[j], gw.z ^ s.z);
      bcnt_acc(acc[j], gw.w ^ s.w);
    }
  }
  if (i >= R) return;
#pragma unroll
  for (int j = 0; j < TB; ++j)
    if (j0 + j < R) out[(int64_t)(j0 + j) * R + i] = (int32_t)acc[j];
}

// out[r] bit k = rows[r] bit index[k]
__global__ __launch_bounds__(256) void k_gather_bits(const uint32_t* __restrict__ rows, int64_t R,
                                                     int64_t Wsrc, const int32_t* __restrict__ index,
                                                     int64_t K, int64_t Wout,
                                                     uint32_t* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t w = blockIdx.y;
  if (r >= R) return;
  const uint32_t* row = rows + r * Wsrc;
  uint32_t word = 0;
  for (int b = 0; b < 32; ++b) {
    const int64_t k = w * 32 + b;
    if (k < K) {
      const int src = index[k];
      word |= ((row[src >> 5] >> (src & 31)) & 1u) << b;
    }
  }
  out[r * Wout + w] = word;
}

// ----------------------------------------------------------------------------
// f-4: presence-pattern hash for --collapse
// ----------------------------------------------------------------------------
// Two independent 64-bit multiply-xorshift chains over the masked words of a
// gene row (lane = gene, mask words wave-uniform).
__device__ __forceinline__ uint64_t mix64(uint64_t h, uint32_t w, uint64_t k) {
  h ^= (uint64_t)w + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  h *= k;
  return h ^ (h >> 29);
}

__global__ __launch_bounds__(256) void k_row_hash(const uint4* __restrict__ tiled,
                                                  const uint32_t* __restrict__ masks, int G, int Gp,
                                                  int Qp, uint64_t* __restrict__ out) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int t = blockIdx.y;
  const uint4* mrow = reinterpret_cast<const uint4*>(masks + (int64_t)t * Qp * 4);
  uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull;
  for (int q = 0; q < Qp; ++q) {
    const uint4 gw = tiled[(int64_t)q * Gp + g];
    const uint4 m = mrow[q];
    const uint32_t w[4] = {gw.x & m.x, gw.y & m.y, gw.z & m.z, gw.w & m.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h0 = mix64(h0, w[j], 0xBF58476D1CE4E5B9ull);
      h1 = mix64(h1, w[j] ^ 0x5bd1e995u, 0x94D049BB133111EBull);
    }
  }
  if (g < G) {
    out[((int64_t)t * G + g) * 2] = h0;
    out[((int64_t)t * G + g) * 2 + 1] = h1;
  }
}

// ----------------------------------------------------------------------------
// f-1: maximum contrasting pairs on a tree (PhyloTree, scoary/classes.py:199-592)
// ----------------------------------------------------------------------------
// State index 0 = AB, 1 = Ab, 2 = aB, 3 = ab, 4 = "0" (no free path); per state
// (total, supporting, opposing) pairs, or "unreachable".
//
// The reference keeps, per state, the best total and -- among the candidate
// pairings that reach it -- the best supporting and the best opposing count,
// each maximised on its own (classes.py:407-453).  That rule is exactly an
// integer max over two packed keys
//     ks = total << 16 | supporting      ko = total << 16 | opposing
// (counts <= tips/2 < 2^15, so adding two keys never carries between fields),
// with "unreachable" = a large negative key that stays negative through one
// addition and is re-clamped after every merge.  A pairing of a left and a
// right state is then TWO integer adds, choosing between pairings TWO v_max.
struct TreeNode {
  int ks[5], ko[5];
};
constexpr int kTreeNone = -(1 << 30);

__device__ __forceinline__ void tree_tip(TreeNode& n, int state) {
#pragma unroll
  for (int c = 0; c < 5; ++c) n.ks[c] = n.ko[c] = (c == state) ? 0 : kTreeNone;
}

// Generic node merge (classes.py:268-572).  For a free state c the nine
// candidates of the reference are {L[c]} x {all five R states} and
// {the four L states other than c} x {R[c]}; max distributes over +, so
//   out[c] = max( L[c] + max(R[0..4]),  max(L[x], x != c) + R[c] ).
// "No free path": both closed, or one new pair across the root (+1 total and
// +1 supporting for AB|ab, +1 opposing for Ab|aB).  ~80 VALU ops.
__device__ __forceinline__ void tree_merge(const TreeNode& L, const TreeNode& R, TreeNode& out) {
  int rs = R.ks[0], ro = R.ko[0];
#pragma unroll
  for (int x = 1; x < 5; ++x) {
    rs = max(rs, R.ks[x]);
    ro = max(ro, R.ko[x]);
  }
  // max of L over the states below / above c
  int pre_s[5], pre_o[5], suf_s[5], suf_o[5];
  pre_s[0] = pre_o[0] = kTreeNone;
#pragma unroll
  for (int x = 1; x < 5; ++x) {
    pre_s[x] = max(pre_s[x - 1], L.ks[x - 1]);
    pre_o[x] = max(pre_o[x - 1], L.ko[x - 1]);
  }
  suf_s[4] = suf_o[4] = kTreeNone;
#pragma unroll
  for (int x = 3; x >= 0; --x) {
    suf_s[x] = max(suf_s[x + 1], L.ks[x + 1]);
    suf_o[x] = max(suf_o[x + 1], L.ko[x + 1]);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int es = max(pre_s[c], suf_s[c]), eo = max(pre_o[c], suf_o[c]);
    out.ks[c] = max(max(L.ks[c] + rs, es + R.ks[c]), kTreeNone);
    out.ko[c] = max(max(L.ko[c] + ro, eo + R.ko[c]), kTreeNone);
  }
  constexpr int kPair = 1 << 16;
  int ns = L.ks[4] + R.ks[4], no = L.ko[4] + R.ko[4];
  ns = max(ns, L.ks[0] + R.ks[3] + (kPair + 1));
  no = max(no, L.ko[0] + R.ko[3] + kPair);
  ns = max(ns, L.ks[3] + R.ks[0] + (kPair + 1));
  no = max(no, L.ko[3] + R.ko[0] + kPair);
  ns = max(ns, L.ks[1] + R.ks[2] + kPair);
  no = max(no, L.ko[1] + R.ko[2] + (kPair + 1));
  ns = max(ns, L.ks[2] + R.ks[1] + kPair);
  no = max(no, L.ko[2] + R.ko[1] + (kPair + 1));
  out.ks[4] = max(ns, kTreeNone);
  out.ko[4] = max(no, kTreeNone);
}

// Merge with a Tip of state s (one-hot node, classes.py:575-592): the candidate
// list collapses to
//   out[c] = L[c]                          for the free states c != s,
//   out[s] = best of ALL five states of L  (the tip supplies the free path),
//   out[4] = L[3 - s] + one new pair       (AB|ab supporting, Ab|aB opposing).
__device__ __forceinline__ void tree_merge_tip(const TreeNode& L, int s, TreeNode& out) {
  int as = L.ks[0], ao = L.ko[0];
#pragma unroll
  for (int x = 1; x < 5; ++x) {
    as = max(as, L.ks[x]);
    ao = max(ao, L.ko[x]);
  }
  const int cs = 3 - s;
  int ps = kTreeNone, po = kTreeNone;
#pragma unroll
  for (int x = 0; x < 4; ++x)
    if (x == cs) {
      ps = L.ks[x];
      po = L.ko[x];
    }
  const bool supporting = (s == 0) || (s == 3);
  constexpr int kPair = 1 << 16;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    out.ks[c] = (c == s) ? as : L.ks[c];
    out.ko[c] = (c == s) ? ao : L.ko[c];
  }
  out.ks[4] = max(ps + kPair + (supporting ? 1 : 0), kTreeNone);
  out.ko[4] = max(po + kPair + (supporting ? 0 : 1), kTreeNone);
}

// One thread per (gene row g, label row l).  The stack program is wave-uniform
// (scalar loads, uniform branches); the top of the stack lives in registers,
// deeper entries in LDS as int32 keys [depth][10][64 lanes].
template <bool EXCEED>
__global__ __launch_bounds__(64) void k_tree_dp(const int32_t* __restrict__ ops, int nops,
                                                const uint32_t* __restrict__ gene_bits,
                                                const uint32_t* __restrict__ label_bits, int64_t G,
                                                int64_t L, int Wt, const int32_t* __restrict__ obs,
                                                int32_t* __restrict__ out3,
                                                uint8_t* __restrict__ exceed) {
  extern __shared__ __attribute__((aligned(16))) int stack_lds[];
  const int lane = threadIdx.x;
  const int64_t id = (int64_t)blockIdx.x * kWave + lane;
  const bool live = id < G * L;
  const int64_t g = live ? id / L : 0, l = live ? id % L : 0;
  const uint32_t* grow = gene_bits + g * Wt;
  const uint32_t* lrow = label_bits + l * Wt;
  TreeNode top;
  tree_tip(top, 4);
  int sp = 0;  // entries below `top`
  int curw = -1;
  uint32_t gw = 0, lw = 0;
  for (int k = 0; k < nops; ++k) {
    const int op = ops[k];
    if (op == -1) {  // merge the two top entries
      TreeNode left;
      --sp;
#pragma unroll
      for (int f = 0; f < 5; ++f) {
        left.ks[f] = stack_lds[((sp * 10) + f) * kWave + lane];
        left.ko[f] = stack_lds[((sp * 10) + 5 + f) * kWave + lane];
      }
      TreeNode m;
      tree_merge(left, top, m);
      top = m;
    } else {
      const int tip = op >= 0 ? op : -2 - op;
      if ((tip >> 5) != curw) {
        curw = tip >> 5;
        gw = grow[curw];
        lw = lrow[curw];
      }
      const int state = (((gw >> (tip & 31)) & 1u) ? 0 : 2) + (((lw >> (tip & 31)) & 1u) ? 0 : 1);
      if (op >= 0) {  // push
        if (k > 0) {
#pragma unroll
          for (int f = 0; f < 5; ++f) {
            stack_lds[((sp * 10) + f) * kWave + lane] = top.ks[f];
            stack_lds[((sp * 10) + 5 + f) * kWave + lane] = top.ko[f];
          }
          ++sp;
        }
        tree_tip(top, state);
      } else {  // merge the top entry with a tip
        TreeNode m;
        tree_merge_tip(top, state, m);
        top = m;
      }
    }
  }
  if (!live) return;
  // three independent maxima over the five states (classes.py:246-249)
  int bt = -1, bp = -1, ba = -1;
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const bool ok = top.ks[c] >= 0;
    bt = max(bt, ok ? (top.ks[c] >> 16) : -1);
    bp = max(bp, ok ? (top.ks[c] & 0xffff) : -1);
    ba = max(ba, ok ? (top.ko[c] & 0xffff) : -1);
  }
  if (!EXCEED) {
    out3[id * 3 + 0] = bt;
    out3[id * 3 + 1] = bp;
    out3[id * 3 + 2] = ba;
  } else {
    const int ot = obs[g * 3], op_ = obs[g * 3 + 1], oa = obs[g * 3 + 2];
    const bool use_pro = op_ >= oa;
    const double est = (double)(use_pro ? op_ : oa) / (double)ot;
    const int x = use_pro ? bp : ba;
    exceed[id] = (bt > 0 && (double)x / (double)bt >= est) ? 1 : 0;
  }
}

}  // namespace

// ============================================================================
// Host side: context, error handling, launches
// ============================================================================
struct scoary_ctx {
  int device = 0;
  int num_cu = 256;
  std::string err;
  bool timing = false;
  int lists_lds_optin = 0;   // k_permute_lists variants (by LG) with the 160 KB LDS opt-in done
  struct Timed {
    std::string name;
    hipEvent_t start, stop;
  };
  std::vector<Timed> timed;
};

namespace {

int fail(scoary_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  return code;
}

#define HIP_TRY(h, expr)                                                              \
  do {                                                                                \
    hipError_t e_ = (expr);                                                           \
    if (e_ != hipSuccess)                                                             \
      return fail(h, SCOARY_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// Sets the handle's device for the duration of a call and restores the
// caller's (torch's) current device afterwards.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = (hipSetDevice(dev) == hipSuccess);
  }
  ~DeviceGuard() {
    if (switched) (void)hipSetDevice(prev);
  }
};

struct KernelTimer {
  scoary_handle h;
  hipStream_t s;
  hipEvent_t start = nullptr, stop = nullptr;
  KernelTimer(scoary_handle h_, hipStream_t s_, const char* name) : h(h_), s(s_) {
    if (!h->timing) return;
    if (hipEventCreate(&start) != hipSuccess || hipEventCreate(&stop) != hipSuccess) {
      start = stop = nullptr;
      return;
    }
    (void)hipEventRecord(start, s);
    h->timed.push_back({name, start, stop});
  }
  ~KernelTimer() {
    if (stop) (void)hipEventRecord(stop, s);
  }
};

template <int RQ, int GL>
void launch_permute_reg(dim3 grid, hipStream_t s, const uint32_t* tiled, const uint32_t* perms,
                        const uint32_t* crit, int G, int Gp, int64_t P, int pchunk, uint32_t* r) {
  hipLaunchKernelGGL((k_permute_reg<RQ, GL>), grid, dim3(kWave), 0, s,
                     reinterpret_cast<const uint4*>(tiled), perms,
                     reinterpret_cast<const uint2*>(crit), G, Gp, P, pchunk, r);
}

}  // namespace

extern "C" {

int scoary_abi_version(void) { return SCOARY_ABI_VERSION; }

int64_t scoary_tiled_quads(int64_t N) { return tiled_quads(N); }
int64_t scoary_tiled_genes(int64_t G) { return round_up(G < 1 ? 1 : G, kGeneAlign); }
int64_t scoary_tiled_bytes(int64_t G, int64_t N) {
  return 16 * scoary_tiled_quads(N) * scoary_tiled_genes(G);
}
int64_t scoary_row_words(int64_t N) { return 4 * scoary_tiled_quads(N); }

int scoary_create(int device, scoary_handle* out) {
  if (!out) return SCOARY_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SCOARY_ERR_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SCOARY_ERR_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::fprintf(stderr, "scoary_hip: device %d is %s; this library is built for gfx950 only\n",
                 device, prop.gcnArchName);
    return SCOARY_ERR_DEVICE;
  }
  scoary_ctx* h = new scoary_ctx();
  h->device = device;
  h->num_cu = prop.multiProcessorCount;
  *out = h;
  return SCOARY_OK;
}

void scoary_destroy(scoary_handle h) {
  if (!h) return;
  for (auto& t : h->timed) {
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  delete h;
}

const char* scoary_last_error(scoary_handle h) { return h ? h->err.c_str() : "null handle"; }

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

int scoary_counts(scoary_handle h, const uint32_t* d_tiled, const uint32_t* d_traits,
                  const uint32_t* d_masks, int64_t G, int64_t T, int64_t N, int32_t* d_counts,
                  int32_t* d_margins, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_traits || !d_masks || !d_counts || !d_margins || G < 1 || T < 1 || N < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_counts: bad argument");
  if (G > (int64_t)1 << 30 || T > 65535 * 8) return fail(h, SCOARY_ERR_SIZE, "scoary_counts: G or T too large");
  DeviceGuard guard(h->device);
  const int64_t Gp = scoary_tiled_genes(G), Qp = scoary_tiled_quads(N);
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    KernelTimer kt(h, s, "k_margins");
    hipLaunchKernelGGL(k_margins, dim3((unsigned)T), dim3(kWave), 0, s, d_traits, d_masks,
                       (int)(Qp * 4), d_margins);
  }
  constexpr int TB = 4;
  {
    KernelTimer kt(h, s, "k_counts");
    hipLaunchKernelGGL((k_counts<TB>), dim3((unsigned)(Gp / 256), (unsigned)((T + TB - 1) / TB)),
                       dim3(256), 0, s, reinterpret_cast<const uint4*>(d_tiled), d_traits, d_masks,
                       d_margins, (int)G, (int)Gp, (int)Qp, (int)T, reinterpret_cast<int4*>(d_counts));
  }
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int scoary_fisher(scoary_handle h, const int32_t* d_tables, int64_t M, double* d_p, double* d_or,
                  uint32_t* d_crit, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tables || !d_p || !d_or || M < 1) return fail(h, SCOARY_ERR_ARG, "scoary_fisher: bad argument");
  if ((M + kWave - 1) / kWave > 0x7fffffffLL) return fail(h, SCOARY_ERR_SIZE, "scoary_fisher: M too large");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_fisher");
  hipLaunchKernelGGL(k_fisher, dim3((unsigned)((M + kWave - 1) / kWave)), dim3(kWave), 0, s,
                     reinterpret_cast<const int4*>(d_tables), M, d_p, d_or,
                     reinterpret_cast<uint2*>(d_crit));
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int scoary_perm_generate(scoary_handle h, const uint32_t* d_masks, const int32_t* d_margins,
                         int64_t T, int64_t N, int64_t P, int64_t perm_base, int64_t trait_base,
                         uint64_t seed,
                         uint32_t* d_perms, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_masks || !d_margins || !d_perms || T < 1 || N < 1 || P < 1 || perm_base < 0 ||
      trait_base < 0)
    return fail(h, SCOARY_ERR_ARG, "scoary_perm_generate: bad argument");
  if (T > 65535 || trait_base + T > 0x7fffffffLL || perm_base + P > 0xffffffffLL)
    return fail(h, SCOARY_ERR_SIZE, "scoary_perm_generate: T > 65535 or permutation index >= 2^32");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_perm_generate");
  hipLaunchKernelGGL(k_perm_generate, dim3((unsigned)((P + kWave - 1) / kWave), (unsigned)T),
                     dim3(kWave), 0, s, d_masks, d_margins, (int)N, (int)scoary_row_words(N), P,
                     perm_base, (int)trait_base, (uint32_t)seed, (uint32_t)(seed >> 32), d_perms);
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

int scoary_hamming(scoary_handle h, const uint32_t* d_tiled, const uint32_t* d_vecrows, int64_t R,
                   int64_t N, int32_t* d_out, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_vecrows || !d_out || R < 1 || N < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_hamming: bad argument");
  if (R > (int64_t)1 << 20) return fail(h, SCOARY_ERR_SIZE, "scoary_hamming: more than 2^20 rows");
  DeviceGuard guard(h->device);
  const int64_t Rp = scoary_tiled_genes(R), Qp = scoary_tiled_quads(N);
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr int TB = 8;
  KernelTimer kt(h, s, "k_hamming");
  hipLaunchKernelGGL((k_hamming<TB>), dim3((unsigned)(Rp / 256), (unsigned)((R + TB - 1) / TB)),
                     dim3(256), 0, s, reinterpret_cast<const uint4*>(d_tiled), d_vecrows, (int)R,
                     (int)Rp, (int)Qp, d_out);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int scoary_gather_bits(scoary_handle h, const uint32_t* d_rows, int64_t R, int64_t Wsrc,
                       const int32_t* d_index, int64_t K, uint32_t* d_out,
                       scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_rows || !d_index || !d_out || R < 1 || Wsrc < 1 || K < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_gather_bits: bad argument");
  const int64_t Wout = (K + 31) / 32;
  if (Wout > 65535) return fail(h, SCOARY_ERR_SIZE, "scoary_gather_bits: K too large");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_gather_bits");
  hipLaunchKernelGGL(k_gather_bits, dim3((unsigned)((R + 255) / 256), (unsigned)Wout), dim3(256), 0,
                     s, d_rows, R, Wsrc, d_index, K, Wout, d_out);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

static int launch_tree(scoary_handle h, const char* what, bool exceed_mode, const int32_t* d_ops,
                       int64_t nops, int64_t stack_depth, const uint32_t* d_gene_bits,
                       const uint32_t* d_label_bits, int64_t G, int64_t L, int64_t K,
                       const int32_t* d_obs, int32_t* d_out3, uint8_t* d_exceed,
                       scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_ops || !d_gene_bits || !d_label_bits || nops < 1 || G < 1 || L < 1 || K < 1 ||
      stack_depth < 1 || (exceed_mode ? (!d_obs || !d_exceed) : !d_out3))
    return fail(h, SCOARY_ERR_ARG, std::string(what) + ": bad argument");
  if (stack_depth > 32) return fail(h, SCOARY_ERR_SIZE, std::string(what) + ": stack_depth > 32");
  if (K > 65534) return fail(h, SCOARY_ERR_SIZE, std::string(what) + ": more than 65534 tips");
  const int64_t threads = G * L;
  if ((threads + kWave - 1) / kWave > 0x7fffffffLL)
    return fail(h, SCOARY_ERR_SIZE, std::string(what) + ": G*L too large for one launch");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)stack_depth * 10 * kWave * sizeof(int);
  const int Wt = (int)((K + 31) / 32);
  dim3 grid((unsigned)((threads + kWave - 1) / kWave));
  KernelTimer kt(h, s, "k_tree_dp");
  if (exceed_mode)
    hipLaunchKernelGGL((k_tree_dp<true>), grid, dim3(kWave), lds, s, d_ops, (int)nops, d_gene_bits,
                       d_label_bits, G, L, Wt, d_obs, (int32_t*)nullptr, d_exceed);
  else
    hipLaunchKernelGGL((k_tree_dp<false>), grid, dim3(kWave), lds, s, d_ops, (int)nops, d_gene_bits,
                       d_label_bits, G, L, Wt, (const int32_t*)nullptr, d_out3, (uint8_t*)nullptr);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int scoary_tree_pairs(scoary_handle h, const int32_t* d_ops, int64_t nops, int64_t stack_depth,
                      const uint32_t* d_gene_bits, const uint32_t* d_label_bits, int64_t G,
                      int64_t L, int64_t K, int32_t* d_out, scoary_stream_t stream) {
  return launch_tree(h, "scoary_tree_pairs", false, d_ops, nops, stack_depth, d_gene_bits,
                     d_label_bits, G, L, K, nullptr, d_out, nullptr, stream);
}

int scoary_tree_permute(scoary_handle h, const int32_t* d_ops, int64_t nops, int64_t stack_depth,
                        const uint32_t* d_gene_bits, const uint32_t* d_label_bits, int64_t G,
                        int64_t L, int64_t K, const int32_t* d_obs, uint8_t* d_exceed,
                        scoary_stream_t stream) {
  return launch_tree(h, "scoary_tree_permute", true, d_ops, nops, stack_depth, d_gene_bits,
                     d_label_bits, G, L, K, d_obs, nullptr, d_exceed, stream);
}

int scoary_row_hash(scoary_handle h, const uint32_t* d_tiled, const uint32_t* d_masks, int64_t G,
                    int64_t T, int64_t N, uint64_t* d_out, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_masks || !d_out || G < 1 || T < 1 || N < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_row_hash: bad argument");
  if (T > 65535) return fail(h, SCOARY_ERR_SIZE, "scoary_row_hash: T > 65535");
  DeviceGuard guard(h->device);
  const int64_t Gp = scoary_tiled_genes(G), Qp = scoary_tiled_quads(N);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_row_hash");
  hipLaunchKernelGGL(k_row_hash, dim3((unsigned)(Gp / 256), (unsigned)T), dim3(256), 0, s,
                     reinterpret_cast<const uint4*>(d_tiled), d_masks, (int)G, (int)Gp, (int)Qp,
                     d_out);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int64_t scoary_list_tiles_words(int64_t N, int64_t P, int64_t T) {
  const int LG = list_lg(N);
  if (!LG) return 0;
  const int64_t tile_perms = LG * 32;
  const int64_t ntiles = (P + tile_perms - 1) / tile_perms;
  return T * ntiles * list_tile_dwords(N, LG);
}
int64_t scoary_list_tile_words(int64_t N) { return list_lg(N) ? list_tile_dwords(N, list_lg(N)) : 0; }
int64_t scoary_list_max_isolates(void) { return 10239; }
// Words (32 permutations each) per lane: 4 = k_permute_lists128 (the default), 1 =
// k_permute_lists, the ds_read_b32 kernel it replaced, kept for A/B measurements
// and selected with SCOARY_LISTS_WPL=1 in the environment.
static int lists_wpl(int64_t) {
  const char* e = std::getenv("SCOARY_LISTS_WPL");
  return e && e[0] == '1' && !e[1] ? 1 : 4;
}
int scoary_list_params(int64_t N, int64_t* out5) {
  if (!out5) return SCOARY_ERR_ARG;
  const int LG = list_lg(N);
  const int wpl = lists_wpl(N);
  out5[0] = LG;                            /* tile row width in dwords (0: N too large for LDS tiles) */
  out5[1] = LG * 4;                        /* LDS / tile row stride in bytes */
  out5[2] = LG ? kWave / LG * wpl : 0;     /* genes per wavefront: lists padded to equal length */
  out5[3] = LG ? (wpl == 4 ? 64 : 32) / LG : 0;   /* residue classes of the isolate index:
                                              ds_read_b128 is banked over 256 B, ds_read_b32 over 128 B */
  out5[4] = LG && wpl == 4 ? LG : 0;       /* interleave piece, entries (0: contiguous lists) */
  return LG ? SCOARY_OK : SCOARY_ERR_SIZE;
}

int scoary_perm_generate_tiles(scoary_handle h, const uint32_t* d_masks, const int32_t* d_margins,
                               int64_t T, int64_t N, int64_t P, int64_t perm_base,
                               int64_t trait_base, uint64_t seed, uint32_t* d_tiles,
                               scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_masks || !d_margins || !d_tiles || T < 1 || N < 1 || P < 1 || perm_base < 0 || trait_base < 0)
    return fail(h, SCOARY_ERR_ARG, "scoary_perm_generate_tiles: bad argument");
  if (T > 65535 || perm_base + P > 0xffffffffLL)
    return fail(h, SCOARY_ERR_SIZE, "scoary_perm_generate_tiles: T > 65535 or permutation index >= 2^32");
  const int LG = list_lg(N);
  if (!LG) return fail(h, SCOARY_ERR_SIZE, "scoary_perm_generate_tiles: N too large for LDS tiles");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int64_t tile_perms = LG * 32;
  const int64_t ntiles = (P + tile_perms - 1) / tile_perms;
  const dim3 grid((unsigned)(ntiles * (tile_perms / kWave)), (unsigned)T);
  KernelTimer kt(h, s, "k_perm_generate_tiles");
  // one wavefront per 64 permutations when there are enough of them to fill the
  // chip; otherwise a workgroup each, with the Philox draws spread over more lanes
  const bool wg = (int64_t)grid.x * grid.y < (int64_t)h->num_cu * 4 * kGenSplitBelow;
#define GEN_TILES(LGV)                                                                            \
  if (wg)                                                                                         \
    hipLaunchKernelGGL((k_perm_generate_tiles_wg<LGV>), grid, dim3(kWave * (1 + kGenProducers)),  \
                       0, s, d_masks, d_margins, (int)N, (int)scoary_row_words(N), P, perm_base,  \
                       (int)trait_base, (uint32_t)seed, (uint32_t)(seed >> 32), (int)ntiles,      \
                       d_tiles);                                                                  \
  else                                                                                            \
    hipLaunchKernelGGL((k_perm_generate_tiles<LGV>), grid, dim3(kWave), 0, s, d_masks, d_margins, \
                       (int)N, (int)scoary_row_words(N), P, perm_base, (int)trait_base,           \
                       (uint32_t)seed, (uint32_t)(seed >> 32), (int)ntiles, d_tiles)
  if (LG == 16) { GEN_TILES(16); } else if (LG == 8) { GEN_TILES(8); } else { GEN_TILES(4); }
#undef GEN_TILES
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

extern "C++" {
template <int LG, int KC, int KD, int WPL = 1>
static int launch_permute_lists(scoary_handle h, hipStream_t s, const uint32_t* d_tiles,
                                const uint32_t* d_lidx, const int32_t* d_lstart,
                                const int32_t* d_lngroups, const int32_t* d_lorder,
                                const uint8_t* d_lflipped, const uint32_t* d_crit,
                                const int32_t* d_margins, uint32_t* d_lcrit, int64_t G, int64_t T,
                                int64_t N, int64_t P, uint32_t* d_r) {
  {
    KernelTimer kt(h, s, "k_lists_crit");
    hipLaunchKernelGGL(k_lists_crit, dim3((unsigned)((G + 255) / 256), (unsigned)T), dim3(256), 0, s,
                       reinterpret_cast<const uint2*>(d_crit), d_margins, d_lorder, d_lflipped,
                       (int)G, KD, reinterpret_cast<uint2*>(d_lcrit));
  }
  const int64_t tile_perms = LG * 32;
  const int64_t ntiles = (P + tile_perms - 1) / tile_perms;
  constexpr int GPW = kWave / LG * WPL;
  const int64_t nquads = (G + GPW - 1) / GPW;
  // enough blocks for >= 16 rounds over the CUs, and gene chunks whose index
  // lists (~2 MB) stay in an XCD's 4 MB L2 while the (trait, tile) blocks of the
  // chunk run; each chunk a multiple of 16 wave-groups
  int64_t chunks = ((int64_t)h->num_cu * 16 + ntiles * T - 1) / (ntiles * T);
  const int64_t by_l2 = (G * N / 4 * 4 + (2 << 20) - 1) / (2 << 20);   // ~N/4 entries x 4 B per gene
  if (chunks < by_l2) chunks = by_l2;
  if (chunks > 65535) chunks = 65535;
  if (chunks < 1) chunks = 1;
  int64_t qpb = (nquads + chunks - 1) / chunks;
  qpb = (qpb + 15) / 16 * 16;
  chunks = (nquads + qpb - 1) / qpb;
  const size_t lds = (size_t)(N + 1) * LG * sizeof(uint32_t);
  constexpr int kFlag = WPL == 4 ? 64 * LG : LG;
  const void* fn;
  if constexpr (WPL == 4) fn = reinterpret_cast<const void*>(&k_permute_lists128<LG / 4, KC, KD>);
  else fn = reinterpret_cast<const void*>(&k_permute_lists<LG, KC, KD>);
  if (!(h->lists_lds_optin & kFlag)) {   // once per handle (= per device) and kernel variant
    HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    h->lists_lds_optin |= kFlag;
  }
  KernelTimer kt(h, s, "k_permute_lists");
  const dim3 grid((unsigned)(ntiles * T), (unsigned)chunks);
  const uint2* lcrit2 = reinterpret_cast<const uint2*>(d_lcrit);
  if constexpr (WPL == 4)
    hipLaunchKernelGGL((k_permute_lists128<LG / 4, KC, KD>), grid, dim3(1024), lds, s, d_tiles, d_lidx,
                       d_lstart, d_lngroups, d_lorder, lcrit2, (int)G, (int)N, P, (int)ntiles,
                       (int)qpb, d_r);
  else
    hipLaunchKernelGGL((k_permute_lists<LG, KC, KD>), grid, dim3(1024), lds, s, d_tiles, d_lidx,
                       d_lstart, d_lngroups, d_lorder, lcrit2, (int)G, (int)N, P, (int)ntiles,
                       (int)qpb, d_r);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}
}  // extern "C++"

int scoary_permute_lists(scoary_handle h, const uint32_t* d_tiles, const uint32_t* d_lidx,
                         const int32_t* d_lstart, const int32_t* d_lngroups,
                         const int32_t* d_lorder, const uint8_t* d_lflipped,
                         const uint32_t* d_crit, const int32_t* d_margins, uint32_t* d_lcrit,
                         int64_t G, int64_t T, int64_t N, int64_t P, uint32_t* d_r,
                         scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiles || !d_lidx || !d_lstart || !d_lngroups || !d_lorder || !d_lflipped || !d_crit ||
      !d_margins || !d_lcrit || !d_r || G < 1 || T < 1 || N < 1 || P < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_permute_lists: bad argument");
  const int LG = list_lg(N);
  if (!LG)
    return fail(h, SCOARY_ERR_SIZE, "scoary_permute_lists: label tile does not fit in LDS for this N");
  if (T > 65535) return fail(h, SCOARY_ERR_SIZE, "scoary_permute_lists: T > 65535");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // counter planes KC: lists hold <= N/2 entries; compare planes KD: 2N+3 <= 2^KD
  const int wpl = lists_wpl(N);
  if (LG == 16 && wpl == 4)
    return launch_permute_lists<16, 11, 13, 4>(h, s, d_tiles, d_lidx, d_lstart, d_lngroups, d_lorder,
                                               d_lflipped, d_crit, d_margins, d_lcrit, G, T, N, P, d_r);
  if (LG == 16)
    return launch_permute_lists<16, 11, 13>(h, s, d_tiles, d_lidx, d_lstart, d_lngroups, d_lorder,
                                            d_lflipped, d_crit, d_margins, d_lcrit, G, T, N, P, d_r);
  if (LG == 8 && wpl == 4)
    return launch_permute_lists<8, 12, 14, 4>(h, s, d_tiles, d_lidx, d_lstart, d_lngroups, d_lorder,
                                              d_lflipped, d_crit, d_margins, d_lcrit, G, T, N, P, d_r);
  if (LG == 4 && wpl == 4)
    return launch_permute_lists<4, 13, 15, 4>(h, s, d_tiles, d_lidx, d_lstart, d_lngroups, d_lorder,
                                              d_lflipped, d_crit, d_margins, d_lcrit, G, T, N, P, d_r);
  if (LG == 8)
    return launch_permute_lists<8, 12, 14>(h, s, d_tiles, d_lidx, d_lstart, d_lngroups, d_lorder,
                                           d_lflipped, d_crit, d_margins, d_lcrit, G, T, N, P, d_r);
  return launch_permute_lists<4, 13, 15>(h, s, d_tiles, d_lidx, d_lstart, d_lngroups, d_lorder,
                                         d_lflipped, d_crit, d_margins, d_lcrit, G, T, N, P, d_r);
}

int scoary_set_timing(scoary_handle h, int enabled) {
  if (!h) return SCOARY_ERR_ARG;
  for (auto& t : h->timed) {
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  h->timed.clear();
  h->timing = enabled != 0;
  return SCOARY_OK;
}

int scoary_last_kernel_ms(scoary_handle h, const char* kernel, double* ms_out) {
  if (!h || !kernel || !ms_out) return SCOARY_ERR_ARG;
  DeviceGuard guard(h->device);
  double total = 0.0;
  int n = 0;
  for (auto& t : h->timed) {
    if (t.name != kernel) continue;
    HIP_TRY(h, hipEventSynchronize(t.stop));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, t.start, t.stop));
    total += ms;
    ++n;
  }
  if (n == 0) return fail(h, SCOARY_ERR_ARG, std::string("no timed launches of ") + kernel);
  *ms_out = total / n;
  return SCOARY_OK;
}

}  // extern "C"
