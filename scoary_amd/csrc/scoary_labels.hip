// scoary_labels.hip -- a8: label permutations (PermuteGTC, scoary/methods.py:1371-1384),
// spec S4 of round 5: a sampler with NO dependence from one isolate to the next.
//
// Rounds 1-4 drew a permutation by sequential selection sampling: N dependent steps per
// permutation, a latency chain that every rank of a gene-sharded run repeated in full (the
// serial fraction of cfg4's strong split).  The law is the contract -- the
// reference's shuffle is unseeded -- so the bits were redefined (the
// CPU checker restates them independently):
//
//   round 0  every valid isolate is MARKED independently with probability q/256.  Thirty-two
//            permutations share eight Philox words per isolate and the 8-bit comparison runs
//            bit-sliced: lane = isolate, bit = permutation, one tile-row dword comes out of a
//            register -- no ballot, no v_writelane transposition;
//   counts   K = marks per permutation: bit-sliced per-lane counters, a butterfly over the
//            wavefront, one LDS atomic per wavefront and permutation;
//   fix-up   lane = permutation: |m - K| uniformly drawn positions are toggled (rejection on
//            non-candidates; Lemire rejection makes the position exactly uniform).  Conditional
//            on K the marked set is uniform among K-subsets, and adding / removing uniformly
//            chosen isolates keeps it uniform -- the result is exactly uniform under ideal
//            random words;
//   output   isolate-major LDS tiles (scoary_perm_generate_tiles) or permutation-major bit rows
//            (scoary_perm_generate, through 32 ballots per 64 isolates).
//
// A block = (trait, NB dword columns of 32 permutations); its rows live in LDS until the fix-up is
// done: N * NB * 4 bytes (NB = 1 or 2 by how many blocks the launch has).  Beyond N = 40 000 a
// row keeps only 16 / 8 / ... 1 of the 32 permutations of a dword (more Philox work per label,
// still parallel in N).
#include "scoary_common.hpp"

namespace {

constexpr uint32_t kDomBernLo = 0x53434F42u;   // "SCOB": words R_0..R_3 of round 0
constexpr uint32_t kDomBernHi = 0x53434F43u;   // "SCOC": words R_4..R_7
constexpr uint32_t kDomFix = 0x53434F44u;      // "SCOD": fix-up position draws
constexpr uint32_t kFixMaxCalls = 1u << 20;    // safety stop of the fix-up loop (the CPU checker has the same bound)
constexpr int kCntPlanes = 10;                 // per-lane bit-sliced mark count: <= 1023 rows per thread
constexpr int kSumPlanes = 16;                 // per-wavefront sum: <= 64 * 1023
constexpr int kMaxRowsPerThread = (1 << kCntPlanes) - 1;
constexpr int kLabelsMaxLds = 160 * 1024;

struct LabelPlan {
  uint32_t m, q;     // marks to place; 8-bit round-0 probability
  bool flip;         // the complement is emitted (npos > nval / 2)
};
__device__ __forceinline__ uint32_t isqrt64(uint64_t v) {
  uint64_t r = 0;
  for (int s = 31; s >= 0; --s) {
    const uint64_t c = r | ((uint64_t)1 << s);
    if (c * c <= v) r = c;
  }
  return (uint32_t)r;
}
// the plan of spec S4 -- wave-uniform integer arithmetic, the same in the CPU checker
__device__ __forceinline__ LabelPlan label_plan(int32_t npos_i, int32_t nval_i) {
  const int64_t npos = npos_i, nval = nval_i;
  LabelPlan p;
  p.flip = 2 * npos > nval;
  const int64_t m = p.flip ? nval - npos : npos;
  p.m = m > 0 ? (uint32_t)m : 0u;
  p.q = 0u;
  if (m > 0 && nval > 0) {
    const uint64_t s = isqrt64((uint64_t)m * (uint64_t)(nval - m) / (uint64_t)nval);
    const uint64_t b = 12u * s * (uint64_t)(nval - 2 * m) / (5u * (uint64_t)nval);
    const uint64_t target = (uint64_t)m > b ? (uint64_t)m - b : 0u;
    p.q = (uint32_t)(256u * target / (uint64_t)nval);
  }
  return p;
}

// Round 0: the marks of isolate i in the 32 permutations of block B (bit b = permutation 32 B + b).
// fold_j (q_j ? x | R_j : x & R_j) from x = 0: leading zero bits of q keep x = 0, so the first
// Philox call is skipped when q's low nibble is zero.
__device__ __forceinline__ uint32_t bern_word(uint32_t i, uint32_t B, uint32_t t, uint32_t q,
                                              uint32_t k0, uint32_t k1) {
  uint32_t x = 0u, r[4];
  if (q == 0u) return 0u;
  if (q & 0xFu) {
    philox4x32_10(i, B, t, kDomBernLo, k0, k1, r);
#pragma unroll
    for (int j = 0; j < 4; ++j) x = ((q >> j) & 1u) ? (x | r[j]) : (x & r[j]);
  }
  philox4x32_10(i, B, t, kDomBernHi, k0, k1, r);
#pragma unroll
  for (int j = 0; j < 4; ++j) x = ((q >> (4 + j)) & 1u) ? (x | r[j]) : (x & r[j]);
  return x;
}

struct LabelArgs {
  const uint32_t* masks;     // vecrows [T][Wp]
  const int32_t* margins;    // [T][2] = (npos, nval)
  int N, Wp;
  int64_t P, perm_base;
  int trait_base;
  uint32_t k0, k1;
  int elt_log2;              // log2 of the permutations of a dword column one block keeps (5: all 32)
  int debug;                 // SCOARY_LABELS_DEBUG (timing experiments): 1 = no fix-up, 2 = no round 0
  // tiles
  int ntiles, TW;
  int64_t first_flat, nflat; // flat (trait, tile) range of this launch
  uint32_t* out;
};

// NB dword columns per block (NB > 1 only with all 32 permutations per dword); OUT 0: tiles, 1: rows
template <int NB, int OUT>   // NB = 1, 2
__global__ __launch_bounds__(1024) void k_labels(const LabelArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int tid = threadIdx.x, lane = tid & 63, tpb = blockDim.x, nwaves = tpb >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int N = a.N;
  const int elt = 1 << a.elt_log2;                  // permutation bits a row keeps per column
  const int subs = 32 >> a.elt_log2;                // blocks that share one dword column
  // ---- which (trait, dword column(s), bits) this block produces ----
  int t, bit0;
  uint32_t Bglob;                                    // Philox block of column 0
  int64_t lc0 = 0;                                   // tiles: dword column among this launch's permutations
  uint32_t* tile_base = nullptr;                     // tiles: row 0, column `col` of the block's tile
  if constexpr (OUT == 0) {
    // block id -> (flat tile, unit): the units of one tile get ids that are equal mod 8, i.e. run
    // on one XCD (observed placement: block b on XCD b % 8), so the 4 * NB-byte pieces they
    // write into the same tile rows meet in one L2 and leave it as whole lines
    const int units = a.TW * subs / NB;
    const int64_t id = blockIdx.x;
    const int64_t qd = id >> 3;
    const int unit = (int)(qd % units);
    const int64_t fl = (qd / units) * 8 + (id & 7);
    if (fl >= a.nflat) return;
    const int64_t f = a.first_flat + fl;
    t = (int)(f / a.ntiles);
    const int tile = (int)(f % a.ntiles);
    const int col = unit / subs * NB;
    bit0 = (unit % subs) * elt;
    lc0 = (int64_t)tile * a.TW + col;
    Bglob = (uint32_t)((a.perm_base >> 5) + lc0);
    const int64_t tile_dw = a.TW == kSegTW ? list_tile_dwords_seg(N, a.TW) : list_tile_dwords(N, a.TW);
    tile_base = a.out + f * tile_dw + col;
  } else {
    t = blockIdx.y;
    Bglob = (uint32_t)((a.perm_base >> 5) + blockIdx.x / subs);
    bit0 = (int)(blockIdx.x % subs) * elt;
  }
  // permutations of column w that exist, as a mask over the 32 bits of the Philox block
  uint32_t live[NB];
#pragma unroll
  for (int w = 0; w < NB; ++w) {
    int64_t lo, hi;                                  // bits [lo, hi) of the block exist
    if constexpr (OUT == 0) {
      lo = 0;
      hi = a.P - (lc0 + w) * 32;
    } else {
      const int64_t first = (int64_t)(Bglob + w) * 32;
      lo = a.perm_base - first;
      hi = a.perm_base + a.P - first;
    }
    lo = lo < 0 ? 0 : (lo > 32 ? 32 : lo);
    hi = hi < 0 ? 0 : (hi > 32 ? 32 : hi);
    const uint32_t below_hi = hi >= 32 ? 0xffffffffu : ((1u << hi) - 1u);
    const uint32_t below_lo = lo >= 32 ? 0xffffffffu : ((1u << lo) - 1u);
    live[w] = hi > lo ? (below_hi & ~below_lo) : 0u;
  }
  const uint32_t fieldmask = elt == 32 ? 0xffffffffu : ((1u << elt) - 1u);
  const uint32_t tglob = (uint32_t)(a.trait_base + t);
  const LabelPlan plan = label_plan(__builtin_amdgcn_readfirstlane(a.margins[2 * t]),
                                    __builtin_amdgcn_readfirstlane(a.margins[2 * t + 1]));
  const uint32_t* mrow = a.masks + (int64_t)t * a.Wp;

  // ---- LDS: rows (stride_bits per isolate), validity words, mark counts ----
  const int stride_bits = elt * NB;
  const int xs_dwords = (int)(((int64_t)N * stride_bits + 31) >> 5);
  uint32_t* xs = lds;
  uint32_t* vm = lds + ((xs_dwords + 3) & ~3);
  const int nmw = (N + 31) >> 5;
  int* kc = reinterpret_cast<int*>(vm + ((nmw + 3) & ~3));
  for (int k = tid; k < nmw; k += tpb) vm[k] = mrow[k];
  if (tid < 32 * NB) kc[tid] = 0;
  if (elt != 32) {                                   // sub-dword rows are OR-ed into place
    for (int k = tid; k < xs_dwords; k += tpb) xs[k] = 0u;
    __syncthreads();
  }

  // ---- round 0: lane = isolate ----
  uint32_t cp[NB][kCntPlanes];
#pragma unroll
  for (int w = 0; w < NB; ++w)
#pragma unroll
    for (int k = 0; k < kCntPlanes; ++k) cp[w][k] = 0u;
  const int rows_per_thread = (N + tpb - 1) / tpb;
  const int kpl = 32 - __builtin_clz((unsigned)rows_per_thread | 1u);   // planes a lane's count needs
  for (int row = tid; row < N; row += tpb) {
    const bool valid = (mrow[row >> 5] >> (row & 31)) & 1u;
    uint32_t x[NB];
#pragma unroll
    for (int w = 0; w < NB; ++w) {
      x[w] = 0u;
      if (valid && !(a.debug & 2))
        x[w] = ((bern_word((uint32_t)row, Bglob + (uint32_t)w, tglob, plan.q, a.k0, a.k1) & live[w]) >> bit0) &
               fieldmask;
      uint32_t carry = x[w];
#pragma unroll
      for (int k = 0; k < kCntPlanes; ++k)
        if (k < kpl) {
          const uint32_t nc = cp[w][k] & carry;
          cp[w][k] ^= carry;
          carry = nc;
        }
    }
    if (elt == 32) {
      if constexpr (NB == 2) {
        *reinterpret_cast<uint2*>(xs + (int64_t)row * 2) = make_uint2(x[0], x[1]);
      } else {
        xs[row] = x[0];
      }
    } else if (x[0]) {
      const uint32_t bp = (uint32_t)row << a.elt_log2;
      atomicOr(&xs[bp >> 5], x[0] << (bp & 31u));
    }
  }
  // ---- marks per permutation: butterfly over the wavefront, then one LDS atomic per wavefront ----
#pragma unroll
  for (int w = 0; w < NB; ++w) {
    if (a.debug & 8) break;
    uint32_t c[kSumPlanes];
#pragma unroll
    for (int k = 0; k < kSumPlanes; ++k) c[k] = k < kCntPlanes ? cp[w][k] : 0u;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      uint32_t carry = 0u;
#pragma unroll
      for (int k = 0; k < kSumPlanes; ++k)
        if (k < kpl + s + 1) {
          const uint32_t y = (uint32_t)__shfl_xor((int)c[k], 1 << s);
          const uint32_t sum = c[k] ^ y ^ carry;
          carry = (c[k] & y) | (carry & (c[k] | y));
          c[k] = sum;
        }
    }
    if (lane < 32) {
      int K = 0;
#pragma unroll
      for (int k = 0; k < kSumPlanes; ++k) K |= (int)((c[k] >> lane) & 1u) << k;
      // the atomics wait for the barrier below: kc was zeroed by other threads
      cp[w][0] = (uint32_t)K;
    }
  }
  __syncthreads();
  if (lane < 32) {
#pragma unroll
    for (int w = 0; w < NB; ++w)
      if (cp[w][0]) atomicAdd(&kc[w * 32 + lane], (int)cp[w][0]);
  }
  __syncthreads();

  // ---- fix-up: a group of L lanes per permutation ----
  // Spec S4 consumes the position draws c = 0, 1, 2, ... one after the other; but as long as at
  // least as many marks are missing as a batch has draws, EVERY distinct candidate the batch hits
  // is toggled, whatever the order -- so lane l of the group takes Philox call base + l (four
  // draws) and the group toggles with LDS atomics (the returned old bit says who was first at a
  // position).  A round uses k = min(L, |d| / 4) calls, so it can never overshoot; the last
  // < 4 marks are placed by the group's first lane, draw by draw.
  {
    const int nperm = stride_bits;                       // permutations of this block
    const int L = min(kWave, tpb / nperm);               // a power of two, >= 2
    const int pidx = tid / L, l = tid & (L - 1);
    const bool in_group = pidx < nperm;                  // tpb > 64 * nperm: the last wavefronts idle
    const int w = pidx >> 5, j = pidx & 31;
    const bool alive = in_group && ((live[NB == 1 ? 0 : (w & (NB - 1))] >> (bit0 + j)) & 1u);
    const uint32_t pi = (Bglob + (uint32_t)w) * 32u + (uint32_t)(bit0 + j);
    int d = alive && !(a.debug & 1) ? (int)plan.m - kc[in_group ? pidx : 0] : 0;   // > 0: add marks, < 0: remove
    const uint32_t reject_below = (uint32_t)(((uint64_t)1 << 32) % (uint64_t)N);
    const uint64_t gmask = (L == 64 ? ~(uint64_t)0 : (((uint64_t)1 << L) - 1)) << (lane & ~(L - 1));
    const int leader = lane & ~(L - 1);
    // one draw: position from a Philox word, toggled if it is a candidate; returns 1 if this lane
    // changed the bit
    // Four draws of one Philox call: positions, then the toggles as LDS atomics -- an OR on a
    // marked isolate / an AND on an unmarked one changes nothing, so only invalid isolates have
    // to be kept away (add mode), and the returned old word says whether THIS lane changed the
    // bit.  The four atomics are independent: issued back to back, one wait.  `upto`: stop after
    // that many changes (the sequential tail; 4 = no limit).
    auto draws4 = [&](const uint32_t (&rnd)[4], bool add) -> int {
      uint32_t idx[4], bit[4];
      bool ok[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint64_t prod = (uint64_t)rnd[u] * (uint64_t)(uint32_t)N;
        const uint32_t pos = (uint32_t)(prod >> 32);
        const uint32_t bp = pos * (uint32_t)stride_bits + (uint32_t)pidx;
        idx[u] = bp >> 5;
        bit[u] = 1u << (bp & 31u);
        ok[u] = (uint32_t)prod >= reject_below;          // Lemire rejection: no draw otherwise
        if (add) ok[u] = ok[u] && ((vm[pos >> 5] >> (pos & 31u)) & 1u);
      }
      uint32_t old[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        old[u] = add ? bit[u] : 0u;                      // "no change" for the draws that are none
        if (ok[u])
          old[u] = add ? __hip_atomic_fetch_or(&xs[idx[u]], bit[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)
                       : __hip_atomic_fetch_and(&xs[idx[u]], ~bit[u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      int won = 0;
#pragma unroll
      for (int u = 0; u < 4; ++u) won += add ? ((old[u] & bit[u]) ? 0 : 1) : ((old[u] & bit[u]) ? 1 : 0);
      return won;
    };
    // one draw at a time (the last < 4 marks): returns 1 if the bit changed
    auto draw1 = [&](uint32_t rnd, bool add) -> int {
      const uint64_t prod = (uint64_t)rnd * (uint64_t)(uint32_t)N;
      if ((uint32_t)prod < reject_below) return 0;
      const uint32_t pos = (uint32_t)(prod >> 32);
      const uint32_t bp = pos * (uint32_t)stride_bits + (uint32_t)pidx, b1 = 1u << (bp & 31u);
      if (add) {
        if (!((vm[pos >> 5] >> (pos & 31u)) & 1u)) return 0;
        return (__hip_atomic_fetch_or(&xs[bp >> 5], b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & b1) ? 0 : 1;
      }
      return (__hip_atomic_fetch_and(&xs[bp >> 5], ~b1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & b1) ? 1 : 0;
    };
    uint32_t base = 0;                                   // next Philox call of this permutation
    while (d != 0 && base < kFixMaxCalls) {              // group-uniform
      const bool add = d > 0;
      const int need = add ? d : -d;
      const int k = min(L, need >> 2);
      uint32_t r[4];
      if (k == 0) {                                      // the last marks: in draw order
        int left = need;
        if (l == 0) {
          philox4x32_10(base, pi, tglob, kDomFix, a.k0, a.k1, r);
#pragma unroll
          for (int u = 0; u < 4; ++u)
            if (left > 0) left -= draw1(r[u], add);
        }
        left = __shfl(left, leader);
        d = add ? left : -left;
        base += 1u;
      } else {
        int won = 0;
        if (l < k) {
          philox4x32_10(base + (uint32_t)l, pi, tglob, kDomFix, a.k0, a.k1, r);
          won = draws4(r, add);
        }
        // marks toggled by the group in this round
        int got = 0;
#pragma unroll
        for (int bitn = 0; bitn < 3; ++bitn)
          got += __popcll(__ballot((won >> bitn) & 1) & gmask) << bitn;
        d = add ? d - got : d + got;
        base += (uint32_t)k;
      }
    }
  }
  __syncthreads();

  // ---- output ----
  // the labels of row `row`, column w: marks, or their complement among the valid isolates
  auto row_field = [&](int row, int w) -> uint32_t {
    uint32_t f;
    if (elt == 32) {
      f = xs[(int64_t)row * NB + w];
    } else {
      const uint32_t bp = (uint32_t)row << a.elt_log2;
      f = (xs[bp >> 5] >> (bp & 31u)) & fieldmask;
    }
    if (plan.flip) {
      const bool valid = (vm[row >> 5] >> (row & 31)) & 1u;
      f = valid ? (~f & (live[w] >> bit0) & fieldmask) : 0u;
    }
    return f;
  };
  if (a.debug & 4) return;
  if constexpr (OUT == 0) {
    const bool seg = a.TW == kSegTW;                  // two-dword tiles: segmented above N = 20479
    auto row_off = [&](int64_t row) -> int64_t { return seg ? list_row_dword(N, row) : row * a.TW; };
    if (elt == 32) {
      for (int row = tid; row < N; row += tpb) {
        uint32_t* dst = tile_base + row_off(row);
        if constexpr (NB == 2) {
          *reinterpret_cast<uint2*>(dst) = make_uint2(row_field(row, 0), row_field(row, 1));
        } else {
          dst[0] = row_field(row, 0);
        }
      }
    } else {                                          // 16 / 8 permutations of the dword: narrow stores
      for (int row = tid; row < N; row += tpb) {
        uint8_t* dst = reinterpret_cast<uint8_t*>(tile_base + row_off(row)) + (bit0 >> 3);
        const uint32_t f = row_field(row, 0);
        if (elt == 16) *reinterpret_cast<uint16_t*>(dst) = (uint16_t)f;
        else *dst = (uint8_t)f;
      }
    }
    // the all-zero row(s) list padding points at: one per segment
    const int nseg = seg ? list_segments(N) : 1;
    if (tid < nseg) {
      const int64_t z = nseg > 1 ? (int64_t)tid * kSegStride + list_seg_rows(N, tid) * kSegTW : row_off(N);
      if (elt == 32) {
#pragma unroll
        for (int w = 0; w < NB; ++w) tile_base[z + w] = 0u;
      } else {
        uint8_t* dst = reinterpret_cast<uint8_t*>(tile_base + z) + (bit0 >> 3);
        if (elt == 16) *reinterpret_cast<uint16_t*>(dst) = 0;
        else *dst = 0;
      }
    }
  } else {
    static_assert(OUT == 0 || NB == 1, "bit rows: one dword column per block");
    // permutation-major rows: 64 isolates per wavefront step, one ballot per permutation
    const int nchunks = (N + 63) >> 6;
    const int64_t pl = (int64_t)Bglob * 32 + bit0 + lane - a.perm_base;   // lane < elt: its permutation
    const bool mine = lane < elt && ((live[0] >> (bit0 + (lane & 31))) & 1u);
    uint32_t* prow = a.out + ((int64_t)t * a.P + (mine ? pl : 0)) * a.Wp;
    for (int c = wave; c < nchunks; c += nwaves) {
      const int row = c * 64 + lane;
      const uint32_t f = row < N ? row_field(row, 0) : 0u;
      uint32_t lo = 0u, hi = 0u;
      for (int j = 0; j < elt; ++j) {
        const uint64_t m64 = __ballot((f >> j) & 1u);
        if (lane == j) {
          lo = (uint32_t)m64;
          hi = (uint32_t)(m64 >> 32);
        }
      }
      if (mine) *reinterpret_cast<uint2*>(prow + 2 * c) = make_uint2(lo, hi);
    }
    if (mine)
      for (int k = 2 * nchunks + wave; k < a.Wp; k += nwaves) prow[k] = 0u;
  }
}

// ---- launch geometry --------------------------------------------------------------------------
int64_t labels_lds_bytes(int64_t N, int elt, int NB) {
  const int64_t xs = ((N * elt * NB + 31) / 32 + 3) / 4 * 4, vm = ((N + 31) / 32 + 3) / 4 * 4;
  return (xs + vm + 32 * NB) * 4;
}
// permutations of a dword column a block keeps: 32 while its rows fit LDS, else 16, 8 (tiles),
// ... 1 (bit rows); 0: N too large
int labels_elt(int64_t N, int min_elt) {
  for (int elt = 32; elt >= min_elt; elt >>= 1)
    if (labels_lds_bytes(N, elt, 1) <= kLabelsMaxLds) return elt;
  return 0;
}
// tuning experiments (tools/gen_time.py): SCOARY_LABELS_TPB / SCOARY_LABELS_NB override the launch geometry
int labels_env(const char* name) {
  const char* e = std::getenv(name);
  return e ? std::atoi(e) : 0;
}
int labels_debug() {
  static const int v = [] { const char* e = std::getenv("SCOARY_LABELS_DEBUG"); return e ? std::atoi(e) : 0; }();
  return v;
}
int ilog2(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return l;
}
// threads per block: 256, more when the launch has too few wavefronts for the chip and the rows
// are long (the row loop is what the extra wavefronts share)
int labels_threads(int64_t blocks, int64_t N, int num_cu) {
  int tpb = 256;
  while (tpb < 1024 && blocks * (tpb / 64) < (int64_t)num_cu * 8 && N / tpb >= 4) tpb *= 2;
  // a lane counts its round-0 marks in kCntPlanes bit-sliced planes: at most 1023 rows per thread
  // (N > 261 888 needs 512 threads, N > 523 776 needs 1024; scoary_perm_max_isolates() = 654 848)
  while (tpb < 1024 && (N + tpb - 1) / tpb > kMaxRowsPerThread) tpb *= 2;
  return tpb;
}
template <int NB, int OUT>
int launch_labels(scoary_handle h, hipStream_t s, const LabelArgs& a, dim3 grid, int tpb, int elt) {
  const size_t lds = (size_t)labels_lds_bytes(a.N, elt, NB);
  // the per-lane mark counters would wrap (labels_threads never picks such a geometry; an
  // SCOARY_LABELS_TPB override can)
  if (tpb < 64 || tpb > 1024 || (tpb & (tpb - 1)) || (a.N + tpb - 1) / tpb > kMaxRowsPerThread)
    return fail(h, SCOARY_ERR_SIZE, "k_labels: more than 1023 isolates per thread (or a block size that is no "
                                    "power of two in 64..1024)");
  const void* fn = reinterpret_cast<const void*>(&k_labels<NB, OUT>);
  const int bit = 1 << (NB + 8 * OUT);             // 2, 4, 16 | 512
  if (lds > 64 * 1024 && !(h->labels_lds_optin & bit)) {
    HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kLabelsMaxLds));
    h->labels_lds_optin |= bit;
  }
  hipLaunchKernelGGL((k_labels<NB, OUT>), grid, dim3((unsigned)tpb), lds, s, a);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

}  // namespace

extern "C" {

int64_t scoary_perm_max_isolates(void) {
  int64_t lo = 1, hi = (int64_t)1 << 24;            // largest N with one permutation per row in LDS
  while (lo < hi) {
    const int64_t mid = (lo + hi + 1) / 2;
    if (labels_elt(mid, 1)) lo = mid; else hi = mid - 1;
  }
  return lo;
}

int scoary_perm_generate(scoary_handle h, const uint32_t* d_masks, const int32_t* d_margins,
                         int64_t T, int64_t N, int64_t P, int64_t perm_base, int64_t trait_base,
                         uint64_t seed, uint32_t* d_perms, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_masks || !d_margins || !d_perms || T < 1 || N < 1 || P < 1 || perm_base < 0 || trait_base < 0)
    return fail(h, SCOARY_ERR_ARG, "scoary_perm_generate: bad argument");
  if (T > 65535 || trait_base + T > 0x7fffffffLL || perm_base + P > 0xffffffffLL)
    return fail(h, SCOARY_ERR_SIZE, "scoary_perm_generate: T > 65535 or permutation index >= 2^32");
  const int elt = labels_elt(N, 1);
  if (!elt) return fail(h, SCOARY_ERR_SIZE, "scoary_perm_generate: more isolates than scoary_perm_max_isolates()");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  LabelArgs a{};
  a.masks = d_masks, a.margins = d_margins, a.N = (int)N, a.Wp = (int)scoary_row_words(N);
  a.P = P, a.perm_base = perm_base, a.trait_base = (int)trait_base;
  a.k0 = (uint32_t)seed, a.k1 = (uint32_t)(seed >> 32), a.elt_log2 = ilog2(elt), a.out = d_perms;
  a.debug = labels_debug();
  const int64_t nblk32 = ((perm_base + P - 1) >> 5) - (perm_base >> 5) + 1;   // Philox blocks touched
  const int64_t gx = nblk32 * (32 / elt);
  if (gx > 0x7fffffffLL) return fail(h, SCOARY_ERR_SIZE, "scoary_perm_generate: grid too large");
  KernelTimer kt(h, s, "k_perm_generate");
  return launch_labels<1, 1>(h, s, a, dim3((unsigned)gx, (unsigned)T),
                             labels_threads(gx * T, N, h->num_cu), elt);
}

int scoary_perm_generate_tiles_range(scoary_handle h, const uint32_t* d_masks, const int32_t* d_margins,
                                     int64_t T, int64_t N, int64_t P, int64_t perm_base,
                                     int64_t trait_base, uint64_t seed, int64_t first_tile,
                                     int64_t n_tiles, uint32_t* d_tiles, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_masks || !d_margins || !d_tiles || T < 1 || N < 1 || P < 1 || perm_base < 0 || trait_base < 0 ||
      first_tile < 0 || n_tiles < 0)
    return fail(h, SCOARY_ERR_ARG, "scoary_perm_generate_tiles: bad argument");
  if (perm_base & 31)
    return fail(h, SCOARY_ERR_ARG, "scoary_perm_generate_tiles: perm_base must be a multiple of 32");
  if (T > 65535 || perm_base + P > 0xffffffffLL)
    return fail(h, SCOARY_ERR_SIZE, "scoary_perm_generate_tiles: T > 65535 or permutation index >= 2^32");
  const int TW = list_tw(N);
  if (!TW) return fail(h, SCOARY_ERR_SIZE, "scoary_perm_generate_tiles: N too large for LDS tiles");
  const int64_t tile_perms = TW * 32;
  const int64_t ntiles = (P + tile_perms - 1) / tile_perms;
  if (first_tile + n_tiles > T * ntiles)
    return fail(h, SCOARY_ERR_ARG, "scoary_perm_generate_tiles: tile range past the last (trait, tile)");
  if (n_tiles == 0) return SCOARY_OK;
  const int elt = labels_elt(N, 8);
  if (!elt) return fail(h, SCOARY_ERR_SIZE, "scoary_perm_generate_tiles: N too large for the label generator");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  LabelArgs a{};
  a.masks = d_masks, a.margins = d_margins, a.N = (int)N, a.Wp = (int)scoary_row_words(N);
  a.P = P, a.perm_base = perm_base, a.trait_base = (int)trait_base;
  a.k0 = (uint32_t)seed, a.k1 = (uint32_t)(seed >> 32), a.elt_log2 = ilog2(elt);
  a.ntiles = (int)ntiles, a.TW = TW, a.first_flat = first_tile, a.nflat = n_tiles, a.out = d_tiles;
  a.debug = labels_debug();
  // dword columns per block: as many as keep >= 2 blocks per CU in the launch and >= 2 blocks
  // of LDS per CU (wider pieces per tile row, fewer count reductions)
  // (four columns per block were measured too: the fix-up then has two lanes per permutation and
  // the kernel is slower at the headline shape, 0.086 against 0.077 ms)
  int NB = 1;
  if (elt == 32)
    for (int nb = 2; nb > 1; nb >>= 1)
      if (nb <= TW && n_tiles * (TW / nb) >= 2 * (int64_t)h->num_cu &&
          labels_lds_bytes(N, 32, nb) <= kLabelsMaxLds / 2) {
        NB = nb;
        break;
      }
  if (elt == 32 && labels_env("SCOARY_LABELS_NB") && labels_env("SCOARY_LABELS_NB") <= (TW < 2 ? TW : 2)) NB = labels_env("SCOARY_LABELS_NB");
  const int64_t units = (int64_t)TW * (32 / elt) / NB;
  const int64_t gx = (n_tiles + 7) / 8 * 8 * units;
  if (gx > 0x7fffffffLL) return fail(h, SCOARY_ERR_SIZE, "scoary_perm_generate_tiles: grid too large");
  int tpb = labels_threads(n_tiles * units, N, h->num_cu);
  if (labels_env("SCOARY_LABELS_TPB")) tpb = labels_env("SCOARY_LABELS_TPB");
  KernelTimer kt(h, s, "k_perm_generate_tiles");
  if (NB == 2) return launch_labels<2, 0>(h, s, a, dim3((unsigned)gx), tpb, elt);
  return launch_labels<1, 0>(h, s, a, dim3((unsigned)gx), tpb, elt);
}

int scoary_perm_generate_tiles(scoary_handle h, const uint32_t* d_masks, const int32_t* d_margins,
                               int64_t T, int64_t N, int64_t P, int64_t perm_base,
                               int64_t trait_base, uint64_t seed, uint32_t* d_tiles,
                               scoary_stream_t stream) {
  const int TW = list_tw(N);
  const int64_t ntiles = TW && P > 0 ? (P + TW * 32 - 1) / (TW * 32) : 0;
  return scoary_perm_generate_tiles_range(h, d_masks, d_margins, T, N, P, perm_base, trait_base, seed,
                                          0, T * ntiles, d_tiles, stream);
}

}  // extern "C"
