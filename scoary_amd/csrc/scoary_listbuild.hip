// scoary_listbuild.hip -- the minority index lists of the list-driven permutation
// kernel (scoary_lists.hip), built on the device from the tiled gene matrix that is
// already in HBM: per-gene popcount -> flip decision -> stable length sort (LSD radix,
// 8-bit digits) -> padded lengths and group bases (prefix sum) -> scatter of the listed
// positions in the bank-rotation order.  The layout is spec S6;
// the host builder scoary_lists_build (scoary_io.cpp) implements the same spec
// independently and is the checker in the tests.
#include "scoary_common.hpp"

namespace {

constexpr int kSortItems = 2048;   // items per sort block (one wavefront)

__host__ __device__ inline int64_t lb_align8(int64_t x) { return (x + 7) / 8 * 8; }

// Scratch layout (bytes), shared by scoary_lists_plan and scoary_lists_fill.
struct ListScratch {
  int64_t len, ord_a, ord_b, hist, base, padded, seglen, total;
  int64_t nblk, nwg, nseg;
};
inline ListScratch list_scratch(int64_t G, int64_t N) {
  ListScratch s{};
  const int TW = list_tw(N);
  const int64_t gpw = TW ? kWave / list_lpg(TW) : 64;
  s.nblk = (G + kSortItems - 1) / kSortItems;
  s.nwg = (G + gpw - 1) / gpw;
  s.nseg = list_segments(N) > 1 ? list_segments(N) : 1;      // sub-lists per gene (N > 20479)
  int64_t off = 0;
  s.len = off;    off += lb_align8(4 * G);
  s.ord_a = off;  off += lb_align8(4 * G);
  s.ord_b = off;  off += lb_align8(4 * G);
  s.hist = off;   off += lb_align8(4 * 256 * s.nblk);
  s.base = off;   off += 8 * (s.nwg * s.nseg + 1);           // [nwg][nseg] + the total
  s.padded = off; off += lb_align8(4 * s.nwg * s.nseg);       // [nwg][nseg]
  s.seglen = off; off += s.nseg > 1 ? lb_align8(4 * G * s.nseg) : 0;   // [nseg][G]
  s.total = off;
  return s;
}

// lane = gene: minority length and flip flag from the tiled rows (pad bits are zero)
__global__ __launch_bounds__(256) void k_lists_len(const uint4* __restrict__ tiled, int64_t Gp,
                                                   int G, int N, int Qn, int32_t* __restrict__ len,
                                                   uint8_t* __restrict__ flipped) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  int n1 = 0;
  for (int q = 0; q < Qn; ++q) {
    const uint4 v = tiled[(int64_t)q * Gp + g];
    n1 += __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  }
  const bool fl = 2 * n1 > N;
  len[g] = fl ? N - n1 : n1;
  flipped[g] = fl ? 1 : 0;
}

// ---- stable LSD radix sort of the genes by DESCENDING list length ----------------
// key = kmax - len (ascending key = descending length), 8-bit digits; one wavefront
// per block of kSortItems consecutive items keeps the in-block order without any
// cross-wavefront hand-over.
__device__ __forceinline__ int sort_digit(const int32_t* in_order, const int32_t* len, int i,
                                          int kmax, int shift, int& g) {
  g = in_order ? in_order[i] : i;
  return ((kmax - len[g]) >> shift) & 255;
}
__global__ __launch_bounds__(64) void k_sort_hist(const int32_t* __restrict__ in_order,
                                                  const int32_t* __restrict__ len, int G, int kmax,
                                                  int shift, int nblk, int32_t* __restrict__ hist) {
  __shared__ int32_t h[256];
  const int lane = threadIdx.x, b = blockIdx.x;
  for (int d = lane; d < 256; d += 64) h[d] = 0;
  __syncthreads();
  const int lo = b * kSortItems, hi = min(G, lo + kSortItems);
  for (int i = lo + lane; i < hi; i += 64) {
    int g;
    atomicAdd(&h[sort_digit(in_order, len, i, kmax, shift, g)], 1);
  }
  __syncthreads();
  for (int d = lane; d < 256; d += 64) hist[(int64_t)d * nblk + b] = h[d];
}
// exclusive prefix sum of n int32 values in place, one block
__global__ __launch_bounds__(1024) void k_scan_excl(int32_t* __restrict__ v, int64_t n) {
  __shared__ int32_t wsum[16];
  __shared__ int32_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t base = 0; base < n; base += 1024) {
    const int64_t i = base + tid;
    const int32_t x = i < n ? v[i] : 0;
    int32_t s = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int32_t y = __shfl_up(s, off);
      if (lane >= off) s += y;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int32_t wo = 0;
    for (int w = 0; w < wave; ++w) wo += wsum[w];
    const int32_t carry = carry_s;
    if (i < n) v[i] = carry + wo + s - x;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wo + s;
    __syncthreads();
  }
}
__global__ __launch_bounds__(64) void k_sort_scatter(const int32_t* __restrict__ in_order,
                                                     const int32_t* __restrict__ len, int G,
                                                     int kmax, int shift, int nblk,
                                                     const int32_t* __restrict__ hist,
                                                     int32_t* __restrict__ out_order) {
  __shared__ int32_t cnt[256];
  const int lane = threadIdx.x, b = blockIdx.x;
  for (int d = lane; d < 256; d += 64) cnt[d] = hist[(int64_t)d * nblk + b];
  __syncthreads();
  const int lo = b * kSortItems, hi = min(G, lo + kSortItems);
  for (int i0 = lo; i0 < hi; i0 += 64) {
    const int i = i0 + lane;
    const bool valid = i < hi;
    int g = 0;
    const int d = valid ? sort_digit(in_order, len, i, kmax, shift, g) : 0;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const bool set = (d >> bit) & 1;
      const uint64_t bb = __ballot(valid && set);
      peers &= set ? bb : ~bb;
    }
    const int rank = __popcll(peers & (((uint64_t)1 << lane) - 1));
    if (valid) out_order[cnt[d] + rank] = g;
    __syncthreads();
    if (valid && rank == 0) cnt[d] += __popcll(peers);
    __syncthreads();
  }
}

// Padded list length of every wavefront group (the longest list of the group, rounded
// up to kListPad) and the first entry of the group: exclusive prefix sum, one block.
__global__ __launch_bounds__(1024) void k_lists_plan(const int32_t* __restrict__ len,
                                                     const int32_t* __restrict__ order, int G,
                                                     int gpw, int64_t nwg,
                                                     int32_t* __restrict__ padded,
                                                     int64_t* __restrict__ base) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t q0 = 0; q0 < nwg; q0 += 1024) {
    const int64_t q = q0 + tid;
    int32_t L = 0;
    if (q < nwg) {
      L = (len[order[q * gpw]] + kListPad - 1) / kListPad * kListPad;
      padded[q] = L;
    }
    const int64_t x = (int64_t)L * gpw;
    int64_t s = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int64_t y = __shfl_up(s, off);
      if (lane >= off) s += y;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int64_t wo = 0;
    for (int w = 0; w < wave; ++w) wo += wsum[w];
    const int64_t carry = carry_s;
    if (q < nwg) base[q] = carry + wo + s - x;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wo + s;
    __syncthreads();
  }
  if (tid == 0) base[nwg] = carry_s;
}
__global__ __launch_bounds__(256) void k_lists_slots(const int32_t* __restrict__ padded,
                                                     const int64_t* __restrict__ base, int G,
                                                     int gpw, int32_t* __restrict__ start,
                                                     int32_t* __restrict__ ngroups) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= G) return;
  const int q = k / gpw;
  start[k] = (int32_t)(base[q] / kListStartUnit);      // gpw * padded is a multiple of 16 * 16
  ngroups[k] = padded[q] / kListPad;                   // half-steps of 16 entries
}

// Spec S6, entry order.  Class of a position = position mod C; within a class positions are
// ranked ascending (rho = 0, 1, ...).  The position (class c, rank rho) belongs on grid slot
//   e = rho * C + ((c - k) mod C)
// of list slot k -- entry e then comes from class (k + e) mod C, and the genes of an LDS lane
// group, whose slots k are consecutive, sit on distinct bank slots at every step.  The list
// has total = sum of the class counts entries and no gaps: positions whose grid slot is
// >= total ("overflow": the classes with more positions than average) fill, in grid order,
// the holes below total (grid slots of classes that have run dry), in hole order.  Everything
// is closed form -- filled(x) = sum over c' of min(cnt[c'], ceil((x - d(c')) / C)) grid slots
// below x are occupied -- so every position finds its entry independently: an overflow
// position of rank ov among the overflow goes to the smallest x with x + 1 - filled(x + 1)
// = ov + 1 (binary search).  All but the hole entries are aligned.
// C lanes per slot (lane = class), 64 / C slots per wavefront.
template <int C>
__global__ __launch_bounds__(256) void k_lists_fill(const uint4* __restrict__ tiled, int64_t Gp,
                                                    int G, int N, int Qn,
                                                    const int32_t* __restrict__ len,
                                                    const int32_t* __restrict__ order,
                                                    const uint8_t* __restrict__ flipped,
                                                    const int64_t* __restrict__ base,
                                                    const int32_t* __restrict__ padded, int gpw,
                                                    int piece, uint32_t row_stride, int64_t nslots,
                                                    uint32_t* __restrict__ idx) {
  __shared__ int32_t s_cnt[256];
  constexpr int SPW = 64 / C;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane % C;
  const int64_t k = ((int64_t)blockIdx.x * 4 + wave) * SPW + lane / C;
  const bool exists = k < nslots, live = k < G;
  const int64_t q = exists ? k / gpw : 0;
  const int j = (int)(k - q * gpw);
  const int L = exists ? padded[q] : 0;
  const int64_t b0 = exists ? base[q] : 0;
  const int g = live ? order[k] : 0;
  const int total = live ? len[g] : 0;
  const uint32_t inv = (live && flipped[g]) ? 0xffffffffu : 0u;
  static_assert(C <= 32, "a class is a set of bits of every 32-bit word");
  uint32_t cmask = 0u;                       // bits of a 32-bit word that belong to class c
  for (int b = c; b < 32; b += C) cmask |= 1u << b;
  auto class_bits = [&](uint32_t word, int w) -> uint32_t {
    const int first = 32 * w;
    uint32_t bits = word ^ inv;
    if (first + 32 > N) bits &= first < N ? ((1u << (N - first)) - 1u) : 0u;
    return bits & cmask;
  };
  auto at = [&](int n) -> int64_t {          // interleaved position of entry n of this slot
    return b0 + ((int64_t)(n / piece) * gpw + j) * piece + n % piece;
  };
  int my_cnt = 0;
  if (live)
    for (int qd = 0; qd < Qn; ++qd) {
      const uint4 v = tiled[(int64_t)qd * Gp + g];
      my_cnt += __popc(class_bits(v.x, 4 * qd)) + __popc(class_bits(v.y, 4 * qd + 1)) +
                __popc(class_bits(v.z, 4 * qd + 2)) + __popc(class_bits(v.w, 4 * qd + 3));
    }
  s_cnt[tid] = my_cnt;
  __syncthreads();
  const int32_t* cnt = s_cnt + (tid - c);    // the C class counts of this slot
  const int dk = (int)((c - k) & (C - 1));
  // grid slots below x that hold a position: class x' owns the slots rho * C + d(x'), rho < cnt[x']
  auto filled = [&](int x) -> int {
    int f = 0;
    for (int xc = 0; xc < C; ++xc) {
      const int dx = (int)((xc - k) & (C - 1));
      f += min(cnt[xc], max(0, (x - dx + C - 1) / C));
    }
    return f;
  };
  const int filled_total = filled(total);
  if (live) {
    int rho = 0;
    for (int qd = 0; qd < Qn; ++qd) {
      const uint4 v = tiled[(int64_t)qd * Gp + g];
      const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        uint32_t bits = class_bits(words[w4], 4 * qd + w4);
        while (bits) {
          const int b = __builtin_ctz(bits);
          bits &= bits - 1;
          int pos = rho * C + dk;                    // the position's aligned grid slot
          if (pos >= total) {                          // overflow: goes to the ov-th hole
            const int ov = filled(pos) - filled_total;
            int lo = 0, hi = total - 1;                // smallest x with holes(x + 1) >= ov + 1
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (mid + 1 - filled(mid + 1) >= ov + 1) hi = mid; else lo = mid + 1;
            }
            pos = lo;
          }
          idx[at(pos)] = (uint32_t)(32 * (4 * qd + w4) + b) * row_stride;
          ++rho;
        }
      }
    }
  }
  const uint32_t zero_row = (uint32_t)N * row_stride;
  for (int n = total + c; n < L; n += C) idx[at(n)] = zero_row;
}

// ---- segmented lists (N > 20479, scoary_common.hpp): one sub-list per gene and segment ----
// Order and flip are the whole row's (k_lists_len + sort); the sub-lists of a wave group are
// padded to the longest of the group in that segment; entries are 16-bit row indices inside the
// segment (row - segment start), padding points at the segment's own zero row.
// Segments start on multiples of 128 isolates, so a position's residue class mod 32 is the
// same inside the segment as in the row.

// lane = gene: minority count inside every segment
__global__ __launch_bounds__(256) void k_lists_seglen(const uint4* __restrict__ tiled, int64_t Gp,
                                                      int G, int N, int nseg,
                                                      const uint8_t* __restrict__ flipped,
                                                      int32_t* __restrict__ seglen) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= G) return;
  const uint32_t inv = flipped[g] ? 0xffffffffu : 0u;
  for (int sgm = 0; sgm < nseg; ++sgm) {
    const int row0 = sgm * kSegRows, rows = (int)list_seg_rows(N, sgm);
    int n = 0;
    for (int qd = row0 / 128; qd < (row0 + rows + 127) / 128; ++qd) {
      const uint4 v = tiled[(int64_t)qd * Gp + g];
      const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int w4 = 0; w4 < 4; ++w4) {
        const int first = 32 * (4 * qd + w4);
        uint32_t bits = words[w4] ^ inv;
        if (first + 32 > N) bits &= first < N ? ((1u << (N - first)) - 1u) : 0u;
        n += __popc(bits);
      }
    }
    seglen[(int64_t)sgm * G + g] = n;
  }
}
// thread = wave group: padded sub-list length per segment = the group's longest, rounded up
__global__ __launch_bounds__(256) void k_lists_segpad(const int32_t* __restrict__ seglen,
                                                      const int32_t* __restrict__ order, int G,
                                                      int gpw, int64_t nwg, int nseg,
                                                      int32_t* __restrict__ padded) {
  const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (q >= nwg) return;
  for (int sgm = 0; sgm < nseg; ++sgm) {
    int m = 0;
    for (int j = 0; j < gpw; ++j) {
      const int64_t k = q * gpw + j;
      if (k < G) m = max(m, seglen[(int64_t)sgm * G + order[k]]);
    }
    padded[q * nseg + sgm] = (m + kListPad - 1) / kListPad * kListPad;
  }
}
// base[i] = entries before item i (item = (wave group, segment), padded[i] * gpw entries each),
// base[n] = all entries: exclusive prefix sum, one block
__global__ __launch_bounds__(1024) void k_lists_base(const int32_t* __restrict__ padded, int64_t n,
                                                     int gpw, int64_t* __restrict__ base) {
  __shared__ int64_t wsum[16];
  __shared__ int64_t carry_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry_s = 0;
  __syncthreads();
  for (int64_t i0 = 0; i0 < n; i0 += 1024) {
    const int64_t i = i0 + tid;
    const int64_t x = i < n ? (int64_t)padded[i] * gpw : 0;
    int64_t s = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int64_t y = __shfl_up(s, off);
      if (lane >= off) s += y;
    }
    if (lane == 63) wsum[wave] = s;
    __syncthreads();
    int64_t wo = 0;
    for (int w = 0; w < wave; ++w) wo += wsum[w];
    const int64_t carry = carry_s;
    if (i < n) base[i] = carry + wo + s - x;
    __syncthreads();
    if (tid == 1023) carry_s = carry + wo + s;
    __syncthreads();
  }
  if (tid == 0) base[n] = carry_s;
}
__global__ __launch_bounds__(256) void k_lists_segslots(const int32_t* __restrict__ padded,
                                                        const int64_t* __restrict__ base, int G,
                                                        int gpw, int nseg, int32_t* __restrict__ start,
                                                        int32_t* __restrict__ ngroups) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= G) return;
  const int64_t q = k / gpw;
  for (int sgm = 0; sgm < nseg; ++sgm) {
    start[(int64_t)sgm * G + k] = (int32_t)(base[q * nseg + sgm] / (2 * kListStartUnit));   // 16-bit entries: 128 bytes = 64
    ngroups[(int64_t)sgm * G + k] = padded[q * nseg + sgm] / kListPad;
  }
}
// Segmented sub-lists, round 4: 16-BIT entries (the row index inside the segment, <= kSegRows = the
// segment's zero row; the kernel shifts it to an LDS address), eight per 16-byte index vector.
// Entry order: spec S6's alignment property with a cheaper hole assignment.  Class c = position
// mod 32 -- one bit of every 32-bit word, so a lane tests exactly one bit per word, no per-lane bit
// loops -- rank rho inside the class, grid slot rho * 32 + d, d = (c - k) mod 32.  With
// R = total / 32 complete grid rows:
//   * (c, rho), rho < R: ALIGNED, entry rho * 32 + d -- entry e then comes from class (k + e) mod 32 and
//     the 32 genes of an LDS service group read 32 distinct bank slots;
//   * (c, rho), rho >= R ("overflow" of the classes above the average, ~3 % of a random list): overflow
//     rank r = sum_{c' < c} max(0, cnt[c'] - R) + rho - R; the holes below row R are ranked class by
//     class (class c' has max(0, R - cnt[c']) of them, rows cnt[c'] .. R - 1): overflow r goes to hole r,
//     and the overflow beyond the holes to the partial last row R * 32 + (r - holes).
// Two 32-term prefix sums per (slot, segment) and a 5-step search per overflow entry replace round 3's
// binary search over a 32-term sum per overflow entry (k_lists_fill's closed form, 13.7 ms of a 16.7 ms
// set-up at 20 000 x 50 000).  (A plain compaction of the grid -- no holes to assign -- was tried first:
// its last ~25 % of rows are misaligned, LDS bank conflicts of k_permute_seglists 16 % -> 29 % of the
// LDS cycles, profiles/r04_wide50000_*.)  32 lanes per slot (lane = residue class), 2 slots per
// wavefront, 8 per block.
__global__ __launch_bounds__(256) void k_lists_fill_seg(const uint4* __restrict__ tiled, int64_t Gp,
                                                        int G, int N, int nseg,
                                                        const int32_t* __restrict__ seglen,
                                                        const int32_t* __restrict__ order,
                                                        const uint8_t* __restrict__ flipped,
                                                        const int64_t* __restrict__ base,
                                                        const int32_t* __restrict__ padded,
                                                        int64_t nslots, uint16_t* __restrict__ idx) {
  __shared__ int32_t s_cnt[256], s_hlp[256];
  constexpr int C = 32, gpw = 64, piece = kSegPiece;
  static_assert(64 / kSegTW == C, "residue classes of the two-dword tiles");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane % C;
  const int64_t k = ((int64_t)blockIdx.x * 4 + wave) * (64 / C) + lane / C;
  const bool exists = k < nslots, live = k < G;
  const int64_t q = exists ? k / gpw : 0;
  const int j = (int)(k - q * gpw);
  const int g = live ? order[k] : 0;
  const uint32_t inv = (live && flipped[g]) ? 0xffffffffu : 0u;
  // bit c of word w (position 32 w + c) of the minority pattern, 0 beyond N
  auto class_bit = [&](uint32_t word, int w) -> uint32_t {
    return (32 * w + c < N) ? (((word ^ inv) >> c) & 1u) : 0u;
  };
  const int32_t* cnt = s_cnt + (tid - c);      // the 32 class counts of this slot
  const int dk = (int)((c - k) & (C - 1));
  for (int sgm = 0; sgm < nseg; ++sgm) {
    const int row0 = sgm * kSegRows, rows = (int)list_seg_rows(N, sgm);
    const int qd0 = row0 / 128, qd1 = (row0 + rows + 127) / 128;
    const int L = exists ? padded[q * nseg + sgm] : 0;
    const int64_t b0 = exists ? base[q * nseg + sgm] : 0;          // in 16-bit entries
    const int total = live ? seglen[(int64_t)sgm * G + g] : 0;
    auto at = [&](int n) -> int64_t {          // interleaved position of entry n of this slot
      return b0 + ((int64_t)(n / piece) * gpw + j) * piece + n % piece;
    };
    int my_cnt = 0;
    if (live)
      for (int qd = qd0; qd < qd1; ++qd) {
        const uint4 v = tiled[(int64_t)qd * Gp + g];
        my_cnt += (int)(class_bit(v.x, 4 * qd) + class_bit(v.y, 4 * qd + 1) + class_bit(v.z, 4 * qd + 2) +
                        class_bit(v.w, 4 * qd + 3));
      }
    __syncthreads();                           // the previous segment's counts have been used
    s_cnt[tid] = my_cnt;
    __syncthreads();
    const int R = total / C;                   // complete grid rows (total = the sum of the 32 counts)
    const int ov = max(0, my_cnt - R), hl = max(0, R - my_cnt);
    int ovp = ov, hlp = hl;                    // inclusive prefix sums over the slot's classes 0 .. c
#pragma unroll
    for (int off = 1; off < C; off <<= 1) {
      const int o = __shfl_up(ovp, off, C), hh = __shfl_up(hlp, off, C);
      if (c >= off) {
        ovp += o;
        hlp += hh;
      }
    }
    const int holes = __shfl(hlp, C - 1, C);   // all holes of the slot
    ovp -= ov;                                 // exclusive: overflow rank of this class's first overflow entry
    s_hlp[tid] = hlp - hl;                     // exclusive: rank of this class's first hole
    __syncthreads();
    const int32_t* hbase = s_hlp + (tid - c);
    if (live) {
      int rho = 0;
      for (int qd = qd0; qd < qd1; ++qd) {
        const uint4 v = tiled[(int64_t)qd * Gp + g];
        const uint32_t words[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int w4 = 0; w4 < 4; ++w4) {
          if (!class_bit(words[w4], 4 * qd + w4)) continue;
          int pos = rho * C + dk;              // the aligned grid slot
          if (rho >= R) {                      // overflow: to the r-th hole, or into the partial last row
            const int r = ovp + (rho - R);
            if (r < holes) {
              int xc = 0;                      // the last class whose first hole has rank <= r
#pragma unroll
              for (int step = C / 2; step > 0; step >>= 1)
                if (hbase[xc + step] <= r) xc += step;
              pos = (cnt[xc] + (r - hbase[xc])) * C + (int)((xc - k) & (C - 1));
            } else {
              pos = R * C + (r - holes);
            }
          }
          idx[at(pos)] = (uint16_t)(32 * (4 * qd + w4) + c - row0);
          ++rho;
        }
      }
    }
    for (int n = total + c; n < L; n += C) idx[at(n)] = (uint16_t)rows;     // the segment's zero row
  }
}

__global__ __launch_bounds__(256) void k_lists_slack(uint32_t* __restrict__ idx, int64_t entries) {
  if (threadIdx.x < kListSlack) idx[entries + threadIdx.x] = 0u;
}

}  // namespace

extern "C" {

int64_t scoary_lists_scratch_bytes(int64_t G, int64_t N) {
  if (G < 1 || N < 1 || !list_tw(N)) return 0;
  return list_scratch(G, N).total;
}
int64_t scoary_lists_slack_entries(void) { return kListSlack; }

int scoary_lists_plan(scoary_handle h, const uint32_t* d_tiled, int64_t G, int64_t N,
                      void* d_scratch, int32_t* d_start, int32_t* d_ngroups, int32_t* d_order,
                      uint8_t* d_flipped, int64_t* entries_out, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_scratch || !d_start || !d_ngroups || !d_order || !d_flipped || !entries_out ||
      G < 1 || N < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_lists_plan: bad argument");
  const int TW = list_tw(N);
  if (!TW) return fail(h, SCOARY_ERR_SIZE, "scoary_lists_plan: N too large for LDS label tiles");
  if (G > 0x7fffffffLL - kSortItems)
    return fail(h, SCOARY_ERR_SIZE, "scoary_lists_plan: G >= 2^31");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const ListScratch L = list_scratch(G, N);
  char* sc = static_cast<char*>(d_scratch);
  int32_t* len = reinterpret_cast<int32_t*>(sc + L.len);
  int32_t* ord[2] = {reinterpret_cast<int32_t*>(sc + L.ord_a), reinterpret_cast<int32_t*>(sc + L.ord_b)};
  int32_t* hist = reinterpret_cast<int32_t*>(sc + L.hist);
  int64_t* base = reinterpret_cast<int64_t*>(sc + L.base);
  int32_t* padded = reinterpret_cast<int32_t*>(sc + L.padded);
  const int64_t Gp = scoary_tiled_genes(G);
  const int Qn = (int)((N + 127) / 128);
  const int gpw = kWave / list_lpg(TW);
  KernelTimer kt(h, s, "k_lists_plan");
  hipLaunchKernelGGL(k_lists_len, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const uint4*>(d_tiled), Gp, (int)G, (int)N, Qn, len, d_flipped);
  const int kmax = (int)(N / 2);               // minority lists hold <= N/2 entries
  int bits = 0;
  while ((kmax >> bits) != 0) ++bits;
  const int passes = bits <= 8 ? 1 : (bits <= 16 ? 2 : 3);
  for (int p = 0; p < passes; ++p) {
    const int32_t* in = p == 0 ? nullptr : ord[(p - 1) & 1];
    int32_t* out = p == passes - 1 ? d_order : ord[p & 1];
    hipLaunchKernelGGL(k_sort_hist, dim3((unsigned)L.nblk), dim3(64), 0, s, in, len, (int)G, kmax,
                       8 * p, (int)L.nblk, hist);
    hipLaunchKernelGGL(k_scan_excl, dim3(1), dim3(1024), 0, s, hist, (int64_t)256 * L.nblk);
    hipLaunchKernelGGL(k_sort_scatter, dim3((unsigned)L.nblk), dim3(64), 0, s, in, len, (int)G,
                       kmax, 8 * p, (int)L.nblk, hist, out);
  }
  if (L.nseg > 1) {           // N > 20479: sub-lists per segment, d_start / d_ngroups are [nseg][G]
    int32_t* seglen = reinterpret_cast<int32_t*>(sc + L.seglen);
    hipLaunchKernelGGL(k_lists_seglen, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(d_tiled), Gp, (int)G, (int)N, (int)L.nseg,
                       d_flipped, seglen);
    hipLaunchKernelGGL(k_lists_segpad, dim3((unsigned)((L.nwg + 255) / 256)), dim3(256), 0, s, seglen,
                       d_order, (int)G, gpw, L.nwg, (int)L.nseg, padded);
    hipLaunchKernelGGL(k_lists_base, dim3(1), dim3(1024), 0, s, padded, L.nwg * L.nseg, gpw, base);
    hipLaunchKernelGGL(k_lists_segslots, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, padded,
                       base, (int)G, gpw, (int)L.nseg, d_start, d_ngroups);
  } else {
    hipLaunchKernelGGL(k_lists_plan, dim3(1), dim3(1024), 0, s, len, d_order, (int)G, gpw, L.nwg,
                       padded, base);
    hipLaunchKernelGGL(k_lists_slots, dim3((unsigned)((G + 255) / 256)), dim3(256), 0, s, padded, base,
                       (int)G, gpw, d_start, d_ngroups);
  }
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(entries_out, base + L.nwg * L.nseg, sizeof(int64_t), hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  // entries_out counts 32-bit words of index array; the segmented sub-lists hold two 16-bit
  // entries per word (every sub-list is a multiple of 64 x 16 entries: the total is even)
  if (L.nseg > 1) *entries_out /= 2;
  return SCOARY_OK;
}

int scoary_lists_fill(scoary_handle h, const uint32_t* d_tiled, int64_t G, int64_t N,
                      const void* d_scratch, const int32_t* d_order, const uint8_t* d_flipped,
                      int64_t entries, uint32_t* d_idx, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_scratch || !d_order || !d_flipped || !d_idx || G < 1 || N < 1 || entries < 0)
    return fail(h, SCOARY_ERR_ARG, "scoary_lists_fill: bad argument");
  const int TW = list_tw(N);
  if (!TW) return fail(h, SCOARY_ERR_SIZE, "scoary_lists_fill: N too large for LDS label tiles");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const ListScratch L = list_scratch(G, N);
  const char* sc = static_cast<const char*>(d_scratch);
  const int32_t* len = reinterpret_cast<const int32_t*>(sc + L.len);
  const int64_t* base = reinterpret_cast<const int64_t*>(sc + L.base);
  const int32_t* padded = reinterpret_cast<const int32_t*>(sc + L.padded);
  const int64_t Gp = scoary_tiled_genes(G);
  const int Qn = (int)((N + 127) / 128);
  const int gpw = kWave / list_lpg(TW), piece = 4 * list_lpg(TW), C = 64 / TW;
  const int64_t nslots = L.nwg * gpw;
  const int spb = 4 * (64 / C);                      // slots per block of four wavefronts
  const dim3 grid((unsigned)((nslots + spb - 1) / spb));
  KernelTimer kt(h, s, "k_lists_fill");
  if (L.nseg > 1) {
    hipLaunchKernelGGL(k_lists_fill_seg, dim3((unsigned)((nslots + 7) / 8)), dim3(256), 0, s,
                       reinterpret_cast<const uint4*>(d_tiled), Gp, (int)G, (int)N, (int)L.nseg,
                       reinterpret_cast<const int32_t*>(sc + L.seglen), d_order, d_flipped, base,
                       padded, nslots, reinterpret_cast<uint16_t*>(d_idx));
    hipLaunchKernelGGL(k_lists_slack, dim3(1), dim3(256), 0, s, d_idx, entries);
    HIP_TRY(h, hipGetLastError());
    return SCOARY_OK;
  }
#define FILL(CV)                                                                               \
  hipLaunchKernelGGL((k_lists_fill<CV>), grid, dim3(256), 0, s,                                \
                     reinterpret_cast<const uint4*>(d_tiled), Gp, (int)G, (int)N, Qn, len,     \
                     d_order, d_flipped, base, padded, gpw, piece, (uint32_t)(TW * 4), nslots, \
                     d_idx)
  if (C == 4) FILL(4); else if (C == 8) FILL(8); else if (C == 16) FILL(16); else FILL(32);
#undef FILL
  hipLaunchKernelGGL(k_lists_slack, dim3(1), dim3(256), 0, s, d_idx, entries);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

}  // extern "C"
