// scoary_common.hpp -- shared by the translation units of libscoary_hip.so: the
// handle, error / timing helpers, layout constants and the Philox generator of
// spec S4 (scoary_labels.hip).  Everything here is internal; the contract is include/scoary_hip.h.
#ifndef SCOARY_COMMON_HPP
#define SCOARY_COMMON_HPP
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "scoary_hip.h"

struct scoary_ctx {
  int device = 0;
  int num_cu = 256;
  std::string err;
  bool timing = false;
  int lists_lds_optin = 0;   // k_permute_lists instances (by tile width) with the 160 KB LDS opt-in done
  int labels_lds_optin = 0;  // k_labels instances with it
  struct Timed {
    std::string name;
    hipEvent_t start, stop;
  };
  std::vector<Timed> timed;
};

namespace {

constexpr int kWave = 64;
constexpr int kGeneAlign = 256;

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

// ---- list-driven permutation path: layout constants shared by scoary_lists.hip
// (kernels) and scoary_listbuild.hip (device-side list builder) ----
// TW = tile row width in dwords (32 permutations each): 16 while a tile of 512
// permutations x (N+1) rows fits in LDS (N <= 2559), 8 (tiles of 256) up to
// N <= 5119, 4 (tiles of 128) up to N <= 10239, 2 (tiles of 64, two words per lane)
// beyond.  A gene takes max(TW/4, 1) lanes and a wavefront 64 / that many genes of
// similar list length.  A two-dword tile fits LDS whole up to N = 20479; wider matrices
// (N <= 131070) cut the isolates into SEGMENTS of kSegRows rows: a block loads one segment
// of its tile at a time and a gene's list is one sub-list per segment (k_permute_seglists;
// the counter planes live across the reloads).  (Round 3 first did this with one-dword
// tiles, 32 permutations per lane: 2.1x slower per listed row, tools/sweep_isolates.py.)
constexpr int kSegRows = 20352;                      // 159 * 128: whole word quads; (rows + 1) * 8 B fit 160 KB
constexpr int kSegTW = 2;                            // dwords per row of a segmented tile
constexpr int kSegStride = ((kSegRows + 1) * kSegTW + 3) / 4 * 4;   // dwords from one segment of a tile to the next
constexpr int kSegPiece = 8;                         // 16-bit entries per lane and index vector (16 bytes) of a segmented sub-list
constexpr int kMaxListIsolates = 131070;             // N / 2 < 2^16: sixteen counter planes; at most 7 segments
__host__ __device__ constexpr int list_segments(int64_t N) {
  return N <= 20479 ? 1 : (N <= kMaxListIsolates ? (int)((N + kSegRows - 1) / kSegRows) : 0);
}
__host__ __device__ constexpr int list_tw(int64_t N) {
  return N <= 2559 ? 16 : (N <= 5119 ? 8 : (N <= 10239 ? 4 : (list_segments(N) ? 2 : 0)));
}
__host__ __device__ constexpr int list_lpg(int TW) { return TW >= 4 ? TW / 4 : 1; }   // lanes per gene
__host__ __device__ constexpr int list_nw(int TW) { return TW >= 4 ? 4 : TW; }        // words per lane
// dwords per label tile in HBM: rows 0..N plus padding to a 16-byte multiple
__host__ __device__ constexpr int64_t list_tile_dwords(int64_t N, int TW) {
  return ((N + 1) * TW + 3) / 4 * 4;
}
// the same for any N the list path takes; segmented tiles (N > 20479): kSegStride dwords per
// segment, each with its own all-zero row after its last isolate
__host__ __device__ constexpr int64_t list_tile_dwords_seg(int64_t N, int TW) {
  return list_segments(N) > 1 ? (int64_t)list_segments(N) * kSegStride : list_tile_dwords(N, TW);
}
// rows of segment s, and the first dword of row `row` inside a two-dword tile
__host__ __device__ constexpr int64_t list_seg_rows(int64_t N, int s) {
  return list_segments(N) > 1 ? (N - (int64_t)s * kSegRows < kSegRows ? N - (int64_t)s * kSegRows : kSegRows) : N;
}
__host__ __device__ constexpr int64_t list_row_dword(int64_t N, int64_t row) {
  return list_segments(N) > 1 ? row / kSegRows * kSegStride + row % kSegRows * kSegTW : row * kSegTW;
}

constexpr int kListPad = 16;        // list lengths are padded to a multiple of this many entries (half a 32-entry step)
constexpr int kListStartUnit = 32;  // d_lstart counts in units of this many entries (128 bytes)
constexpr int kListChunkMB = 2;     // MB of index lists per k_permute_lists block (an XCD's L2 is 4 MB)
constexpr int kListSlack = 256;     // zero entries after the last list (one wavefront index load)

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

__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32x32->64 multiply per product (v_mad_u64_u32) instead of hi + lo
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t h0 = (uint32_t)(p0 >> 32), l0 = (uint32_t)p0;
    const uint32_t h1 = (uint32_t)(p1 >> 32), l1 = (uint32_t)p1;
    // three-input xor = one v_bitop3_b32 (LUT 0x96); two v_xor_b32 otherwise
    const uint32_t n0 = __builtin_amdgcn_bitop3_b32(h1, c1, k0, 0x96), n2 = __builtin_amdgcn_bitop3_b32(h0, c3, k1, 0x96);
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

// acc += popcount(x) as ONE v_bcnt_u32_b32 (its second operand is the
// accumulator).  Opaque to the optimiser on purpose: left to itself LLVM
// reassociates the accumulate chain into short chains joined by v_add3_u32,
// ~15 % more VALU work in the permutation inner loop.
__device__ __forceinline__ void bcnt_acc(uint32_t& acc, uint32_t x) {
  asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc) : "v"(x));
}

}  // namespace
#endif
