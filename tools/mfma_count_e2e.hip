// mfma_count_e2e.hip -- EVIDENCE ONLY, not product (north_star excludes MFMA from the path).
//
// tools/mfma_count_probe.hip measured the inner loop a matrix-core formulation of the permutation counts would
// have (fp4 x fp4 v_mfma_scale_f32_32x32x64_f8f6f4 fed from LDS).  This is the whole thing, end to end, so that
// the record holds one honest number for what the exclusion costs on dense genes:
//     count[g][pi] = sum_i gene[g][i] * label[i][pi]         (0/1 operands as fp4 1.0 / 0.0, exact in fp32)
//     r[t][g]      = #{pi : count outside the acceptance interval of (g, t)}          (spec S5)
// from the SAME inputs the product kernels read -- the tiled 1-bit gene matrix, the 1-bit label rows of
// scoary_perm_generate, the (base, span) regions of scoary_fisher -- and checked bit-identical to scoary_permute
// by tools/mfma_count_e2e.py.  Block = 128 genes x (all permutations of one trait, 128 at a time): per 128
// isolates every thread fetches one 16-byte quad of bits (a gene row or a label row), expands it to fp4 in
// registers (bit -> nibble 0b0010) and stores 64 bytes to LDS (double-buffered, next quad prefetched); the four
// wavefronts each own a 64 x 64 corner (2 x 2 MFMA tiles, accumulators in registers).  After the K loop the region
// test runs on the accumulators (ballot + popcount per row) into per-gene counters in LDS; one plain store per
// (gene, trait) at the end -- no atomics in HBM.
//
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/mfma_count_e2e.hip -o tools/mfma_count_e2e.so
#include <hip/hip_runtime.h>
#include <cstdint>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

namespace {
constexpr int BT = 128;   // block tile: genes x permutations
constexpr int RS = 80;    // LDS row stride, bytes: 128 fp4 values (64 B) + 16 B of padding against bank conflicts

// 8 presence bits -> 8 fp4 nibbles (bit i -> nibble i = 0b0010 = 1.0)
__device__ __forceinline__ uint32_t spread8(uint32_t x) {
  x = (x | (x << 12)) & 0x000F000Fu;
  x = (x | (x << 6)) & 0x03030303u;
  x = (x | (x << 3)) & 0x11111111u;
  return x << 1;
}
__device__ __forceinline__ void store_row(unsigned char* row, const uint4 q) {
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {                       // word k = isolates 32k .. 32k + 31 of the quad
    v4i o;
    o.x = (int)spread8(w[k] & 0xffu);
    o.y = (int)spread8((w[k] >> 8) & 0xffu);
    o.z = (int)spread8((w[k] >> 16) & 0xffu);
    o.w = (int)spread8(w[k] >> 24);
    *reinterpret_cast<v4i*>(row + 16 * k) = o;
  }
}

__global__ __launch_bounds__(256) void k_mfma_exceed(const uint4* __restrict__ tiled, int64_t Gp, int Qp,
                                                     const uint4* __restrict__ perms, int64_t P,
                                                     const uint2* __restrict__ crit, int64_t G,
                                                     uint32_t* __restrict__ r) {
  __shared__ __attribute__((aligned(16))) unsigned char sA[2][BT * RS];
  __shared__ __attribute__((aligned(16))) unsigned char sB[2][BT * RS];
  __shared__ uint32_t s_cnt[BT];
  __shared__ uint2 s_crit[BT];
  const int t = blockIdx.y;
  const int64_t g0 = (int64_t)blockIdx.x * BT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;             // the wavefront's 64 x 64 corner of the block tile
  const int half = lane >> 5, l31 = lane & 31;
  if (tid < BT) {
    s_cnt[tid] = 0u;
    s_crit[tid] = g0 + tid < G ? crit[(int64_t)t * G + g0 + tid] : make_uint2(0u, 0xffffffffu);
  }
  const bool is_a = tid < BT;
  const int row_ld = is_a ? tid : tid - BT;            // the gene / label row this thread stages
  const uint4* a_src = tiled + g0 + row_ld;            // + q * Gp
  for (int64_t p0 = 0; p0 < P; p0 += BT) {
    const uint4* b_src = perms + ((int64_t)t * P + p0 + row_ld) * Qp;     // + q
    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    uint4 nxt = is_a ? a_src[0] : b_src[0];
    for (int q = 0; q < Qp; ++q) {
      unsigned char* bufA = sA[q & 1];
      unsigned char* bufB = sB[q & 1];
      store_row((is_a ? bufA : bufB) + row_ld * RS, nxt);
      __syncthreads();                                  // one barrier per step: two buffers
      if (q + 1 < Qp) nxt = is_a ? a_src[(int64_t)(q + 1) * Gp] : b_src[q + 1];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {                  // two K steps of 64 isolates per quad
        v8i a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const v4i x = *reinterpret_cast<const v4i*>(bufA + (wi * 64 + i * 32 + l31) * RS + ks * 32 + half * 16);
          a[i] = v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const v4i x = *reinterpret_cast<const v4i*>(bufB + (wj * 64 + j * 32 + l31) * RS + ks * 32 + half * 16);
          b[j] = v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 4, 4, 0, 0x7f7f7f7f,
                                                                        0, 0x7f7f7f7f);
      }
    }
    // region test on the accumulators: C/D layout col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = wi * 64 + i * 32 + (k & 3) + 8 * (k >> 2) + 4 * half;
        const uint2 c = s_crit[row];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint32_t cntv = (uint32_t)acc[i][j][k];
          const uint64_t bal = __ballot((cntv - c.x) >= c.y);
          cnt += __popc(half ? (uint32_t)(bal >> 32) : (uint32_t)bal);
        }
        if (l31 == 0) atomicAdd(&s_cnt[row], (uint32_t)cnt);
      }
    __syncthreads();                                    // the LDS tiles are free again for the next column tile
  }
  if (tid < BT && g0 + tid < G) r[(int64_t)t * G + g0 + tid] = s_cnt[tid];
}

// Third version = the streaming kernel with the two cheap fixes its profile asks for: (1) bit -> fp4 expansion
// through v_perm_b32 on a 4-byte pool (5 ops per 8 isolates instead of 8); (2) the region test keeps one counter per
// accumulator element in a register across the permutation tiles -- a block's rows are the same genes for every
// tile -- so a tile costs convert + subtract + compare + add per element, and the 32 lanes of a row meet only once,
// at the end (shuffles), instead of a ballot, two popcounts and an LDS atomic per row and tile.  Neither moves the
// time (6.6 against 6.5 ms at cfg3): the loop is bound by its barrier per 128 isolates and by operand loads
// that are only one step (eight MFMAs, ~300 clocks) ahead of their use -- what a real GEMM pipeline fixes with a
// multi-stage ring; a first attempt at one (runtime ring index: 80 bytes of scratch) was slower still (10.4 ms).
__device__ __forceinline__ uint32_t spread8p(uint32_t w, int byte);
__global__ __launch_bounds__(256) void k_mfma_exceed_v3(const uint4* __restrict__ tiled, int64_t Gp, int Qp,
                                                        const uint4* __restrict__ perms, int64_t P,
                                                        const uint2* __restrict__ crit, int64_t G,
                                                        uint32_t* __restrict__ r) {
  __shared__ __attribute__((aligned(16))) unsigned char sA[2][BT * RS];
  __shared__ __attribute__((aligned(16))) unsigned char sB[2][BT * RS];
  __shared__ uint32_t s_cnt[BT];
  const int t = blockIdx.y;
  const int64_t g0 = (int64_t)blockIdx.x * BT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  if (tid < BT) s_cnt[tid] = 0u;
  uint2 c[2][16];                                      // the regions of this lane's 32 rows
  uint32_t ex[2][16];                                  // exceedances seen so far, per accumulator element
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const int64_t g = g0 + wi * 64 + i * 32 + (k & 3) + 8 * (k >> 2) + 4 * half;
      c[i][k] = g < G ? crit[(int64_t)t * G + g] : make_uint2(0u, 0xffffffffu);
      ex[i][k] = 0u;
    }
  const bool is_a = tid < BT;
  const int row_ld = is_a ? tid : tid - BT;
  const uint4* a_src = tiled + g0 + row_ld;
  for (int64_t p0 = 0; p0 < P; p0 += BT) {
    const uint4* b_src = perms + ((int64_t)t * P + p0 + row_ld) * Qp;
    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    uint4 nxt = is_a ? a_src[0] : b_src[0];
    for (int q = 0; q < Qp; ++q) {
      unsigned char* bufA = sA[q & 1];
      unsigned char* bufB = sB[q & 1];
      {
        unsigned char* row = (is_a ? bufA : bufB) + row_ld * RS;
        const uint32_t w[4] = {nxt.x, nxt.y, nxt.z, nxt.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          *reinterpret_cast<v4i*>(row + 16 * k) = v4i{(int)spread8p(w[k], 0), (int)spread8p(w[k], 1),
                                                      (int)spread8p(w[k], 2), (int)spread8p(w[k], 3)};
      }
      __syncthreads();
      if (q + 1 < Qp) nxt = is_a ? a_src[(int64_t)(q + 1) * Gp] : b_src[q + 1];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        v8i a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const v4i x = *reinterpret_cast<const v4i*>(bufA + (wi * 64 + i * 32 + l31) * RS + ks * 32 + half * 16);
          a[i] = v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const v4i x = *reinterpret_cast<const v4i*>(bufB + (wj * 64 + j * 32 + l31) * RS + ks * 32 + half * 16);
          b[j] = v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 4, 4, 0, 0x7f7f7f7f,
                                                                        0, 0x7f7f7f7f);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          ex[i][k] += ((uint32_t)acc[i][j][k] - c[i][k].x) >= c[i][k].y ? 1u : 0u;
    __syncthreads();                                    // the LDS tiles are free again for the next column tile
  }
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      uint32_t v = ex[i][k];
#pragma unroll
      for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off);     // the 32 lanes (columns) of the row
      if (l31 == 0) atomicAdd(&s_cnt[wi * 64 + i * 32 + (k & 3) + 8 * (k >> 2) + 4 * half], v);
    }
  __syncthreads();
  if (tid < BT && g0 + tid < G) r[(int64_t)t * G + g0 + tid] = s_cnt[tid];
}

// Second version, for rows that fit LDS whole (N <= 2048: cfg3): the block's 128 gene rows are expanded to fp4
// ONCE and stay in LDS for every permutation tile (128 x 1040 B); only the label rows are staged per step, every
// thread expanding half a quad.  Halves the expansion work and takes the gene loads out of the loop.
__device__ __forceinline__ uint32_t spread8p(uint32_t w, int byte) {
  const uint32_t b = (w >> (8 * byte)) & 0xffu;
  uint32_t y = b | (b << 12);
  y = (y | (y << 6)) & 0x03030303u;
  return __builtin_amdgcn_perm(0u, 0x22200200u, y);
}
__device__ __forceinline__ uint32_t spread8_perm(uint32_t w, int byte) {
  // 2 bits -> one byte of two fp4 values via v_perm_b32 on the pool {0x00, 0x02, 0x20, 0x22}
  const uint32_t b = (w >> (8 * byte)) & 0xffu;
  uint32_t y = b | (b << 12);
  y = (y | (y << 6)) & 0x03030303u;
  return __builtin_amdgcn_perm(0u, 0x22200200u, y);
}
template <int QP>
__global__ __launch_bounds__(256) void k_mfma_exceed_resident(const uint4* __restrict__ tiled, int64_t Gp,
                                                              const uint2* __restrict__ perms, int64_t P,
                                                              const uint2* __restrict__ crit, int64_t G,
                                                              uint32_t* __restrict__ r) {
  constexpr int ARS = QP * 64 + 16;                    // bytes per resident gene row (260 dwords: conflict-free b128 reads)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  unsigned char* sA = lds;                             // [BT][ARS]
  unsigned char* sBq = lds + BT * ARS;                 // [2][BT][RS]
  uint32_t* s_cnt = reinterpret_cast<uint32_t*>(sBq + 2 * BT * RS);
  uint2* s_crit = reinterpret_cast<uint2*>(s_cnt + BT);
  const int t = blockIdx.y;
  const int64_t g0 = (int64_t)blockIdx.x * BT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  if (tid < BT) {
    s_cnt[tid] = 0u;
    s_crit[tid] = g0 + tid < G ? crit[(int64_t)t * G + g0 + tid] : make_uint2(0u, 0xffffffffu);
  }
  const int row_ld = tid >> 1, part = tid & 1;         // two threads per staged row
  for (int q = part; q < QP; q += 2) {                 // the gene rows, once
    const uint4 v = tiled[(int64_t)q * Gp + g0 + row_ld];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      *reinterpret_cast<v4i*>(sA + row_ld * ARS + q * 64 + 16 * k) =
          v4i{(int)spread8_perm(w[k], 0), (int)spread8_perm(w[k], 1), (int)spread8_perm(w[k], 2),
              (int)spread8_perm(w[k], 3)};
  }
  for (int64_t p0 = 0; p0 < P; p0 += BT) {
    const uint2* b_src = perms + (((int64_t)t * P + p0 + row_ld) * QP) * 2 + part;      // + 2 q
    v16f acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
    uint2 nxt = b_src[0];
#pragma unroll 2
    for (int q = 0; q < QP; ++q) {
      unsigned char* bufB = sBq + (q & 1) * (BT * RS);
      {
        const uint32_t w[2] = {nxt.x, nxt.y};
#pragma unroll
        for (int k = 0; k < 2; ++k)
          *reinterpret_cast<v4i*>(bufB + row_ld * RS + part * 32 + 16 * k) =
              v4i{(int)spread8_perm(w[k], 0), (int)spread8_perm(w[k], 1), (int)spread8_perm(w[k], 2),
                  (int)spread8_perm(w[k], 3)};
      }
      __syncthreads();
      if (q + 1 < QP) nxt = b_src[2 * (q + 1)];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        v8i a[2], b[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const v4i x = *reinterpret_cast<const v4i*>(sA + (wi * 64 + i * 32 + l31) * ARS + q * 64 + ks * 32 + half * 16);
          a[i] = v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0};
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const v4i x = *reinterpret_cast<const v4i*>(bufB + (wj * 64 + j * 32 + l31) * RS + ks * 32 + half * 16);
          b[j] = v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 4, 4, 0, 0x7f7f7f7f,
                                                                        0, 0x7f7f7f7f);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int row = wi * 64 + i * 32 + (k & 3) + 8 * (k >> 2) + 4 * half;
        const uint2 c = s_crit[row];
        int cnt = 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const uint32_t cntv = (uint32_t)acc[i][j][k];
          const uint64_t bal = __ballot((cntv - c.x) >= c.y);
          cnt += __popc(half ? (uint32_t)(bal >> 32) : (uint32_t)bal);
        }
        if (l31 == 0) atomicAdd(&s_cnt[row], (uint32_t)cnt);
      }
    __syncthreads();
  }
  if (tid < BT && g0 + tid < G) r[(int64_t)t * G + g0 + tid] = s_cnt[tid];
}
}  // namespace

// tiled: the product's word-quad-major gene matrix [Qp][Gp][4]; perms: label rows [T][P][Wp = 4 Qp] of
// scoary_perm_generate; crit: (base, span) [T][G] of scoary_fisher; r: uint32 [T][G] (overwritten).
// P must be a multiple of 128 (the evidence runs use P = 10240).
// variant: 0 = streaming (any N), 1 = gene rows resident in LDS (Qp == 16 only: N <= 2048), 2 = streaming with
// v_perm expansion and per-lane exceedance counters
extern "C" int mfma_exceed(const uint32_t* tiled, int64_t Gp, int64_t Qp, const uint32_t* perms, int64_t T, int64_t P,
                           const uint32_t* crit, int64_t G, uint32_t* r, int variant, void* stream) {
  if (P % BT != 0 || Gp % BT != 0) return -1;
  const dim3 grid((unsigned)((G + BT - 1) / BT), (unsigned)T);
  if (variant == 2) {
    hipLaunchKernelGGL(k_mfma_exceed_v3, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint4*>(tiled), Gp, (int)Qp, reinterpret_cast<const uint4*>(perms), P,
                       reinterpret_cast<const uint2*>(crit), G, r);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  if (variant == 1) {
    if (Qp != 16) return -3;
    constexpr int QP = 16;
    const size_t lds = (size_t)BT * (QP * 64 + 16) + 2 * BT * RS + BT * 4 + BT * 8;
    static bool optin = false;
    if (!optin) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mfma_exceed_resident<QP>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -4;
      optin = true;
    }
    hipLaunchKernelGGL((k_mfma_exceed_resident<QP>), grid, dim3(256), lds, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint4*>(tiled), Gp, reinterpret_cast<const uint2*>(perms), P,
                       reinterpret_cast<const uint2*>(crit), G, r);
    return hipGetLastError() == hipSuccess ? 0 : -2;
  }
  hipLaunchKernelGGL(k_mfma_exceed, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                     reinterpret_cast<const uint4*>(tiled), Gp, (int)Qp, reinterpret_cast<const uint4*>(perms), P,
                     reinterpret_cast<const uint2*>(crit), G, r);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
