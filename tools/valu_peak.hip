// valu_peak.hip -- measures the chip's sustained issue rate for the exact op
// mix of k_permute's inner loop (v_and_b32 with an SGPR operand feeding
// v_bcnt_u32_b32 accumulate), with no memory traffic at all.  The result is the
// practical VALU ceiling the permutation kernel is priced against in DESIGN.md.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int NR, bool BATCH>
__global__ __launch_bounds__(64) void k_and_bcnt(uint32_t* out, int iters, uint32_t seed) {
  uint32_t g[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) g[i] = (threadIdx.x * 2654435761u) ^ (i * 40503u) ^ seed;
  uint32_t acc[4] = {0, 0, 0, 0};
  uint32_t s = seed;
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
    s = s * 1664525u + 1013904223u;  // scalar (SALU) LCG: uniform operand
    if (!BATCH) {
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        uint32_t x = g[i] & s;
        asm("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i & 3]) : "v"(x));
      }
    } else {
#pragma unroll
      for (int i = 0; i < NR; i += 4) {
        uint32_t x0, x1, x2, x3;
        asm volatile("v_and_b32 %0, %4, %5\n\tv_and_b32 %1, %4, %6\n\tv_and_b32 %2, %4, %7\n\t"
                     "v_and_b32 %3, %4, %8"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3)
                     : "s"(s), "v"(g[i]), "v"(g[i + 1]), "v"(g[i + 2]), "v"(g[i + 3]));
        asm volatile("v_bcnt_u32_b32 %0, %4, %0\n\tv_bcnt_u32_b32 %1, %5, %1\n\t"
                     "v_bcnt_u32_b32 %2, %6, %2\n\tv_bcnt_u32_b32 %3, %7, %3"
                     : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
                     : "v"(x0), "v"(x1), "v"(x2), "v"(x3));
      }
    }
  }
  out[blockIdx.x * 64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

__global__ __launch_bounds__(64) void k_fma(float* out, int iters, float a) {
  float x[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, 1.0f);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

// 16 independent chains of one instruction, all-VGPR operands: the issue rate of
// the op itself.  KIND 0: v_bitop3_b32 (xor3), 1: v_xor_b32, 2: v_add_u32,
// 3: v_add_u32_dpp quad_perm, 4: v_bitop3 full adder pair (sum + carry of the same inputs)
template <int KIND>
__global__ __launch_bounds__(64) void k_op(uint32_t* out, int iters, uint32_t seed) {
  uint32_t x[16], y = threadIdx.x * 2654435761u ^ seed, z = threadIdx.x * 40503u + seed;
#pragma unroll
  for (int i = 0; i < 16; ++i) x[i] = (threadIdx.x + i) * 2246822519u ^ seed;
#pragma clang loop unroll(disable)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (KIND == 0)
        asm volatile("v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96" : "+v"(x[i]) : "v"(y), "v"(z));
      else if (KIND == 1)
        asm volatile("v_xor_b32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
      else if (KIND == 2)
        asm volatile("v_add_u32 %0, %0, %1" : "+v"(x[i]) : "v"(y));
      else if (KIND == 3)
        asm volatile("v_add_u32_dpp %0, %1, %0 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf"
                     : "+v"(x[i]) : "v"(y));
      else {
        uint32_t c;
        asm volatile("v_bitop3_b32 %1, %0, %2, %3 bitop3:0xe8\n\tv_bitop3_b32 %0, %0, %2, %3 bitop3:0x96"
                     : "+v"(x[i]), "=&v"(c) : "v"(y), "v"(z));
        y ^= c & 1u ? 0u : 0u;   // keep c alive without extra VALU work being measured
      }
    }
  }
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s ^= x[i];
  out[blockIdx.x * 64 + threadIdx.x] = s ^ y;
}

template <int KIND>
static void run_op(const char* name, uint32_t* d, int blocks, int iters, int ops_per_slot) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_op<KIND>), dim3(blocks), dim3(64), 0, 0, d, iters, 777u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)blocks * 64 * iters * 16 * ops_per_slot;
    printf("%s: %.3f ms  %.3e lane-ops/s = %.1f%% of 256CU*4*16*2.4GHz\n", name, ms, ops / (ms * 1e-3),
           100.0 * ops / (ms * 1e-3) / (256.0 * 4 * 16 * 2.4e9));
  }
}

int main() {
  const int blocks = 256 * 4 * 8;  // 8 waves per SIMD
  uint32_t* d;
  hipMalloc(&d, blocks * 64 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int rep = 0; rep < 6; ++rep) {
    hipEventRecord(e0);
    if (rep < 3)
      hipLaunchKernelGGL((k_and_bcnt<64, false>), dim3(blocks), dim3(64), 0, 0, d, iters, 12345u);
    else
      hipLaunchKernelGGL((k_and_bcnt<64, true>), dim3(blocks), dim3(64), 0, 0, d, iters, 12345u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 64 * iters * 64 * 2;  // and + bcnt per register
    printf("and+bcnt (s-operand, %s): %.3f ms  %.3e lane-ops/s  = %.1f%% of 256CU*4*32*2.4GHz\n", rep < 3 ? "and->bcnt back to back" : "4 and then 4 bcnt", ms,
           ops / (ms * 1e-3), 100.0 * ops / (ms * 1e-3) / (256.0 * 4 * 32 * 2.4e9));
  }
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(64), 0, 0, (float*)d, iters * 4, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)blocks * 64 * iters * 4 * 16;
    printf("v_fma_f32: %.3f ms  %.3e lane-ops/s = %.1f%%  (%.1f TFLOP/s)\n", ms, ops / (ms * 1e-3),
           100.0 * ops / (ms * 1e-3) / (256.0 * 4 * 32 * 2.4e9), 2 * ops / (ms * 1e-3) / 1e12);
  }
  run_op<0>("v_bitop3_b32 (xor3)", d, blocks, iters * 4, 1);
  run_op<4>("v_bitop3_b32 pair (carry + sum)", d, blocks, iters * 2, 2);
  run_op<1>("v_xor_b32", d, blocks, iters * 4, 1);
  run_op<2>("v_add_u32", d, blocks, iters * 4, 1);
  run_op<3>("v_add_u32_dpp quad_perm", d, blocks, iters * 4, 1);
  return 0;
}
