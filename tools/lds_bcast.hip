// lds_bcast.hip -- cost of ds_read_b64 / ds_read_b128 address patterns on gfx950 (LDS pipe cycles
// per wavefront instruction, 16 waves per CU all issuing reads back to back).
//   hipcc --offload-arch=gfx950 -O3 tools/lds_bcast.hip -o tools/lds_bcast && tools/lds_bcast
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int WIDTH>
__global__ __launch_bounds__(1024) void k(uint32_t* out, int iters, int pattern) {
  extern __shared__ uint32_t lds[];
  for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lg = lane >> 2, q = lane & 3;
  constexpr int WB = WIDTH == 2 || WIDTH == 6 ? 8 : 16;   // bytes per lane slot
  uint32_t a = 0;
  switch (pattern) {
    case 0: a = lane * (WB); break;                         // dense, all lanes distinct
    case 1: a = lg * 64; break;                                    // 4-lane broadcast, genes 64 B apart
    case 2: a = lg * 64 + ((lg >> 2) & 1) * 8; break;              // + swizzle by 8 B
    case 3: a = 0; break;                                          // whole-wave broadcast
    case 4: a = lg * (WB); break;                           // 4-lane broadcast, genes dense
    case 5: a = lg * 80; break;                                    // 4-lane broadcast, 80-B stride
    case 6: a = (lane >> 1) * (WB); break;                  // 2-lane broadcast, dense
    case 7: a = lg * 64 + q * 16; break;                           // no broadcast, 16 B per lane slots
    case 8: a = (lane & 15) * (WB); break;                  // 16 distinct, repeated per 16 lanes
    case 9: a = lg * 72; break;                                    // 72-B stride
  }
  a += wave * 4096;
  uint32_t acc = 0;
  for (int it = 0; it < iters; ++it) {
    if (WIDTH == 2) {
      uint2 v0, v1, v2, v3;
      asm volatile("ds_read_b64 %0, %4\n ds_read_b64 %1, %4 offset:1024\n ds_read_b64 %2, %4 offset:2048\n"
                   "ds_read_b64 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a));
      acc += v0.x ^ v1.y ^ v2.x ^ v3.y;
    } else if (WIDTH == 3) {            // ds_read2_b64: two 8-byte reads per lane
      uint4 v0, v1, v2, v3;
      asm volatile("ds_read2_b64 %0, %4 offset0:0 offset1:2\n ds_read2_b64 %1, %4 offset0:4 offset1:6\n"
                   "ds_read2_b64 %2, %4 offset0:128 offset1:130\n ds_read2_b64 %3, %4 offset0:132 offset1:134\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a));
      acc += v0.x ^ v1.y ^ v2.z ^ v3.w;
    } else if (WIDTH == 5) {            // ds_write_b128
      u32x4 v = {acc, acc, acc, (uint32_t)it};
      asm volatile("ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:1024\n ds_write_b128 %0, %1 offset:2048\n"
                   "ds_write_b128 %0, %1 offset:3072\n s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v) : "memory");
    } else if (WIDTH == 6) {            // ds_write_b64
      u32x2 v = {acc, (uint32_t)it};
      asm volatile("ds_write_b64 %0, %1\n ds_write_b64 %0, %1 offset:1024\n ds_write_b64 %0, %1 offset:2048\n"
                   "ds_write_b64 %0, %1 offset:3072\n s_waitcnt lgkmcnt(0)" :: "v"(a), "v"(v) : "memory");
    } else {
      uint4 v0, v1, v2, v3;
      asm volatile("ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:1024\n ds_read_b128 %2, %4 offset:2048\n"
                   "ds_read_b128 %3, %4 offset:3072\n s_waitcnt lgkmcnt(0)"
                   : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a));
      acc += v0.x ^ v1.y ^ v2.z ^ v3.w;
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int WIDTH>
static void run(const char* name, uint32_t* d, int pattern) {
  const int blocks = 256, iters = 20000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<WIDTH>), dim3(blocks), dim3(1024), 96 * 1024, 0, d, iters, pattern);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  // per CU: 16 waves x iters x 4 reads
  const double cyc = best * 1e-3 * 2.4e9 / (16.0 * iters * 4);
  printf("%-13s %-44s %.2f LDS cycles per wavefront read\n", WIDTH == 2 ? "ds_read_b64" : WIDTH == 4 ? "ds_read_b128" : WIDTH == 3 ? "ds_read2_b64" : WIDTH == 5 ? "ds_write_b128" : "ds_write_b64", name, cyc);
  fflush(stdout);
}
int main() {
  uint32_t* d;
  (void)hipMalloc(&d, 256 * 1024 * 4);
  const char* names[] = {"dense (lane * width)", "4-lane broadcast, 64-B gene stride", "same + 8-B swizzle on gene bit 2",
                         "whole-wave broadcast", "4-lane broadcast, genes dense", "4-lane broadcast, 80-B stride",
                         "2-lane broadcast, dense", "gene*64 + quarter*16 (no broadcast)", "16 distinct, x4 across the wave",
                         "4-lane broadcast, 72-B stride"};
  for (int p = 0; p < 10; ++p) run<2>(names[p], d, p);
  for (int p = 0; p < 10; ++p) run<4>(names[p], d, p);
  for (int p : {1, 2, 4}) run<3>(names[p], d, p);
  for (int p : {0, 7}) run<5>(names[p], d, p);
  for (int p : {0, 4}) run<6>(names[p], d, p);
  return 0;
}
