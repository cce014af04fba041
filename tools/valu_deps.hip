// valu_deps.hip -- v_bitop3_b32 issue rate vs dependency distance and waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_deps.hip -o tools/valu_deps && tools/valu_deps
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v1","v2","v3"
// DIST independent in-place chains, 16 instructions per iteration
#define OP(N) "v_bitop3_b32 v" #N ", v" #N ", v2, v3 bitop3:0x96\n"
template <int DIST>
__global__ __launch_bounds__(64) void k(uint32_t* out, int iters) {
  asm volatile("v_mov_b32 v2, 0x7654321\n v_mov_b32 v3, 0x1111\n v_mov_b32 v16, 1\n v_mov_b32 v17, 2\n v_mov_b32 v18, 3\n v_mov_b32 v19, 4\n"
               "v_mov_b32 v20, 5\n v_mov_b32 v21, 6\n v_mov_b32 v22, 7\n v_mov_b32 v23, 8\n v_mov_b32 v24, 9\n v_mov_b32 v25, 10\n v_mov_b32 v26, 11\n"
               "v_mov_b32 v27, 12\n v_mov_b32 v28, 13\n v_mov_b32 v29, 14\n v_mov_b32 v30, 15\n v_mov_b32 v31, 16" ::: CLOB);
  for (int it = 0; it < iters; ++it) {
    if (DIST == 1) asm volatile(OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) OP(16) ::: CLOB);
    if (DIST == 2) asm volatile(OP(16) OP(17) OP(16) OP(17) OP(16) OP(17) OP(16) OP(17) OP(16) OP(17) OP(16) OP(17) OP(16) OP(17) OP(16) OP(17) ::: CLOB);
    if (DIST == 4) asm volatile(OP(16) OP(17) OP(18) OP(19) OP(16) OP(17) OP(18) OP(19) OP(16) OP(17) OP(18) OP(19) OP(16) OP(17) OP(18) OP(19) ::: CLOB);
    if (DIST == 8) asm volatile(OP(16) OP(17) OP(18) OP(19) OP(20) OP(21) OP(22) OP(23) OP(16) OP(17) OP(18) OP(19) OP(20) OP(21) OP(22) OP(23) ::: CLOB);
    if (DIST == 16) asm volatile(OP(16) OP(17) OP(18) OP(19) OP(20) OP(21) OP(22) OP(23) OP(24) OP(25) OP(26) OP(27) OP(28) OP(29) OP(30) OP(31) ::: CLOB);
  }
  uint32_t s;
  asm volatile("v_xor_b32 %0, v16, v17\n v_xor_b32 %0, %0, v18\n v_xor_b32 %0, %0, v31" : "=v"(s) :: CLOB);
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int DIST>
static void run(uint32_t* d, int waves) {
  const int blocks = 256 * 4 * waves, iters = 100000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<DIST>), dim3(blocks), dim3(64), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double ops = (double)blocks * 64 * iters * 16;
  printf("dependency distance %2d, %d waves/SIMD: %.2f cycles per wave-instruction (SIMD aggregate) at 2.4 GHz\n", DIST,
         waves, 1024.0 * 2.4e9 * 64 / (ops / (best * 1e-3)));
}
int main() {
  uint32_t* d;
  (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
  for (int w : {1, 2, 4, 8}) {
    run<1>(d, w); run<2>(d, w); run<4>(d, w); run<8>(d, w); run<16>(d, w);
  }
  return 0;
}
