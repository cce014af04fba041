// valu_bcnt.hip -- the dense kernel's op pair with hard registers: v_and_b32 d, s, g ; v_bcnt_u32_b32 acc, d, acc.
// Does the VGPR bank of d vs acc matter?   hipcc --offload-arch=gfx950 -O3 tools/valu_bcnt.hip -o tools/valu_bcnt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v40","v41","v42","v43","v44","v45","v46","v47","s40"
// KIND 0: and -> v40..v43 (banks 0..3), bcnt acc vN (bank N%4) with src0 bank == acc bank  (conflict)
// KIND 1: same, src0 bank != acc bank
// KIND 2: bcnt only, conflict / KIND 3: bcnt only, no conflict / KIND 4: and only
template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t* out, int iters) {
  asm volatile("s_mov_b32 s40, 0x0f0f0f0f\n v_mov_b32 v40, 1\n v_mov_b32 v41, 2\n v_mov_b32 v42, 3\n v_mov_b32 v43, 4\n"
               "v_mov_b32 v16, %0\n v_mov_b32 v17, %0\n v_mov_b32 v18, %0\n v_mov_b32 v19, %0\n v_mov_b32 v20, 0\n v_mov_b32 v21, 0\n v_mov_b32 v22, 0\n v_mov_b32 v23, 0\n"
               "v_mov_b32 v24, 0\n v_mov_b32 v25, 0\n v_mov_b32 v26, 0\n v_mov_b32 v27, 0" :: "v"(threadIdx.x * 2654435761u) : CLOB);
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0)   // d = v40 (bank 0) -> acc v20,v24 (bank 0); d = v41 (bank 1) -> acc v21, v25 ...
      asm volatile("v_and_b32 v40, s40, v16\n v_bcnt_u32_b32 v20, v40, v20\n v_and_b32 v41, s40, v17\n v_bcnt_u32_b32 v21, v41, v21\n"
                   "v_and_b32 v42, s40, v18\n v_bcnt_u32_b32 v22, v42, v22\n v_and_b32 v43, s40, v19\n v_bcnt_u32_b32 v23, v43, v23\n"
                   "v_and_b32 v40, s40, v16\n v_bcnt_u32_b32 v24, v40, v24\n v_and_b32 v41, s40, v17\n v_bcnt_u32_b32 v25, v41, v25\n"
                   "v_and_b32 v42, s40, v18\n v_bcnt_u32_b32 v26, v42, v26\n v_and_b32 v43, s40, v19\n v_bcnt_u32_b32 v27, v43, v27\n" ::: CLOB);
    if (KIND == 1)   // d bank = acc bank + 1
      asm volatile("v_and_b32 v41, s40, v16\n v_bcnt_u32_b32 v20, v41, v20\n v_and_b32 v42, s40, v17\n v_bcnt_u32_b32 v21, v42, v21\n"
                   "v_and_b32 v43, s40, v18\n v_bcnt_u32_b32 v22, v43, v22\n v_and_b32 v40, s40, v19\n v_bcnt_u32_b32 v23, v40, v23\n"
                   "v_and_b32 v41, s40, v16\n v_bcnt_u32_b32 v24, v41, v24\n v_and_b32 v42, s40, v17\n v_bcnt_u32_b32 v25, v42, v25\n"
                   "v_and_b32 v43, s40, v18\n v_bcnt_u32_b32 v26, v43, v26\n v_and_b32 v40, s40, v19\n v_bcnt_u32_b32 v27, v40, v27\n" ::: CLOB);
    if (KIND == 2)
      asm volatile("v_bcnt_u32_b32 v20, v40, v20\n v_bcnt_u32_b32 v21, v41, v21\n v_bcnt_u32_b32 v22, v42, v22\n v_bcnt_u32_b32 v23, v43, v23\n"
                   "v_bcnt_u32_b32 v24, v40, v24\n v_bcnt_u32_b32 v25, v41, v25\n v_bcnt_u32_b32 v26, v42, v26\n v_bcnt_u32_b32 v27, v43, v27\n"
                   "v_bcnt_u32_b32 v20, v40, v20\n v_bcnt_u32_b32 v21, v41, v21\n v_bcnt_u32_b32 v22, v42, v22\n v_bcnt_u32_b32 v23, v43, v23\n"
                   "v_bcnt_u32_b32 v24, v40, v24\n v_bcnt_u32_b32 v25, v41, v25\n v_bcnt_u32_b32 v26, v42, v26\n v_bcnt_u32_b32 v27, v43, v27\n" ::: CLOB);
    if (KIND == 3)
      asm volatile("v_bcnt_u32_b32 v20, v41, v20\n v_bcnt_u32_b32 v21, v42, v21\n v_bcnt_u32_b32 v22, v43, v22\n v_bcnt_u32_b32 v23, v40, v23\n"
                   "v_bcnt_u32_b32 v24, v41, v24\n v_bcnt_u32_b32 v25, v42, v25\n v_bcnt_u32_b32 v26, v43, v26\n v_bcnt_u32_b32 v27, v40, v27\n"
                   "v_bcnt_u32_b32 v20, v41, v20\n v_bcnt_u32_b32 v21, v42, v21\n v_bcnt_u32_b32 v22, v43, v22\n v_bcnt_u32_b32 v23, v40, v23\n"
                   "v_bcnt_u32_b32 v24, v41, v24\n v_bcnt_u32_b32 v25, v42, v25\n v_bcnt_u32_b32 v26, v43, v26\n v_bcnt_u32_b32 v27, v40, v27\n" ::: CLOB);
    if (KIND == 4)
      asm volatile("v_and_b32 v40, s40, v16\n v_and_b32 v41, s40, v17\n v_and_b32 v42, s40, v18\n v_and_b32 v43, s40, v19\n"
                   "v_and_b32 v44, s40, v16\n v_and_b32 v45, s40, v17\n v_and_b32 v46, s40, v18\n v_and_b32 v47, s40, v19\n"
                   "v_and_b32 v40, s40, v16\n v_and_b32 v41, s40, v17\n v_and_b32 v42, s40, v18\n v_and_b32 v43, s40, v19\n"
                   "v_and_b32 v44, s40, v16\n v_and_b32 v45, s40, v17\n v_and_b32 v46, s40, v18\n v_and_b32 v47, s40, v19\n" ::: CLOB);
  }
  uint32_t s;
  asm volatile("v_add_u32 %0, v20, v21\n v_add_u32 %0, %0, v22\n v_add_u32 %0, %0, v27\n v_add_u32 %0, %0, v44" : "=v"(s) :: CLOB);
  out[blockIdx.x * 64 + threadIdx.x] = s;
}
template <int KIND>
static void run(const char* name, uint32_t* d, int waves) {
  const int blocks = 256 * 4 * waves, iters = 100000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(64), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  const double ops = (double)blocks * 64 * iters * 16;
  printf("%-52s %d waves/SIMD: %.3e lane-ops/s = %.2f cycles per wave-instruction\n", name, waves,
         ops / (best * 1e-3), 1024.0 * 2.4e9 * 64 / (ops / (best * 1e-3)));
  fflush(stdout);
}
int main() {
  uint32_t* d;
  (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
  for (int w : {4, 8}) {
    run<0>("and s + bcnt, bcnt src0 and acc in one bank", d, w);
    run<1>("and s + bcnt, different banks", d, w);
    run<2>("bcnt only, one bank", d, w);
    run<3>("bcnt only, different banks", d, w);
    run<4>("v_and_b32 v, s, v only", d, w);
  }
  return 0;
}
