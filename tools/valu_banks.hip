// valu_banks.hip -- issue rate of v_bitop3_b32 with hard-coded physical registers:
// which source operands' VGPR banks (register index mod 4) must differ?
//   hipcc --offload-arch=gfx950 -O3 tools/valu_banks.hip -o tools/valu_banks && tools/valu_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(M) M(16) M(17) M(18) M(19) M(20) M(21) M(22) M(23) M(24) M(25) M(26) M(27) M(28) M(29) M(30) M(31)
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31", \
             "v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12"

template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t* out, int iters) {
  asm volatile("v_mov_b32 v1, 0x1234567\n v_mov_b32 v2, 0x7654321\n v_mov_b32 v3, 0x1111\n v_mov_b32 v5, 0x2222\n"
               "v_mov_b32 v9, 0x3333\n v_mov_b32 v6, 0x77\n v_mov_b32 v7, 0x99" ::: CLOB);
#define INIT(N) asm volatile("v_mov_b32 v" #N ", %0" :: "v"(threadIdx.x * 77u + N) : CLOB);
  REP16(INIT)
  for (int it = 0; it < iters; ++it) {
#define A0(N) "v_bitop3_b32 v" #N ", v1, v2, v3 bitop3:0x96\n"   /* banks 1,2,3 */
#define A1(N) "v_bitop3_b32 v" #N ", v1, v5, v3 bitop3:0x96\n"   /* src0 = src1 bank */
#define A2(N) "v_bitop3_b32 v" #N ", v1, v2, v5 bitop3:0x96\n"   /* src0 = src2 bank */
#define A3(N) "v_bitop3_b32 v" #N ", v2, v1, v5 bitop3:0x96\n"   /* src1 = src2 bank */
#define A4(N) "v_bitop3_b32 v" #N ", v1, v5, v9 bitop3:0x96\n"   /* all one bank */
#define A5(N) "v_add_u32_dpp v" #N ", v1, v2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
#define A6(N) "v_add_u32_dpp v" #N ", v1, v5 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
#define A7(N) "v_add_u32 v" #N ", v1, v2\n"
#define A8(N) "v_bitop3_b32 v" #N ", v" #N ", v2, v3 bitop3:0x96\n"  /* dst = src0 chain, srcs 2,3 */
    if (KIND == 0) asm volatile(REP16(A0) ::: CLOB);
    if (KIND == 1) asm volatile(REP16(A1) ::: CLOB);
    if (KIND == 2) asm volatile(REP16(A2) ::: CLOB);
    if (KIND == 3) asm volatile(REP16(A3) ::: CLOB);
    if (KIND == 4) asm volatile(REP16(A4) ::: CLOB);
    if (KIND == 5) asm volatile(REP16(A5) ::: CLOB);
    if (KIND == 6) asm volatile(REP16(A6) ::: CLOB);
    if (KIND == 7) asm volatile(REP16(A7) ::: CLOB);
    if (KIND == 8) asm volatile(REP16(A8) ::: CLOB);
  }
  uint32_t s;
  asm volatile("v_xor_b32 %0, v16, v17\n v_xor_b32 %0, %0, v18\n v_xor_b32 %0, %0, v31" : "=v"(s) :: CLOB);
  out[blockIdx.x * 64 + threadIdx.x] = s;
}

template <int KIND>
static void run(const char* name, uint32_t* d, int waves_per_simd) {
  const int blocks = 256 * 4 * waves_per_simd, iters = 100000;
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
  printf("%-46s %d waves/SIMD: %.3e lane-ops/s = %.2f cycles per wave-instruction at 2.4 GHz\n", name,
         waves_per_simd, ops / (best * 1e-3), 1024.0 * 2.4e9 * 64 / (ops / (best * 1e-3)));
}

#include <chrono>
#include <cstdlib>
// sustained mode: valu_banks <waves per SIMD> <seconds> -- the conflict-free v_bitop3 stream back to
// back for that long (tools/clock_valu.sh samples the shader clock and socket power meanwhile)
static int sustained(uint32_t* d, int waves, double seconds) {
  const int blocks = 256 * 4 * waves, iters = 100000;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const auto t0 = std::chrono::steady_clock::now();
  double ms_total = 0.0;
  long launches = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) hipLaunchKernelGGL((k<0>), dim3(blocks), dim3(64), 0, 0, d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms_total += ms;
    launches += 10;
  }
  const double ops = (double)blocks * 64 * iters * 16 * launches;
  printf("sustained v_bitop3_b32 (banks 1,2,3), %d waves/SIMD, %.1f s: %.3e lane-ops/s\n", waves,
         ms_total * 1e-3, ops / (ms_total * 1e-3));
  return 0;
}

int main(int argc, char** argv) {
  uint32_t* d;
  (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
  if (argc > 2) return sustained(d, atoi(argv[1]), atof(argv[2]));
  for (int w : {4, 8}) {
    run<0>("bitop3 d, v1, v2, v3  (banks 1,2,3)", d, w);
    run<1>("bitop3 d, v1, v5, v3  (src0,src1 share a bank)", d, w);
    run<2>("bitop3 d, v1, v2, v5  (src0,src2 share a bank)", d, w);
    run<3>("bitop3 d, v2, v1, v5  (src1,src2 share a bank)", d, w);
    run<4>("bitop3 d, v1, v5, v9  (all one bank)", d, w);
    run<8>("bitop3 d, d, v2, v3   (in-place chain)", d, w);
    run<7>("add    d, v1, v2", d, w);
    run<5>("add_dpp d, v1, v2 quad_perm", d, w);
    run<6>("add_dpp d, v1, v5 quad_perm (same bank)", d, w);
  }
  return 0;
}
