// valu_mix.hip -- does a non-VALU instruction (s_nop, SALU without SCC, s_waitcnt, ds_read) cost a VALU
// issue slot?  16 v_bitop3_b32 per iteration plus 4 of something else, 4 and 8 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_mix.hip -o tools/valu_mix && tools/valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CLOB "v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v2","v3","v40","v41","v42","v43","s40"
#define OP(N) "v_bitop3_b32 v" #N ", v" #N ", v2, v3 bitop3:0x96\n"
#define Q1 OP(16) OP(17) OP(18) OP(19)
#define Q2 OP(20) OP(21) OP(22) OP(23)
#define Q3 OP(24) OP(25) OP(26) OP(27)
#define Q4 OP(28) OP(29) OP(30) OP(31)
template <int KIND>
__global__ __launch_bounds__(64) void k(uint32_t* out, int iters) {
  __shared__ uint32_t lds[1024];
  lds[threadIdx.x] = threadIdx.x;
  __syncthreads();
  asm volatile("v_mov_b32 v2, 0x7654321\n v_mov_b32 v3, 0x1111\n v_lshlrev_b32 v44, 4, %0\n s_mov_b32 s40, 0" :: "v"(threadIdx.x) : CLOB, "v44");
  for (int it = 0; it < iters; ++it) {
    if (KIND == 0) asm volatile(Q1 Q2 Q3 Q4 ::: CLOB);
    if (KIND == 1) asm volatile(Q1 "s_nop 0\n" Q2 "s_nop 0\n" Q3 "s_nop 0\n" Q4 "s_nop 0\n" ::: CLOB);
    if (KIND == 2) asm volatile(Q1 "s_mov_b32 s40, 1\n" Q2 "s_mov_b32 s40, 1\n" Q3 "s_mov_b32 s40, 1\n" Q4 "s_mov_b32 s40, 1\n" ::: CLOB);
    if (KIND == 3) asm volatile(Q1 "s_waitcnt lgkmcnt(0)\n" Q2 "s_waitcnt lgkmcnt(0)\n" Q3 "s_waitcnt lgkmcnt(0)\n" Q4 "s_waitcnt lgkmcnt(0)\n" ::: CLOB);
    if (KIND == 4) asm volatile(Q1 "ds_read_b128 v[40:43], v44\n" Q2 "ds_read_b128 v[40:43], v44\n" Q3 "ds_read_b128 v[40:43], v44\n" Q4 "ds_read_b128 v[40:43], v44\ns_waitcnt lgkmcnt(0)\n" ::: CLOB, "v44");
    if (KIND == 5) asm volatile(Q1 "v_add_u32_dpp v40, v2, v3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n" Q2 "v_add_u32_dpp v41, v2, v3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n" Q3 "v_add_u32_dpp v42, v2, v3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n" Q4 "v_add_u32_dpp v43, v2, v3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n" ::: CLOB);
#define MIX4(I) Q1 I(40) Q2 I(41) Q3 I(42) Q4 I(43)
#define I_MAD16(D) "v_mad_u32_u16 v" #D ", v2, 64, v3 op_sel:[1,0,0,0]\n"
#define I_MAD24(D) "v_mad_u32_u24 v" #D ", v2, 64, v3\n"
#define I_LSHLADD(D) "v_lshl_add_u32 v" #D ", v2, 6, v3\n"
#define I_BFE(D) "v_bfe_u32 v" #D ", v2, 0, 16\n"
#define I_ANDLIT(D) "v_and_b32 v" #D ", 0xffff, v2\n"
#define I_LSHR(D) "v_lshrrev_b32 v" #D ", 16, v2\n"
#define I_SDWA(D) "v_add_u32_sdwa v" #D ", v3, v2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n"
#define I_MOVDPP(D) "v_mov_b32_dpp v" #D ", v2 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
#define I_PERM(D) "v_perm_b32 v" #D ", v2, v3, v2\n"
    if (KIND == 7) asm volatile(MIX4(I_MAD16) ::: CLOB);
    if (KIND == 8) asm volatile(MIX4(I_MAD24) ::: CLOB);
    if (KIND == 9) asm volatile(MIX4(I_LSHLADD) ::: CLOB);
    if (KIND == 10) asm volatile(MIX4(I_BFE) ::: CLOB);
    if (KIND == 11) asm volatile(MIX4(I_ANDLIT) ::: CLOB);
    if (KIND == 12) asm volatile(MIX4(I_LSHR) ::: CLOB);
    if (KIND == 13) asm volatile(MIX4(I_SDWA) ::: CLOB);
    if (KIND == 14) asm volatile(MIX4(I_MOVDPP) ::: CLOB);
    if (KIND == 15) asm volatile(MIX4(I_PERM) ::: CLOB);
#define TAIL4(I) Q1 Q2 Q3 Q4 I(40) I(41) I(42) I(43)
#define I_ADDDPP(D) "v_add_u32_dpp v" #D ", v2, v3 quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n"
    if (KIND == 16) asm volatile(TAIL4(I_ADDDPP) ::: CLOB);        // the 4 odd ops back to back
    if (KIND == 17) asm volatile(TAIL4(I_LSHLADD) ::: CLOB);
    if (KIND == 18) asm volatile(Q1 Q2 I_ADDDPP(40) I_ADDDPP(41) Q3 Q4 I_ADDDPP(42) I_ADDDPP(43) ::: CLOB);
    if (KIND == 6) asm volatile(Q1 "v_add_u32 v40, v2, v3\n" Q2 "v_add_u32 v41, v2, v3\n" Q3 "v_add_u32 v42, v2, v3\n" Q4 "v_add_u32 v43, v2, v3\n" ::: CLOB);
  }
  uint32_t s;
  asm volatile("v_xor_b32 %0, v16, v17\n v_xor_b32 %0, %0, v18\n v_xor_b32 %0, %0, v31\n v_xor_b32 %0, %0, v40" : "=v"(s) :: CLOB);
  out[blockIdx.x * 64 + threadIdx.x] = s + lds[threadIdx.x];
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
  const double iters_total = (double)blocks * iters;   // per SIMD: blocks/1024 waves
  const double cyc = best * 1e-3 * 2.4e9 / (iters_total / 1024.0);
  printf("%-34s %d waves/SIMD: %.1f SIMD cycles per iteration (16 bitop3 alone = baseline)\n", name, waves, cyc);
  fflush(stdout);
}
int main() {
  uint32_t* d;
  (void)hipMalloc(&d, 256 * 4 * 8 * 64 * 4);
  for (int w : {4, 8}) {
    run<0>("16 bitop3", d, w);
    run<1>("16 bitop3 + 4 s_nop", d, w);
    run<2>("16 bitop3 + 4 s_mov_b32", d, w);
    run<3>("16 bitop3 + 4 s_waitcnt", d, w);
    run<4>("16 bitop3 + 4 ds_read_b128", d, w);
    run<5>("16 bitop3 + 4 v_add_u32_dpp", d, w);
    run<6>("16 bitop3 + 4 v_add_u32", d, w);
    run<7>("16 bitop3 + 4 v_mad_u32_u16", d, w);
    run<8>("16 bitop3 + 4 v_mad_u32_u24", d, w);
    run<9>("16 bitop3 + 4 v_lshl_add_u32", d, w);
    run<10>("16 bitop3 + 4 v_bfe_u32", d, w);
    run<11>("16 bitop3 + 4 v_and_b32 literal", d, w);
    run<12>("16 bitop3 + 4 v_lshrrev_b32", d, w);
    run<13>("16 bitop3 + 4 v_add_u32_sdwa", d, w);
    run<14>("16 bitop3 + 4 v_mov_b32_dpp", d, w);
    run<15>("16 bitop3 + 4 v_perm_b32", d, w);
    run<16>("16 bitop3, 4 v_add_u32_dpp in a row", d, w);
    run<17>("16 bitop3, 4 v_lshl_add in a row", d, w);
    run<18>("8 bitop3, 2 dpp, 8 bitop3, 2 dpp", d, w);
  }
  return 0;
}
