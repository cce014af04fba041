// mfma_count_probe.hip -- EVIDENCE ONLY, not product (north_star excludes MFMA from the path).
//
// DESIGN.md section 9 prices a matrix-core formulation of the permutation counts,
//     count[g][pi] = sum_i gene[g][i] * label[i][pi]          (0/1 operands, exact in fp32),
// as an MX-fp4 GEMM from the guide's register-resident MFMA rate.  This probe replaces that estimate with a
// number measured on the box: the rate of v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4, scales 1.0) when every
// operand fragment comes out of LDS -- the inner loop any such GEMM would have -- for three register
// blockings (1x1, 2x2, 4x4 tiles of 32x32 per wavefront; A and B fragments re-read from LDS every K step, the
// accumulators stay in registers), one and two wavefronts per SIMD.  fp4 e2m1: 1.0 = 0b0010, 0 = 0b0000;
// scale e8m0 0x7f = 1.0.  The counts are checked exactly: row m of A holds (m % 29) + 3 ones per 32-isolate
// block and B is all ones, so C[m][n] must equal iterations * 2 * ((m % 29) + 3) for every n (also a check of the
// C/D lane mapping: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)).
//
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_count_probe.hip -o tools/mfma_count_probe && tools/mfma_count_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v16f __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// One wavefront: BM x BN tiles of 32x32.  LDS holds, per wavefront, BM A-fragments and BN B-fragments of one
// K step (64 isolates): fragment = 64 lanes x 16 bytes (32 fp4 values: lane l = row/col l % 32, K block l / 32).
template <int BM, int BN>
__global__ __launch_bounds__(256) void k_probe(const v4i* __restrict__ a_frags, const v4i* __restrict__ b_frags,
                                                int iters, float* __restrict__ out) {
  extern __shared__ v4i lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  v4i* my = lds + (size_t)wave * (BM + BN) * 64;
  for (int f = 0; f < BM; ++f) my[f * 64 + lane] = a_frags[f * 64 + lane];
  for (int f = 0; f < BN; ++f) my[(BM + f) * 64 + lane] = b_frags[f * 64 + lane];
  __syncthreads();
  v16f acc[BM][BN];
#pragma unroll
  for (int i = 0; i < BM; ++i)
#pragma unroll
    for (int j = 0; j < BN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  typedef volatile __attribute__((address_space(3))) v4i lds_v4i;
  const uint32_t my_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) v4i*)my + (uint32_t)lane * 16u;
  for (int it = 0; it < iters; ++it) {
    v8i a[BM], b[BN];
#pragma unroll
    for (int i = 0; i < BM; ++i) {
      const v4i x = *(lds_v4i*)(uintptr_t)(my_lds + i * 1024);          // ds_read_b128 every K step
      a[i] = v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0};
    }
#pragma unroll
    for (int j = 0; j < BN; ++j) {
      const v4i x = *(lds_v4i*)(uintptr_t)(my_lds + (BM + j) * 1024);
      b[j] = v8i{x.x, x.y, x.z, x.w, 0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < BM; ++i)
#pragma unroll
      for (int j = 0; j < BN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[i], b[j], acc[i][j], 4, 4, 0, 0x7f7f7f7f, 0,
                                                                    0x7f7f7f7f);
  }
  // C tile (0, 0) of the first wavefront of block 0 goes out for the exactness check; everything is summed so
  // that no accumulator is dead code
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < BM; ++i)
#pragma unroll
    for (int j = 0; j < BN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  if (blockIdx.x == 0 && wave == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) out[r * 64 + lane] = acc[0][0][r];
  }
  if (s == -1.f) out[16 * 64 + (blockIdx.x * nw + wave) % 64] = s;
}

static int ones_of_row(int m) { return (m % 29) + 3; }

template <int BM, int BN>
static int run(const v4i* d_a, const v4i* d_b, float* d_out, int num_cu, int waves_per_simd, int iters) {
  const int threads = 256 * (waves_per_simd > 1 ? 1 : 1);
  const int blocks = num_cu * waves_per_simd;              // 4 wavefronts per block: one per SIMD
  const size_t lds = (size_t)(threads / 64) * (BM + BN) * 64 * sizeof(v4i);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_probe<BM, BN>), dim3(blocks), dim3(threads), lds, 0, d_a, d_b, iters / 8, d_out);   // warm
  CHECK(hipEventRecord(e0));
  hipLaunchKernelGGL((k_probe<BM, BN>), dim3(blocks), dim3(threads), lds, 0, d_a, d_b, iters, d_out);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0.f;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> out(16 * 64);
  CHECK(hipMemcpy(out.data(), d_out, out.size() * sizeof(float), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int r = 0; r < 16; ++r)
    for (int lane = 0; lane < 64; ++lane) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const float want = (float)iters * 2.f * (float)ones_of_row(row);
      bad += out[r * 64 + lane] != want;
    }
  const double mfmas = (double)blocks * (threads / 64) * iters * BM * BN;
  const double macs = mfmas * 32.0 * 32.0 * 64.0;
  const double rate = macs / (ms * 1e-3);
  printf("%dx%d tiles/wavefront, %d wavefront(s)/SIMD: %8.3f ms  %.3e MAC/s = %6.0f TFLOP/s dense-equivalent  "
         "(LDS reads: %.1f KB per MFMA)  counts %s  => cfg3's 1.0e13 MACs in %.2f ms\n",
         BM, BN, waves_per_simd, ms, rate, 2.0 * rate / 1e12, (double)(BM + BN) / (BM * BN),
         bad ? "WRONG" : "exact", 1.0e13 / rate * 1e3);
  return bad;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int num_cu = prop.multiProcessorCount;
  printf("# tools/mfma_count_probe on %s (%d CUs): fp4 x fp4 v_mfma_scale_f32_32x32x64_f8f6f4, A and B fragments read\n"
         "# from LDS every K step, accumulators in registers; evidence for DESIGN.md section 9, not product code\n",
         prop.gcnArchName, num_cu);
  // A fragments: lane l = row l % 32 of the tile, K block l / 32: ones_of_row(row) leading fp4 ones (0x2 nibbles)
  std::vector<uint32_t> a(4 * 64 * 4), b(4 * 64 * 4, 0x22222222u);
  for (int f = 0; f < 4; ++f)
    for (int lane = 0; lane < 64; ++lane) {
      const int n1 = ones_of_row(lane % 32);
      for (int w = 0; w < 4; ++w) {
        uint32_t v = 0;
        for (int nib = 0; nib < 8; ++nib)
          if (w * 8 + nib < n1) v |= 0x2u << (4 * nib);
        a[(f * 64 + lane) * 4 + w] = v;
      }
    }
  v4i *d_a, *d_b;
  float* d_out;
  CHECK(hipMalloc(&d_a, a.size() * 4));
  CHECK(hipMalloc(&d_b, b.size() * 4));
  CHECK(hipMalloc(&d_out, (16 * 64 + 64) * sizeof(float)));
  CHECK(hipMemcpy(d_a, a.data(), a.size() * 4, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d_b, b.data(), b.size() * 4, hipMemcpyHostToDevice));
  const int iters = 1 << 15;            // counts stay below 2^24: exact in fp32
  int bad = 0;
  for (int wps = 1; wps <= 2; ++wps) {
    bad += run<1, 1>(d_a, d_b, d_out, num_cu, wps, iters);
    bad += run<2, 2>(d_a, d_b, d_out, num_cu, wps, iters);
    bad += run<4, 4>(d_a, d_b, d_out, num_cu, wps, iters);
  }
  printf("# list kernel for comparison: cfg3's 5e9 tests (1.0e13 dense MACs) in 4.45-4.70 ms, and on rare-variant data\n"
         "# (cfg4) 2e9 tests x 5000 isolates = 1.0e13 dense MACs in 1.7-1.8 ms\n");
  return bad ? 1 : 0;
}
