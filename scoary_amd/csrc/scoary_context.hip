// scoary_context.hip -- handle lifetime, layout queries, kernel timing (include/scoary_hip.h).
#include "scoary_common.hpp"

extern "C" {

int scoary_abi_version(void) { return SCOARY_ABI_VERSION; }

int64_t scoary_tiled_quads(int64_t N) { return tiled_quads(N); }
int64_t scoary_tiled_genes(int64_t G) { return round_up(G < 1 ? 1 : G, kGeneAlign); }
int64_t scoary_tiled_bytes(int64_t G, int64_t N) {
  return 16 * scoary_tiled_quads(N) * scoary_tiled_genes(G);
}
int64_t scoary_row_words(int64_t N) { return 4 * scoary_tiled_quads(N); }

int scoary_create(int device, scoary_handle* out) {
  if (!out) return SCOARY_ERR_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return SCOARY_ERR_DEVICE;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return SCOARY_ERR_DEVICE;
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    std::fprintf(stderr, "scoary_hip: device %d is %s; this library is built for gfx950 only\n",
                 device, prop.gcnArchName);
    return SCOARY_ERR_DEVICE;
  }
  scoary_ctx* h = new scoary_ctx();
  h->device = device;
  h->num_cu = prop.multiProcessorCount;
  *out = h;
  return SCOARY_OK;
}

void scoary_destroy(scoary_handle h) {
  if (!h) return;
  for (auto& t : h->timed) {
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  delete h;
}

const char* scoary_last_error(scoary_handle h) { return h ? h->err.c_str() : "null handle"; }

int scoary_set_timing(scoary_handle h, int enabled) {
  if (!h) return SCOARY_ERR_ARG;
  for (auto& t : h->timed) {
    (void)hipEventDestroy(t.start);
    (void)hipEventDestroy(t.stop);
  }
  h->timed.clear();
  h->timing = enabled != 0;
  return SCOARY_OK;
}

int scoary_last_kernel_ms(scoary_handle h, const char* kernel, double* ms_out) {
  if (!h || !kernel || !ms_out) return SCOARY_ERR_ARG;
  DeviceGuard guard(h->device);
  double total = 0.0;
  int n = 0;
  for (auto& t : h->timed) {
    if (t.name != kernel) continue;
    HIP_TRY(h, hipEventSynchronize(t.stop));
    float ms = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&ms, t.start, t.stop));
    total += ms;
    ++n;
  }
  if (n == 0) return fail(h, SCOARY_ERR_ARG, std::string("no timed launches of ") + kernel);
  *ms_out = total / n;
  return SCOARY_OK;
}

}  // extern "C"
