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

// ---- hipGraph capture of a launch sequence (small, launch-bound workloads) ----
struct scoary_graph {
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
};

int scoary_graph_begin(scoary_handle h, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (h->timing) return fail(h, SCOARY_ERR_ARG, "scoary_graph_begin: per-kernel timing is on (events cannot be read back from a replayed graph)");
  DeviceGuard guard(h->device);
  HIP_TRY(h, hipStreamBeginCapture(static_cast<hipStream_t>(stream), hipStreamCaptureModeRelaxed));
  return SCOARY_OK;
}

int scoary_graph_end(scoary_handle h, scoary_stream_t stream, scoary_graph_t* out) {
  if (!h || !out) return SCOARY_ERR_ARG;
  DeviceGuard guard(h->device);
  scoary_graph* g = new scoary_graph();
  hipError_t e = hipStreamEndCapture(static_cast<hipStream_t>(stream), &g->graph);
  if (e == hipSuccess) e = hipGraphInstantiate(&g->exec, g->graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    *out = nullptr;
    return fail(h, SCOARY_ERR_HIP, std::string("scoary_graph_end: ") + hipGetErrorString(e));
  }
  *out = g;
  return SCOARY_OK;
}

int scoary_graph_launch(scoary_handle h, scoary_graph_t g, scoary_stream_t stream) {
  if (!h || !g || !g->exec) return SCOARY_ERR_ARG;
  DeviceGuard guard(h->device);
  HIP_TRY(h, hipGraphLaunch(g->exec, static_cast<hipStream_t>(stream)));
  return SCOARY_OK;
}

void scoary_graph_destroy(scoary_graph_t g) {
  if (!g) return;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
}

}  // extern "C"
