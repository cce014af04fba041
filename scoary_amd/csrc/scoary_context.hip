// scoary_context.hip -- handle lifetime, layout queries, kernel timing, the RCCL gather entry
// (include/scoary_hip.h).
#include <dlfcn.h>

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

// ---- per-gene result records for the exchange step --------------------
// rec[t][g] = { tpgp, tpgn, tngp, tngn, p (2 words), odds (2 words), r, nstop }: the ten
// int32 words every rank sends to rank 0 (RCCL gather) -- one fused pass instead of a chain
// of tensor reshapes / copies / a concatenation on the host side.
extern "C++" {
namespace {
__global__ __launch_bounds__(256) void k_pack_records(const int4* __restrict__ counts,
                                                      const double* __restrict__ p,
                                                      const double* __restrict__ odds,
                                                      const uint32_t* __restrict__ r,
                                                      const uint32_t* __restrict__ nstop,
                                                      int64_t M, uint32_t* __restrict__ rec) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= M) return;
  const int4 c = counts[i];
  const uint64_t pb = (uint64_t)__double_as_longlong(p[i]), ob = (uint64_t)__double_as_longlong(odds[i]);
  uint32_t* o = rec + i * 10;
  o[0] = (uint32_t)c.x; o[1] = (uint32_t)c.y; o[2] = (uint32_t)c.z; o[3] = (uint32_t)c.w;
  o[4] = (uint32_t)pb; o[5] = (uint32_t)(pb >> 32);
  o[6] = (uint32_t)ob; o[7] = (uint32_t)(ob >> 32);
  o[8] = r ? r[i] : 0u;
  o[9] = nstop ? nstop[i] : 0u;
}
}  // namespace
}  // extern "C++"

int scoary_pack_records(scoary_handle h, const int32_t* d_counts, const double* d_p,
                        const double* d_odds, const uint32_t* d_r, const uint32_t* d_nstop,
                        int64_t M, uint32_t* d_rec, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_counts || !d_p || !d_odds || !d_rec || M < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_pack_records: bad argument");
  if ((M + 255) / 256 > 0x7fffffffLL) return fail(h, SCOARY_ERR_SIZE, "scoary_pack_records: M too large");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_pack_records");
  hipLaunchKernelGGL(k_pack_records, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s,
                     reinterpret_cast<const int4*>(d_counts), d_p, d_odds, d_r, d_nstop, M, d_rec);
  HIP_TRY(h, hipGetLastError());
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

// ---- e: the path's one exchange step through RCCL, for a host without torch.distributed ----
// No link-time dependency and no rccl.h in the public header: the four entry points are looked up
// in the RCCL instance the CALLER names (the one its communicator came from -- a process can hold
// more than one librccl, PyTorch ships its own), or in the first one the loader finds.
int scoary_gather(scoary_handle h, void* rccl_dl, void* comm, const void* d_send, void* d_recv,
                  int64_t bytes, int rank, int nranks, int root, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!comm || !d_send || bytes < 1 || nranks < 1 || rank < 0 || rank >= nranks || root < 0 ||
      root >= nranks || (rank == root && !d_recv))
    return fail(h, SCOARY_ERR_ARG, "scoary_gather: bad argument");
  void* dl = rccl_dl;
  if (!dl) dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
  if (!dl) dl = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);
  if (!dl) dl = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!dl) {
    const char* why = dlerror();                     // one call: it clears the message it returns
    return fail(h, SCOARY_ERR_DEVICE, std::string("scoary_gather: no librccl: ") + (why ? why : ""));
  }
  using group_fn = int (*)();
  using p2p_fn = int (*)(void*, size_t, int, int, void*, hipStream_t);
  using err_fn = const char* (*)(int);
  auto gstart = reinterpret_cast<group_fn>(dlsym(dl, "ncclGroupStart"));
  auto gend = reinterpret_cast<group_fn>(dlsym(dl, "ncclGroupEnd"));
  auto send = reinterpret_cast<p2p_fn>(dlsym(dl, "ncclSend"));
  auto recv = reinterpret_cast<p2p_fn>(dlsym(dl, "ncclRecv"));
  auto errs = reinterpret_cast<err_fn>(dlsym(dl, "ncclGetErrorString"));
  if (!gstart || !gend || !send || !recv)
    return fail(h, SCOARY_ERR_DEVICE, "scoary_gather: the RCCL library lacks ncclGroupStart / ncclSend / ncclRecv");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr int kNcclInt8 = 0;                       // ncclInt8 / ncclChar of rccl.h
  auto check = [&](int rc, const char* what) -> int {
    if (rc == 0) return SCOARY_OK;
    return fail(h, SCOARY_ERR_HIP, std::string("scoary_gather: ") + what + ": " + (errs ? errs(rc) : "RCCL error"));
  };
  // one group: every rank's send to the root, and on the root one receive per rank (its own block included:
  // a self send / receive pair inside a group is a device copy)
  int rc = check(gstart(), "ncclGroupStart");
  if (rc != SCOARY_OK) return rc;
  rc = check(send(const_cast<void*>(d_send), (size_t)bytes, kNcclInt8, root, comm, s), "ncclSend");
  if (rc == SCOARY_OK && rank == root)
    for (int r = 0; r < nranks && rc == SCOARY_OK; ++r)
      rc = check(recv(static_cast<char*>(d_recv) + (int64_t)r * bytes, (size_t)bytes, kNcclInt8, r, comm, s), "ncclRecv");
  const int rc_end = check(gend(), "ncclGroupEnd");
  return rc != SCOARY_OK ? rc : rc_end;
}

}  // extern "C"
