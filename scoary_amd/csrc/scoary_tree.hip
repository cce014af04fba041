// scoary_tree.hip -- population-structure stage: pairwise Hamming counts, bit
// gathers, the presence-pattern hash of --collapse and the pairwise-comparison
// tree DP, with their C-ABI.
#include "scoary_common.hpp"

namespace {

// ----------------------------------------------------------------------------
// f-2: pairwise Hamming counts between rows (isolates x variable genes)
// ----------------------------------------------------------------------------
// Same shape as k_counts: lane = row i (coalesced 16 B loads of the tiled
// matrix), blockIdx.y = a group of TB rows j whose words are wave-uniform
// (scalar loads).  out is symmetric, so lane i stores out[j][i]: coalesced.
template <int TB>
__global__ __launch_bounds__(256) void k_hamming(const uint4* __restrict__ tiled,
                                                 const uint32_t* __restrict__ vec, int R, int Rp,
                                                 int Qp, int32_t* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  const int j0 = blockIdx.y * TB;
  const int Wp = Qp * 4;
  uint32_t acc[TB];
  const uint4* rowj[TB];
#pragma unroll
  for (int j = 0; j < TB; ++j) {
    acc[j] = 0;
    rowj[j] = reinterpret_cast<const uint4*>(vec + (int64_t)min(j0 + j, R - 1) * Wp);
  }
  for (int q = 0; q < Qp; ++q) {
    const uint4 gw = tiled[(int64_t)q * Rp + i];
#pragma unroll
    for (int j = 0; j < TB; ++j) {
      const uint4 s = rowj[j][q];
      bcnt_acc(acc[j], gw.x ^ s.x);
      bcnt_acc(acc[j], gw.y ^ s.y);
      bcnt_acc(acc[j], gw.z ^ s.z);
      bcnt_acc(acc[j], gw.w ^ s.w);
    }
  }
  if (i >= R) return;
#pragma unroll
  for (int j = 0; j < TB; ++j)
    if (j0 + j < R) out[(int64_t)(j0 + j) * R + i] = (int32_t)acc[j];
}

// out[r] bit k = rows[r] bit index[k]
__global__ __launch_bounds__(256) void k_gather_bits(const uint32_t* __restrict__ rows, int64_t R,
                                                     int64_t Wsrc, const int32_t* __restrict__ index,
                                                     int64_t K, int64_t Wout,
                                                     uint32_t* __restrict__ out) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t w = blockIdx.y;
  if (r >= R) return;
  const uint32_t* row = rows + r * Wsrc;
  uint32_t word = 0;
  for (int b = 0; b < 32; ++b) {
    const int64_t k = w * 32 + b;
    if (k < K) {
      const int src = index[k];
      word |= ((row[src >> 5] >> (src & 31)) & 1u) << b;
    }
  }
  out[r * Wout + w] = word;
}

// ----------------------------------------------------------------------------
// f-4: presence-pattern hash for --collapse
// ----------------------------------------------------------------------------
// Two independent 64-bit multiply-xorshift chains over the masked words of a
// gene row (lane = gene, mask words wave-uniform).
__device__ __forceinline__ uint64_t mix64(uint64_t h, uint32_t w, uint64_t k) {
  h ^= (uint64_t)w + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
  h *= k;
  return h ^ (h >> 29);
}

__global__ __launch_bounds__(256) void k_row_hash(const uint4* __restrict__ tiled,
                                                  const uint32_t* __restrict__ masks, int G, int Gp,
                                                  int Qp, uint64_t* __restrict__ out) {
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int t = blockIdx.y;
  const uint4* mrow = reinterpret_cast<const uint4*>(masks + (int64_t)t * Qp * 4);
  uint64_t h0 = 0x243F6A8885A308D3ull, h1 = 0x13198A2E03707344ull;
  for (int q = 0; q < Qp; ++q) {
    const uint4 gw = tiled[(int64_t)q * Gp + g];
    const uint4 m = mrow[q];
    const uint32_t w[4] = {gw.x & m.x, gw.y & m.y, gw.z & m.z, gw.w & m.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      h0 = mix64(h0, w[j], 0xBF58476D1CE4E5B9ull);
      h1 = mix64(h1, w[j] ^ 0x5bd1e995u, 0x94D049BB133111EBull);
    }
  }
  if (g < G) {
    out[((int64_t)t * G + g) * 2] = h0;
    out[((int64_t)t * G + g) * 2 + 1] = h1;
  }
}

// ----------------------------------------------------------------------------
// f-1: maximum contrasting pairs on a tree (PhyloTree, scoary/classes.py:199-592)
// ----------------------------------------------------------------------------
// State index 0 = AB, 1 = Ab, 2 = aB, 3 = ab, 4 = "0" (no free path); per state
// (total, supporting, opposing) pairs, or "unreachable".
//
// The reference keeps, per state, the best total and -- among the candidate
// pairings that reach it -- the best supporting and the best opposing count,
// each maximised on its own (classes.py:407-453).  That rule is exactly an
// integer max over two packed keys
//     ks = total << 16 | supporting      ko = total << 16 | opposing
// (counts <= tips/2 < 2^14 -- launch_tree rejects more than 32767 tips -- so adding two
// keys never carries between fields and every reachable key is < 2^30), with
// "unreachable" = -2^30, which therefore stays negative through one addition with a
// reachable key and is re-clamped after every merge.  A pairing of a left and a
// right state is then TWO integer adds, choosing between pairings TWO v_max.
struct TreeNode {
  int ks[5], ko[5];
};
constexpr int kTreeNone = -(1 << 30);

__device__ __forceinline__ void tree_tip(TreeNode& n, int state) {
#pragma unroll
  for (int c = 0; c < 5; ++c) n.ks[c] = n.ko[c] = (c == state) ? 0 : kTreeNone;
}

// Generic node merge (classes.py:268-572).  For a free state c the nine
// candidates of the reference are {L[c]} x {all five R states} and
// {the four L states other than c} x {R[c]}; max distributes over +, so
//   out[c] = max( L[c] + max(R[0..4]),  max(L[x], x != c) + R[c] ).
// "No free path": both closed, or one new pair across the root (+1 total and
// +1 supporting for AB|ab, +1 opposing for Ab|aB).  ~80 VALU ops.
__device__ __forceinline__ void tree_merge(const TreeNode& L, const TreeNode& R, TreeNode& out) {
  int rs = R.ks[0], ro = R.ko[0];
#pragma unroll
  for (int x = 1; x < 5; ++x) {
    rs = max(rs, R.ks[x]);
    ro = max(ro, R.ko[x]);
  }
  // max of L over the states below / above c
  int pre_s[5], pre_o[5], suf_s[5], suf_o[5];
  pre_s[0] = pre_o[0] = kTreeNone;
#pragma unroll
  for (int x = 1; x < 5; ++x) {
    pre_s[x] = max(pre_s[x - 1], L.ks[x - 1]);
    pre_o[x] = max(pre_o[x - 1], L.ko[x - 1]);
  }
  suf_s[4] = suf_o[4] = kTreeNone;
#pragma unroll
  for (int x = 3; x >= 0; --x) {
    suf_s[x] = max(suf_s[x + 1], L.ks[x + 1]);
    suf_o[x] = max(suf_o[x + 1], L.ko[x + 1]);
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int es = max(pre_s[c], suf_s[c]), eo = max(pre_o[c], suf_o[c]);
    out.ks[c] = max(max(L.ks[c] + rs, es + R.ks[c]), kTreeNone);
    out.ko[c] = max(max(L.ko[c] + ro, eo + R.ko[c]), kTreeNone);
  }
  constexpr int kPair = 1 << 16;
  int ns = L.ks[4] + R.ks[4], no = L.ko[4] + R.ko[4];
  ns = max(ns, L.ks[0] + R.ks[3] + (kPair + 1));
  no = max(no, L.ko[0] + R.ko[3] + kPair);
  ns = max(ns, L.ks[3] + R.ks[0] + (kPair + 1));
  no = max(no, L.ko[3] + R.ko[0] + kPair);
  ns = max(ns, L.ks[1] + R.ks[2] + kPair);
  no = max(no, L.ko[1] + R.ko[2] + (kPair + 1));
  ns = max(ns, L.ks[2] + R.ks[1] + kPair);
  no = max(no, L.ko[2] + R.ko[1] + (kPair + 1));
  out.ks[4] = max(ns, kTreeNone);
  out.ko[4] = max(no, kTreeNone);
}

// Merge with a Tip of state s (one-hot node, classes.py:575-592): the candidate
// list collapses to
//   out[c] = L[c]                          for the free states c != s,
//   out[s] = best of ALL five states of L  (the tip supplies the free path),
//   out[4] = L[3 - s] + one new pair       (AB|ab supporting, Ab|aB opposing).
__device__ __forceinline__ void tree_merge_tip(const TreeNode& L, int s, TreeNode& out) {
  int as = L.ks[0], ao = L.ko[0];
#pragma unroll
  for (int x = 1; x < 5; ++x) {
    as = max(as, L.ks[x]);
    ao = max(ao, L.ko[x]);
  }
  const int cs = 3 - s;
  int ps = kTreeNone, po = kTreeNone;
#pragma unroll
  for (int x = 0; x < 4; ++x)
    if (x == cs) {
      ps = L.ks[x];
      po = L.ko[x];
    }
  const bool supporting = (s == 0) || (s == 3);
  constexpr int kPair = 1 << 16;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    out.ks[c] = (c == s) ? as : L.ks[c];
    out.ko[c] = (c == s) ? ao : L.ko[c];
  }
  out.ks[4] = max(ps + kPair + (supporting ? 1 : 0), kTreeNone);
  out.ko[4] = max(po + kPair + (supporting ? 0 : 1), kTreeNone);
}

// One thread per (gene row g, label row l).  The stack program is wave-uniform
// (scalar loads, uniform branches); the top of the stack lives in registers,
// deeper entries in LDS as int32 keys [depth][10][64 lanes].
template <bool EXCEED>
__global__ __launch_bounds__(64) void k_tree_dp(const int32_t* __restrict__ ops, int nops,
                                                const uint32_t* __restrict__ gene_bits,
                                                const uint32_t* __restrict__ label_bits, int64_t G,
                                                int64_t L, int Wt, const int32_t* __restrict__ obs,
                                                int32_t* __restrict__ out3,
                                                uint8_t* __restrict__ exceed) {
  extern __shared__ __attribute__((aligned(16))) int stack_lds[];
  const int lane = threadIdx.x;
  const int64_t id = (int64_t)blockIdx.x * kWave + lane;
  const bool live = id < G * L;
  const int64_t g = live ? id / L : 0, l = live ? id % L : 0;
  const uint32_t* grow = gene_bits + g * Wt;
  const uint32_t* lrow = label_bits + l * Wt;
  TreeNode top;
  tree_tip(top, 4);
  int sp = 0;  // entries below `top`
  int curw = -1;
  uint32_t gw = 0, lw = 0;
  for (int k = 0; k < nops; ++k) {
    const int op = ops[k];
    if (op == -1) {  // merge the two top entries
      TreeNode left;
      --sp;
#pragma unroll
      for (int f = 0; f < 5; ++f) {
        left.ks[f] = stack_lds[((sp * 10) + f) * kWave + lane];
        left.ko[f] = stack_lds[((sp * 10) + 5 + f) * kWave + lane];
      }
      TreeNode m;
      tree_merge(left, top, m);
      top = m;
    } else {
      const int tip = op >= 0 ? op : -2 - op;
      if ((tip >> 5) != curw) {
        curw = tip >> 5;
        gw = grow[curw];
        lw = lrow[curw];
      }
      const int state = (((gw >> (tip & 31)) & 1u) ? 0 : 2) + (((lw >> (tip & 31)) & 1u) ? 0 : 1);
      if (op >= 0) {  // push
        if (k > 0) {
#pragma unroll
          for (int f = 0; f < 5; ++f) {
            stack_lds[((sp * 10) + f) * kWave + lane] = top.ks[f];
            stack_lds[((sp * 10) + 5 + f) * kWave + lane] = top.ko[f];
          }
          ++sp;
        }
        tree_tip(top, state);
      } else {  // merge the top entry with a tip
        TreeNode m;
        tree_merge_tip(top, state, m);
        top = m;
      }
    }
  }
  if (!live) return;
  // three independent maxima over the five states (classes.py:246-249)
  int bt = -1, bp = -1, ba = -1;
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const bool ok = top.ks[c] >= 0;
    bt = max(bt, ok ? (top.ks[c] >> 16) : -1);
    bp = max(bp, ok ? (top.ks[c] & 0xffff) : -1);
    ba = max(ba, ok ? (top.ko[c] & 0xffff) : -1);
  }
  if (!EXCEED) {
    out3[id * 3 + 0] = bt;
    out3[id * 3 + 1] = bp;
    out3[id * 3 + 2] = ba;
  } else {
    const int ot = obs[g * 3], op_ = obs[g * 3 + 1], oa = obs[g * 3 + 2];
    const bool use_pro = op_ >= oa;
    const double est = (double)(use_pro ? op_ : oa) / (double)ot;
    const int x = use_pro ? bp : ba;
    exceed[id] = (bt > 0 && (double)x / (double)bt >= est) ? 1 : 0;
  }
}

// ---------------------------------------------------------------------------
// UPGMA merge loop on the device (scoary/methods.py:640-707, scoary/classes.py:68-196).
// The reference keeps the distances in a quad tree of 2x2 block minima and descends by the
// smallest (value, i, j) of each block; over the whole matrix that is the cell with the
// smallest (value, i-major Morton index of (i, j)).  Two launches per merge:
//   k_upgma_rowmin : one block per live row -> (min value, smallest column attaining it); a row is
//                    rescanned only if the previous merge touched its minimum (else one compare)
//   k_upgma_merge  : one block: pick the row whose (value, Morton(i, j)) is smallest, form
//                    the size-weighted average row with the reference's fp64 operations,
//                    write row/column i, retire row/column j, record (i, j)
// Entries at retired columns are 1.0 in row/column i, as in the reference; a minimum >= 1.0
// (or on the diagonal) is the degenerate case in which the reference merges retired
// clusters -- status is set and the host loop, which mirrors that, takes over.
// ---------------------------------------------------------------------------
constexpr double kUpgmaBig = 9223372036854775807.0;    // float(sys.maxsize), the reference's "infinity"
struct UpgmaState {       // device scratch header
  int32_t status;         // 0, or 1 + step at which the degenerate case showed up
  int32_t prev_i, prev_j; // the previous merge (-1 before the first): which row minima are stale
  int32_t pad;
};
struct UpgmaMin {
  double v;
  int32_t i, j;
};
// a < b in the (value, i-major Morton(i, j)) order
__device__ __forceinline__ bool upgma_less(const UpgmaMin& a, const UpgmaMin& b) {
  if (a.v != b.v) return a.v < b.v;
  const uint32_t x = (uint32_t)(a.i ^ b.i), y = (uint32_t)(a.j ^ b.j);
  const bool j_level_higher = x < y && x < (x ^ y);     // msb(x) < msb(y)
  return j_level_higher ? a.j < b.j : a.i < b.i;
}
__global__ __launch_bounds__(256) void k_upgma_init(const int32_t* __restrict__ counts, int n,
                                                    double ncols, double* __restrict__ D,
                                                    int32_t* __restrict__ alive_list,
                                                    int32_t* __restrict__ pos,
                                                    double* __restrict__ size,
                                                    UpgmaState* __restrict__ st) {
  const int64_t total = (int64_t)n * n;
  for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < total; f += (int64_t)gridDim.x * 256) {
    const int i = (int)(f / n), j = (int)(f % n);
    D[f] = i == j ? 1.0 : (double)counts[f] / ncols;     // pdist 'hamming', diagonal forced to 1
  }
  for (int k = blockIdx.x * 256 + threadIdx.x; k < n; k += gridDim.x * 256) {
    alive_list[k] = k;
    pos[k] = k;
    size[k] = 1.0;
  }
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    st->status = 0;
    st->prev_i = st->prev_j = -1;
  }
}
__global__ __launch_bounds__(256) void k_upgma_rowmin(const double* __restrict__ D, int n,
                                                      const int32_t* __restrict__ alive_list,
                                                      const UpgmaState* __restrict__ st,
                                                      UpgmaMin* __restrict__ rowmin) {
  if (st->status) return;
  const int row = alive_list[blockIdx.x];
  const double* r = D + (int64_t)row * n;
  const int pi = st->prev_i, pj = st->prev_j;
  if (pi >= 0 && row != pi) {                 // cached minimum still valid unless it sat at i or j
    const UpgmaMin c = rowmin[row];
    if (c.j != pi && c.j != pj) {
      if (threadIdx.x == 0) {
        const double v = r[pi];               // the one entry of this row the merge rewrote
        if (v < c.v || (v == c.v && pi < c.j)) rowmin[row] = UpgmaMin{v, row, pi};
      }
      return;
    }
  }
  UpgmaMin best = {kUpgmaBig * 2.0, row, n};
  for (int x = threadIdx.x; x < n; x += 256) {
    const double v = r[x];
    if (v < best.v) {                                    // ascending x per thread: first minimum kept
      best.v = v;
      best.j = x;
    }
  }
  // same row: the order is (value, column)
  __shared__ UpgmaMin sh[256];
  sh[threadIdx.x] = best;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const UpgmaMin o = sh[threadIdx.x + off];
      UpgmaMin& m = sh[threadIdx.x];
      if (o.v < m.v || (o.v == m.v && o.j < m.j)) m = o;
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) rowmin[row] = sh[0];
}
__global__ __launch_bounds__(1024) void k_upgma_merge(double* __restrict__ D, int n, int nalive,
                                                      int step, int32_t* __restrict__ alive_list,
                                                      int32_t* __restrict__ pos,
                                                      double* __restrict__ size,
                                                      double* __restrict__ nd,
                                                      const UpgmaMin* __restrict__ rowmin,
                                                      UpgmaState* __restrict__ st,
                                                      int32_t* __restrict__ merges) {
  if (st->status) return;
  __shared__ UpgmaMin sh[1024];
  UpgmaMin best = {kUpgmaBig * 2.0, 0x7fffffff, 0x7fffffff};
  for (int p = threadIdx.x; p < nalive; p += 1024) {
    const UpgmaMin o = rowmin[alive_list[p]];
    if (upgma_less(o, best)) best = o;
  }
  sh[threadIdx.x] = best;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) {
      const UpgmaMin o = sh[threadIdx.x + off];
      if (upgma_less(o, sh[threadIdx.x])) sh[threadIdx.x] = o;
    }
    __syncthreads();
  }
  const int i = sh[0].i, j = sh[0].j;
  const double vmin = sh[0].v;
  if (!(vmin < 1.0) || i == j || j >= n) {               // degenerate: hand over to the host loop
    if (threadIdx.x == 0) st->status = step + 1;
    return;
  }
  const double si = size[i], sj = size[j], ns = si + sj;
  double* ri = D + (int64_t)i * n;
  double* rj = D + (int64_t)j * n;
  // (d_ik size_i + d_jk size_j) / new_size, each operation rounded as numpy rounds it
  // (the library is built with -ffp-contract=off); 1.0 at retired clusters
  for (int x = threadIdx.x; x < n; x += 1024) {
    const double a = ri[x] * si, b = rj[x] * sj;
    const double sum = a + b;
    const bool live = size[x] != 0.0;
    nd[x] = x == i ? kUpgmaBig : (live ? sum / ns : 1.0);
  }
  __syncthreads();
  for (int x = threadIdx.x; x < n; x += 1024) {
    const double v = nd[x];
    ri[x] = v;
    D[(int64_t)x * n + i] = v;
  }
  __syncthreads();
  for (int x = threadIdx.x; x < n; x += 1024) {
    rj[x] = kUpgmaBig;
    D[(int64_t)x * n + j] = kUpgmaBig;
  }
  if (threadIdx.x == 0) {
    merges[2 * step] = i;
    merges[2 * step + 1] = j;
    size[i] = ns;
    size[j] = 0.0;
    const int pj = pos[j], last = alive_list[nalive - 1];   // drop j from the list of live rows
    alive_list[pj] = last;
    pos[last] = pj;
    st->prev_i = i;
    st->prev_j = j;
  }
}

}  // namespace

extern "C" {

int scoary_hamming(scoary_handle h, const uint32_t* d_tiled, const uint32_t* d_vecrows, int64_t R,
                   int64_t N, int32_t* d_out, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_vecrows || !d_out || R < 1 || N < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_hamming: bad argument");
  if (R > (int64_t)1 << 20) return fail(h, SCOARY_ERR_SIZE, "scoary_hamming: more than 2^20 rows");
  DeviceGuard guard(h->device);
  const int64_t Rp = scoary_tiled_genes(R), Qp = scoary_tiled_quads(N);
  hipStream_t s = static_cast<hipStream_t>(stream);
  constexpr int TB = 8;
  KernelTimer kt(h, s, "k_hamming");
  hipLaunchKernelGGL((k_hamming<TB>), dim3((unsigned)(Rp / 256), (unsigned)((R + TB - 1) / TB)),
                     dim3(256), 0, s, reinterpret_cast<const uint4*>(d_tiled), d_vecrows, (int)R,
                     (int)Rp, (int)Qp, d_out);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}


int64_t scoary_upgma_scratch_bytes(int64_t n) {
  if (n < 1) return 0;
  // D, nd, size (fp64) | partial minima | live-row list, positions | state
  return (n * n + 2 * n) * (int64_t)sizeof(double) + n * (int64_t)sizeof(UpgmaMin) +
         2 * n * (int64_t)sizeof(int32_t) + (int64_t)sizeof(UpgmaState) + 64;
}

int scoary_upgma(scoary_handle h, const int32_t* d_counts, int64_t n, int64_t ncols, void* d_scratch,
                 int32_t* d_merges, int32_t* d_status, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_counts || !d_scratch || !d_merges || !d_status || n < 1 || ncols < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_upgma: bad argument");
  if (n > 46340) return fail(h, SCOARY_ERR_SIZE, "scoary_upgma: more than 46340 isolates");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  char* base = static_cast<char*>(d_scratch);
  double* D = reinterpret_cast<double*>(base);
  double* nd = D + n * n;
  double* size = nd + n;
  UpgmaMin* partial = reinterpret_cast<UpgmaMin*>(size + n);
  int32_t* alive_list = reinterpret_cast<int32_t*>(partial + n);
  int32_t* pos = alive_list + n;
  UpgmaState* st = reinterpret_cast<UpgmaState*>(pos + n + (n & 1));
  KernelTimer kt(h, s, "k_upgma");
  const int64_t init_blocks = (n * n + 255) / 256;
  hipLaunchKernelGGL(k_upgma_init, dim3((unsigned)(init_blocks < 65535 ? init_blocks : 65535)), dim3(256),
                     0, s, d_counts, (int)n, (double)ncols, D, alive_list, pos, size, st);
  for (int64_t step = 0; step + 1 < n; ++step) {
    const int nalive = (int)(n - step);
    hipLaunchKernelGGL(k_upgma_rowmin, dim3((unsigned)nalive), dim3(256), 0, s, D, (int)n, alive_list,
                       st, partial);
    hipLaunchKernelGGL(k_upgma_merge, dim3(1), dim3(1024), 0, s, D, (int)n, nalive, (int)step,
                       alive_list, pos, size, nd, partial, st, d_merges);
  }
  HIP_TRY(h, hipMemcpyAsync(d_status, &st->status, sizeof(int32_t), hipMemcpyDeviceToDevice, s));
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int scoary_gather_bits(scoary_handle h, const uint32_t* d_rows, int64_t R, int64_t Wsrc,
                       const int32_t* d_index, int64_t K, uint32_t* d_out,
                       scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_rows || !d_index || !d_out || R < 1 || Wsrc < 1 || K < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_gather_bits: bad argument");
  const int64_t Wout = (K + 31) / 32;
  if (Wout > 65535) return fail(h, SCOARY_ERR_SIZE, "scoary_gather_bits: K too large");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_gather_bits");
  hipLaunchKernelGGL(k_gather_bits, dim3((unsigned)((R + 255) / 256), (unsigned)Wout), dim3(256), 0,
                     s, d_rows, R, Wsrc, d_index, K, Wout, d_out);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

static int launch_tree(scoary_handle h, const char* what, bool exceed_mode, const int32_t* d_ops,
                       int64_t nops, int64_t stack_depth, const uint32_t* d_gene_bits,
                       const uint32_t* d_label_bits, int64_t G, int64_t L, int64_t K,
                       const int32_t* d_obs, int32_t* d_out3, uint8_t* d_exceed,
                       scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_ops || !d_gene_bits || !d_label_bits || nops < 1 || G < 1 || L < 1 || K < 1 ||
      stack_depth < 1 || (exceed_mode ? (!d_obs || !d_exceed) : !d_out3))
    return fail(h, SCOARY_ERR_ARG, std::string(what) + ": bad argument");
  if (stack_depth > 32) return fail(h, SCOARY_ERR_SIZE, std::string(what) + ": stack_depth > 32");
  // packed keys: total << 16 | count with total <= K/2; K <= 32767 keeps every reachable
  // key below 2^30, so that kTreeNone + reachable stays negative (the "unreachable" test)
  if (K > 32767) return fail(h, SCOARY_ERR_SIZE, std::string(what) + ": more than 32767 tips");
  const int64_t threads = G * L;
  if ((threads + kWave - 1) / kWave > 0x7fffffffLL)
    return fail(h, SCOARY_ERR_SIZE, std::string(what) + ": G*L too large for one launch");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const size_t lds = (size_t)stack_depth * 10 * kWave * sizeof(int);
  const int Wt = (int)((K + 31) / 32);
  dim3 grid((unsigned)((threads + kWave - 1) / kWave));
  KernelTimer kt(h, s, "k_tree_dp");
  if (exceed_mode)
    hipLaunchKernelGGL((k_tree_dp<true>), grid, dim3(kWave), lds, s, d_ops, (int)nops, d_gene_bits,
                       d_label_bits, G, L, Wt, d_obs, (int32_t*)nullptr, d_exceed);
  else
    hipLaunchKernelGGL((k_tree_dp<false>), grid, dim3(kWave), lds, s, d_ops, (int)nops, d_gene_bits,
                       d_label_bits, G, L, Wt, (const int32_t*)nullptr, d_out3, (uint8_t*)nullptr);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

int scoary_tree_pairs(scoary_handle h, const int32_t* d_ops, int64_t nops, int64_t stack_depth,
                      const uint32_t* d_gene_bits, const uint32_t* d_label_bits, int64_t G,
                      int64_t L, int64_t K, int32_t* d_out, scoary_stream_t stream) {
  return launch_tree(h, "scoary_tree_pairs", false, d_ops, nops, stack_depth, d_gene_bits,
                     d_label_bits, G, L, K, nullptr, d_out, nullptr, stream);
}

int scoary_tree_permute(scoary_handle h, const int32_t* d_ops, int64_t nops, int64_t stack_depth,
                        const uint32_t* d_gene_bits, const uint32_t* d_label_bits, int64_t G,
                        int64_t L, int64_t K, const int32_t* d_obs, uint8_t* d_exceed,
                        scoary_stream_t stream) {
  return launch_tree(h, "scoary_tree_permute", true, d_ops, nops, stack_depth, d_gene_bits,
                     d_label_bits, G, L, K, d_obs, nullptr, d_exceed, stream);
}

int scoary_row_hash(scoary_handle h, const uint32_t* d_tiled, const uint32_t* d_masks, int64_t G,
                    int64_t T, int64_t N, uint64_t* d_out, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiled || !d_masks || !d_out || G < 1 || T < 1 || N < 1)
    return fail(h, SCOARY_ERR_ARG, "scoary_row_hash: bad argument");
  if (T > 65535) return fail(h, SCOARY_ERR_SIZE, "scoary_row_hash: T > 65535");
  DeviceGuard guard(h->device);
  const int64_t Gp = scoary_tiled_genes(G), Qp = scoary_tiled_quads(N);
  hipStream_t s = static_cast<hipStream_t>(stream);
  KernelTimer kt(h, s, "k_row_hash");
  hipLaunchKernelGGL(k_row_hash, dim3((unsigned)(Gp / 256), (unsigned)T), dim3(256), 0, s,
                     reinterpret_cast<const uint4*>(d_tiled), d_masks, (int)G, (int)Gp, (int)Qp,
                     d_out);
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}

}  // extern "C"
