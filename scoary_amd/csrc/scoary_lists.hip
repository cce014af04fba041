// scoary_lists.hip -- the list-driven permutation path: label tiles, minority-list
// kernels and their C-ABI (scoary_perm_generate_tiles, scoary_permute_lists).
#include "scoary_common.hpp"

namespace {

// ----------------------------------------------------------------------------
// a7 (list-driven variant): permutation exceedance counts from minority lists
// ----------------------------------------------------------------------------
// The dense kernel (k_permute_reg) pays 2 VALU ops per 32 isolates whatever the
// gene looks like.  Here the roles are swapped: a gene is the list of isolates
// carrying its MINORITY value (scoary_lists_build), the permuted labels are
// stored isolate-major in tiles of TW*32 permutations that live in LDS, and a
// gene's overlap count with 32 permutations at once is a bit-sliced
// ("vertical") counter: every listed isolate adds one LDS row word into KC
// counter planes through v_bitop3 full adders (sum = a^b^c, carry = maj(a,b,c)).
// A lane handles 128 permutations (one ds_read_b128 per listed isolate), so the
// cost is ~2.2 VALU ops per listed isolate per 32 permutations instead of 2 ops
// per 32 isolates per permutation: a gene present in 26 % of 2000 isolates
// costs 41 ops per test instead of 137, a rare variant ~20x less.
//
// list_tw / list_lpg / list_nw / list_tile_dwords: scoary_common.hpp

// VGPR banks.  A VOP3 instruction whose src0 and src1 live in the same VGPR bank
// (register index mod 4) issues in ~4.5 cycles instead of ~2.65 on gfx950
// (tools/valu_banks.hip); src2 is free.  xor3 and majority are symmetric in their
// operands, so the ASSEMBLER picks an operand order with src0 and src1 in
// different banks: the register names are only known after allocation, and
// `scoary_bank_vN` (defined once per kernel by SCOARY_BANK_DEFS) maps a name to
// its bank inside an .if.  39 % of the full-adder ops had the conflict before.
__device__ __forceinline__ void scoary_bank_defs() {
  asm volatile(".ifndef scoary_bank_defs\n"
               ".set scoary_bank_defs, 1\n"
#include "scoary_vgpr_banks.inc"
               ".endif");
}
#define SCOARY_SYM3(LUT)                                                        \
  ".if scoary_bank_%1 != scoary_bank_%2\n v_bitop3_b32 %0, %1, %2, %3 bitop3:" LUT \
  "\n.elseif scoary_bank_%1 != scoary_bank_%3\n v_bitop3_b32 %0, %1, %3, %2 bitop3:" LUT \
  "\n.else\n v_bitop3_b32 %0, %2, %3, %1 bitop3:" LUT "\n.endif"
__device__ __forceinline__ uint32_t bit_xor3(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm(SCOARY_SYM3("0x96") : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
__device__ __forceinline__ uint32_t bit_maj(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm(SCOARY_SYM3("0xe8") : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// majority(~a, b, c): the borrow of a - b - c
__device__ __forceinline__ uint32_t bit_majn(uint32_t a, uint32_t b, uint32_t c) {
  uint32_t d;
  asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0x8e" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
// Bit-sliced compare of the KC counter planes against a per-lane constant: bit j of the
// result = (count_j < thr), the borrow of count - thr -- one LUT op per plane.  thr may use
// KC+1 bits (the counts themselves stay below 2^KC: plane KC of the count is zero).
template <int KC>
__device__ __forceinline__ uint32_t count_lt(const uint32_t (&c)[16], uint32_t thr) {
  uint32_t borrow = 0u;
#pragma unroll
  for (int k = 0; k < KC; ++k)
    borrow = bit_majn(c[k], (uint32_t)__builtin_amdgcn_sbfe((int)thr, k, 1), borrow);   // 0 / ~0 masks
  return borrow | (uint32_t)__builtin_amdgcn_sbfe((int)thr, KC, 1);
}
// Rejection-region test (spec S5) in terms of the LIST count u: the acceptance set is an
// interval [lo, hi1) of u (k_lists_crit), so a permutation is in the region iff
// u < lo or u >= hi1 -- two compare chains, 2 ops per plane (the modular form
// ((u - base) mod M) >= span this replaces took three).
template <int KC>
__device__ __forceinline__ uint32_t region_bits(const uint32_t (&c)[16], uint32_t lo, uint32_t hi1) {
  return count_lt<KC>(c, lo) | ~count_lt<KC>(c, hi1);
}
// Counter plane K of permutation word W, pinned to one physical VGPR
// (scoary_ctr_regs.inc) whose bank is never the bank of a row word W: the full
// adders that consume LDS rows (half of all) then never have src0 and src1 in
// one bank, and for the adders on carries the assembler still picks the order.
#define SCOARY_SYM3_INPLACE(LUT)                                                      \
  ".if scoary_bank_%0 != scoary_bank_%1\n v_bitop3_b32 %0, %0, %1, %2 bitop3:" LUT    \
  "\n.elseif scoary_bank_%0 != scoary_bank_%2\n v_bitop3_b32 %0, %0, %2, %1 bitop3:" LUT \
  "\n.else\n v_bitop3_b32 %0, %1, %2, %0 bitop3:" LUT "\n.endif"
// planes without a pinned register (plane 13 of the two-word kernel): allocator's choice
template <int W, int K>
struct Ctr {
  static __device__ __forceinline__ uint32_t maj(uint32_t c, uint32_t x, uint32_t y) {
    uint32_t d;
    asm(SCOARY_SYM3("0xe8") : "=v"(d) : "v"(c), "v"(x), "v"(y));
    return d;
  }
  static __device__ __forceinline__ void xor3(uint32_t& c, uint32_t x, uint32_t y) {
    asm(SCOARY_SYM3_INPLACE("0x96") : "+v"(c) : "v"(x), "v"(y));
  }
  static __device__ __forceinline__ void xor2(uint32_t& c, uint32_t x) { c ^= x; }
};
#define SCOARY_CTR(W, K, R)                                                            \
  template <>                                                                          \
  struct Ctr<W, K> {                                                                   \
    static __device__ __forceinline__ uint32_t maj(uint32_t c, uint32_t x, uint32_t y) { \
      uint32_t d;                                                                      \
      asm(SCOARY_SYM3("0xe8") : "=v"(d) : "{" R "}"(c), "v"(x), "v"(y));               \
      return d;                                                                        \
    }                                                                                  \
    static __device__ __forceinline__ void xor3(uint32_t& c, uint32_t x, uint32_t y) { \
      asm(SCOARY_SYM3_INPLACE("0x96") : "+{" R "}"(c) : "v"(x), "v"(y));               \
    }                                                                                  \
    static __device__ __forceinline__ void xor2(uint32_t& c, uint32_t x) {             \
      asm("v_xor_b32 %0, %1, %0" : "+{" R "}"(c) : "v"(x));                            \
    }                                                                                  \
  };
#include "scoary_ctr_regs.inc"
#undef SCOARY_CTR
// c += x + y at bit-plane weight 1: returns the carry (weight 2)
template <int W, int K>
__device__ __forceinline__ uint32_t full_add(uint32_t& c, uint32_t x, uint32_t y) {
  const uint32_t carry = Ctr<W, K>::maj(c, x, y);
  Ctr<W, K>::xor3(c, x, y);
  return carry;
}

// Isolate-major label tiles: tiles[t][tile][row 0..N][TW] dwords, row N all
// zero; dword j of a row = labels of permutations tile*TW*32 + 32j .. +31
// (written by k_labels, scoary_labels.hip).

// Per (trait, list slot): the ACCEPTANCE interval of the two-sided test in terms of the
// LIST count u (u = a for a ones-list, npos - a for a zeros-list), as [lo, hi1):
//   crit = (base, span): accept a in [base, base + span)          (k_fisher; span 0 = reject all)
//   ones-list : u in [base, base + span)
//   zeros-list: u in [npos - base - span + 1, npos - base + 1)
// 0 <= lo <= hi1 <= npos + 1 <= N + 1 < 2^(KC+1).  span 0 gives lo = hi1 = 0: u >= 0 always,
// every permutation is in the region (r = P, spec S5).
__global__ __launch_bounds__(256) void k_lists_crit(const uint2* __restrict__ crit,
                                                    const int32_t* __restrict__ margins,
                                                    const int32_t* __restrict__ order,
                                                    const uint8_t* __restrict__ flipped, int G,
                                                    uint2* __restrict__ out) {
  const int k = blockIdx.x * 256 + threadIdx.x;
  const int t = blockIdx.y;
  if (k >= G) return;
  const int g = order[k];
  const uint2 c = crit[(int64_t)t * G + g];
  uint2 o;
  if (c.y == 0u) {
    o = make_uint2(0u, 0u);
  } else if (!flipped[g]) {
    o = make_uint2(c.x, c.x + c.y);
  } else {
    const uint32_t npos = (uint32_t)margins[2 * t];
    o = make_uint2(npos - c.x - c.y + 1u, npos - c.x + 1u);
  }
  out[(int64_t)t * G + k] = o;
}

// lane id from v_mbcnt, opaque to the optimiser (see its use in k_permute_lists)
__device__ __forceinline__ int fresh_lane() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// 128 permutations per lane: LPG = TW/4 lanes per gene read the tile rows with
// ds_read_b128, 64/LPG genes per wavefront; one address add serves four words.
// ds_read_b128 is served in the lane groups {0-3,12-15,20-27}, ... over the full
// 256-byte bank row: with slot k starting at residue class k mod (64/TW)
// (scoary_lists_build) the genes of a group sit on distinct 4*TW-byte slots.
struct Rows4 { uint32_t w0[4], w1[4], w2[4], w3[4]; };   // 4 tile rows x 4 permutation words
// The 4 entries held by lane H of every LPG-lane gene group -> 4 ds_read_b128.
// The four address adds (entry of lane H, DPP quad_perm broadcast, + column) are issued
// back to back from one asm block: a DPP add between v_bitop3 ops costs ~10 cycles, four in
// a row ~10 together (tools/valu_mix.hip -- the price is the switch, not the op).
#define SCOARY_DPP4(QP)                                                                  \
  asm("v_add_u32_dpp %0, %4, %8 " QP " row_mask:0xf bank_mask:0xf\n"                    \
      "v_add_u32_dpp %1, %5, %8 " QP " row_mask:0xf bank_mask:0xf\n"                    \
      "v_add_u32_dpp %2, %6, %8 " QP " row_mask:0xf bank_mask:0xf\n"                    \
      "v_add_u32_dpp %3, %7, %8 " QP " row_mask:0xf bank_mask:0xf"                       \
      : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])                               \
      : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[3]), "v"(colb))
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
template <int LPG, int H, int NW>
__device__ __forceinline__ void read4x4(Rows4& x, const uint32_t (&e)[4], uint32_t colb) {
  uint32_t a[4];
  if constexpr (LPG == 1) {
    a[0] = e[0], a[1] = e[1], a[2] = e[2], a[3] = e[3];
  } else if constexpr (LPG == 4) {             // lane H of the quad
    if constexpr (H == 0) SCOARY_DPP4("quad_perm:[0,0,0,0]");
    if constexpr (H == 1) SCOARY_DPP4("quad_perm:[1,1,1,1]");
    if constexpr (H == 2) SCOARY_DPP4("quad_perm:[2,2,2,2]");
    if constexpr (H == 3) SCOARY_DPP4("quad_perm:[3,3,3,3]");
  } else {                                     // lane H of each pair
    if constexpr (H == 0) SCOARY_DPP4("quad_perm:[0,0,2,2]");
    if constexpr (H == 1) SCOARY_DPP4("quad_perm:[1,1,3,3]");
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    // colb carries the tile's LDS address: a[j] is the absolute LDS address of the row piece
    if constexpr (NW == 4) {
      const u32x4 v = *(const __attribute__((address_space(3))) u32x4*)(uintptr_t)a[j];
      x.w0[j] = v.x;
      x.w1[j] = v.y;
      x.w2[j] = v.z;
      x.w3[j] = v.w;
    } else if constexpr (NW == 2) {            // two words per lane: ds_read_b64
      const u32x2 v = *(const __attribute__((address_space(3))) u32x2*)(uintptr_t)a[j];
      x.w0[j] = v.x;
      x.w1[j] = v.y;
    } else {                                   // one word per lane: ds_read_b32
      x.w0[j] = *(const __attribute__((address_space(3))) uint32_t*)(uintptr_t)a[j];
    }
  }
}
#undef SCOARY_DPP4
// k_permute_seglists: 16-bit entries (row index inside the isolate segment, <= 20352), eight per
// 16-byte index vector; sub-step H takes dwords 2H, 2H + 1.  LDS address = index * 8 (two-dword
// rows, the tile segment sits at LDS address 0): one SDWA shift per entry picks the half word and
// scales it -- the price of halving the index stream (round 3: the kernel ran at 0.25 of the VALU
// peak behind 4.5 TB/s of index fetches, profiles/r03_wide50000_pmc.json).
template <int H>
__device__ __forceinline__ void read4x4_half(Rows4& x, const uint32_t (&e)[4], uint32_t three) {
  uint32_t a[4];
  asm("v_lshlrev_b32_sdwa %0, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
      "v_lshlrev_b32_sdwa %1, %4, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
      "v_lshlrev_b32_sdwa %2, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
      "v_lshlrev_b32_sdwa %3, %4, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1"
      : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
      : "s"(three), "v"(e[2 * H]), "v"(e[2 * H + 1]));
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const u32x2 v = *(const __attribute__((address_space(3))) u32x2*)(uintptr_t)a[j];
    x.w0[j] = v.x;
    x.w1[j] = v.y;
  }
}
// 4 row words -> counter planes 0..1 of word W, returns the carry of weight 4
template <int W>
__device__ __forceinline__ uint32_t sum4(uint32_t (&c)[16], const uint32_t (&x)[4]) {
  const uint32_t a1 = full_add<W, 0>(c[0], x[0], x[1]);
  const uint32_t a2 = full_add<W, 0>(c[0], x[2], x[3]);
  return full_add<W, 1>(c[1], a1, a2);
}
struct Carry4 { uint32_t w[4]; };

// LPG lanes per gene (4, 2, 1 for 16-, 8-, 4-dword tile rows), 64/LPG genes per
// wavefront, NW permutation words per lane (4; 2 / 1 for the 2- / 1-dword rows of
// N > 10239 / 20479, one lane per gene).  Lists are walked in sub-steps of 4 entries: lane j of a gene
// group holds entries 4j..4j+3 of each 4*LPG-entry piece.
// Cache policy of the tile's LDS-DMA loads (A/B builds: -DSCOARY_TILE_LOAD_POLICY='" nt"').  A tile is
// read once per block and never again by that CU; the index lists next to it in L2 are re-read by
// every block of the chunk.
#ifndef SCOARY_TILE_LOAD_POLICY
#define SCOARY_TILE_LOAD_POLICY ""
#endif
template <int LPG, int NW, int KC>
__global__ __launch_bounds__(1024) void k_permute_lists(const uint32_t* __restrict__ tiles,
                                                        const uint32_t* __restrict__ lidx,
                                                        const int32_t* __restrict__ lstart,
                                                        const int32_t* __restrict__ lngroups,
                                                        const uint2* __restrict__ lcrit, int G,
                                                        int N, int64_t P, int ntiles,
                                                        int groups_per_block, int64_t lidx_bytes,
                                                        uint16_t* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) uint32_t tile_lds[];
  scoary_bank_defs();                // assembler symbols for the operand-ordering .if blocks
  constexpr int TW = NW * LPG;       // tile row, dwords
  static_assert(NW == 4 || ((NW == 2 || NW == 1) && LPG == 1), "narrow rows: one lane per gene");
  constexpr int GPW = kWave / LPG;   // genes per wavefront
  // A block = one label tile (trait, 32*TW permutations) in LDS x one chunk of wave groups.
  // blockIdx.x (trait, tile) runs fastest, so the ~num_cu blocks in flight walk the SAME
  // chunk of index lists against different tiles: a list byte comes from HBM once and
  // from L2 for everyone else.  (Round 2 built the alternative -- a block walking several
  // tiles and keeping its counts in a register, one store per (gene, trait): the lists are
  // then re-streamed once per tile by blocks that no longer share them, -4 % at the
  // headline config and -15 % on rare variants; profiles/r02_ab_tileloop.txt.)
  const int t = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: loops over groups stay scalar
  const int col = lane % LPG;
  const int ngroups = (G + GPW - 1) / GPW;         // wave groups of GPW genes
  const int q_lo = blockIdx.y * groups_per_block;
  const int q_hi = min(ngroups, q_lo + groups_per_block);
  const uint32_t lane_off = (uint32_t)lane * 16u;   // 16-byte vectors: tile loads and index loads
  // per-(trait, tile) exceedance counts of every list slot: plain 16-bit stores, summed
  // over the tiles by k_lists_reduce (one device-scope atomicAdd per (gene, tile) cost a
  // 32-byte memory-side write each: 316 MB for a 2 MB result at the headline config)
  uint16_t* out = partial + (int64_t)blockIdx.x * ((int64_t)ngroups * GPW);

  // interleaved lists (piece = TW entries): the wavefront's 64 lanes read 64 consecutive
  // 16-byte index vectors per piece; lane l reads vector piece*64 + l of its group
  struct alignas(16) Ent { uint32_t e[4]; };
#ifndef SCOARY_LIST_LOAD_POLICY
#define SCOARY_LIST_LOAD_POLICY 0
#endif
  constexpr int kListLoadPolicy = SCOARY_LIST_LOAD_POLICY;
  // A wave group's list: its length in half-steps of 16 entries (lists are padded to 16: a last
  // half step costs half a step, where padding to 32 made the average list 3 % longer), its last
  // piece, and a buffer resource on it -- group base in the descriptor (SGPRs), piece offset in
  // the scalar offset, lane offset in one VGPR: no per-load 64-bit VALU address arithmetic.
  // num_records = the bytes from the group's base to the end of the index array: a read past
  // the end (groups without entries still issue their prologue loads) returns 0.
  struct Group { int nhalf, last; __amdgpu_buffer_rsrc_t rsrc; };
  auto open_group = [&](int qq) -> Group {
    const int64_t start = (int64_t)__builtin_amdgcn_readfirstlane(lstart[qq * GPW]);
    const int nh = __builtin_amdgcn_readfirstlane(lngroups[qq * GPW]);
    const int64_t gbytes = lidx_bytes - start * 128;
    return Group{nh, max(nh * (4 / LPG) - 1, 0),
                 __builtin_amdgcn_make_buffer_rsrc(
                     const_cast<Ent*>(reinterpret_cast<const Ent*>(lidx) + start * 8), 0,
                     (int)min(gbytes, (int64_t)0x7fffffff), 0x00020000)};
  };
  auto load_from = [&](const Group& gr, int p) -> Ent {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
        gr.rsrc, lane_off, min(p, gr.last) * (kWave * (int)sizeof(Ent)), kListLoadPolicy);
    return Ent{{v.x, v.y, v.z, v.w}};
  };
  // The first four index vectors of a group are requested one group AHEAD: at the start of the
  // previous group's epilogue (region test, ~0.3 us of VALU work in which the ring registers
  // are dead) -- and for the wavefront's first group here, before the label tile is waited
  // for -- so a group does not open with an exposed L2 round trip.
  int q = q_lo + wave;
  Group cur = open_group(min(q, ngroups - 1));
  Ent ring[4] = {load_from(cur, 0), load_from(cur, 1), load_from(cur, 2), load_from(cur, 3)};
  const uint32_t* src = tiles + (int64_t)blockIdx.x * list_tile_dwords(N, TW);
  {
    // tile -> LDS by LDS-DMA (global_load_lds_dwordx4): a wavefront moves 64 x 16 B
    // per instruction straight into LDS (destination = M0 + 16*lane), no VGPR round trip
    // and no ds_write pass.  Address = wave-uniform base (SGPR pair) + 32-bit lane offset:
    // the saddr form (inline asm: the builtin only takes a per-lane 64-bit address).
    const uint4* src4 = reinterpret_cast<const uint4*>(src);
    const int n4 = (int)(list_tile_dwords(N, TW) / 4);   // HBM tiles are padded to 16 bytes
    const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)tile_lds;
    for (int i = wave * kWave; i < n4; i += nwaves * kWave)
      if (i + lane < n4) {
        uint32_t m0_saved;     // M0 is handed back as it was: nothing else may be assumed about it
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %3" SCOARY_TILE_LOAD_POLICY "\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved)
                     : "s"(lds0 + (uint32_t)i * 16u), "v"(lane_off), "s"(src4 + i)
                     : "memory");
      }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();

  // LPG == 1: an entry is used as the LDS address as it is -- the tile is the kernel's only
  // LDS object and sits at LDS address 0 (checked on the host: no static LDS in this kernel)
  {
  for (; q < q_hi; q += nwaves) {
    const int nhalf = cur.nhalf, last = cur.last;
    const int nsuper = (nhalf + 1) >> 1;                               // 32-entry steps, the last maybe half
    // this lane's column of a tile row, as an absolute LDS address (entries are row byte offsets)
    const uint32_t colb = (uint32_t)col * (NW * 4u) +
                          (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)tile_lds;

    uint32_t c0[16], c1[16], c2[16], c3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c0[k] = c1[k] = c2[k] = c3[k] = 0u;
    constexpr int kVecSub = LPG;       // lane h of a gene's group holds the four entries of sub-step h
#define SCOARY_READ4(H, X, E) read4x4<LPG, H, NW>(X, E, colb)
#include "scoary_list_walk.inc"
#undef SCOARY_READ4
    // next group of this wavefront: open it and request its first index vectors now
    if (q + nwaves < q_hi) {
      cur = open_group(q + nwaves);
      ring[0] = load_from(cur, 0);
      ring[1] = load_from(cur, 1);
      ring[2] = load_from(cur, 2);
      ring[3] = load_from(cur, 3);
    }
    // the lane's gene within the group, recomputed here (volatile asm: not hoisted) rather
    // than held in a register across the list walk -- the walk uses every VGPR there is
    const int lg = fresh_lane() / LPG;
    const int slot = min(q * GPW + lg, G - 1);
    const bool have = q * GPW + lg < G;
    const uint2 cr = lcrit[(int64_t)t * G + slot];
    const uint32_t lo = cr.x, hi1 = cr.y;            // acceptance interval of the list count
    uint32_t valid[4];                                // permutations of this tile that exist
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const int64_t p_first = ((int64_t)tile * TW + NW * col + w) * 32;
      valid[w] = p_first >= P ? 0u : (P - p_first >= 32 ? 0xffffffffu : ((1u << (P - p_first)) - 1u));
    }
    int cnt = 0;
    cnt += __popc(region_bits<KC>(c0, lo, hi1) & valid[0]);
    if constexpr (NW > 1) cnt += __popc(region_bits<KC>(c1, lo, hi1) & valid[1]);
    if constexpr (NW > 2) {
      cnt += __popc(region_bits<KC>(c2, lo, hi1) & valid[2]);
      cnt += __popc(region_bits<KC>(c3, lo, hi1) & valid[3]);
    }
    if (!have) cnt = 0;
#pragma unroll
    for (int off = LPG / 2; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off);
    if (col == 0) out[(int64_t)q * GPW + lg] = (uint16_t)cnt;      // cnt <= 32*TW <= 512
  }
  }
}

// N > 20479: one 64-permutation tile no longer fits the 160 KB of LDS, so the isolates are cut
// into nseg segments of kSegRows rows (scoary_common.hpp).  A gene's index list is one sub-list
// per segment (16-bit entries = row indices inside the segment, two per dword -- round 4, the
// 32-bit LDS addresses of round 3 made the kernel index-bandwidth-bound; lstart / lngroups are [nseg][G]);
// the block walks its wave groups in ROUNDS of one group per wavefront and, inside a round,
// loads the tile segment by segment: the counter planes stay in registers across the reloads
// and the region test comes after the last segment.  One lane per gene, two permutation words
// per lane (the TW = 2 geometry), 16 counter planes (N / 2 < 2^16).  The walk itself is the
// code of k_permute_lists (scoary_list_walk.inc).
template <int KC>
__global__ __launch_bounds__(1024) void k_permute_seglists(const uint32_t* __restrict__ tiles,
                                                           const uint32_t* __restrict__ lidx,
                                                           const int32_t* __restrict__ lstart,
                                                           const int32_t* __restrict__ lngroups,
                                                           const uint2* __restrict__ lcrit, int G,
                                                           int N, int64_t P, int ntiles,
                                                           int groups_per_block, int64_t lidx_bytes,
                                                           int nseg, uint16_t* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) uint32_t tile_lds[];
  scoary_bank_defs();
  constexpr int LPG = 1, NW = kSegTW, TW = kSegTW, GPW = kWave;
  static_assert(kSegTW == 2, "two permutation words per lane");
  // Grid as in k_permute_lists: (trait, tile) fastest, chunk of wave groups in y -- the dispatcher
  // hands the next block to whichever CU is free.  (Round 4 also tried the XCD-aware map of
  // tools/xcd_map.patch here, every XCD one chunk of each row of eight: FETCH 4.2 -> 0.8 GB per
  // launch at 20 000 x 50 000 and no gain in time -- with 16-bit entries the kernel is no longer
  // fetch-bound -- and a 3x LOSS whenever there are fewer than eight chunks, XCDs standing idle:
  // 2432 genes x 40 959 isolates x 4 traits, P = 8192, 2.7 -> 9.1 ms.  Dropped.)
  const int ngroups = (G + GPW - 1) / GPW;
  const int t = blockIdx.x / ntiles, tile = blockIdx.x % ntiles;
  const int tid = threadIdx.x, lane = tid & 63, nwaves = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q_lo = blockIdx.y * groups_per_block;
  const int q_hi = min(ngroups, q_lo + groups_per_block);
  const uint32_t lane_off = (uint32_t)lane * 16u;
  uint16_t* out = partial + (int64_t)blockIdx.x * ((int64_t)ngroups * GPW);
  struct alignas(16) Ent { uint32_t e[4]; };
  constexpr int kListLoadPolicy = SCOARY_LIST_LOAD_POLICY;
  struct Group { int nhalf, last; __amdgpu_buffer_rsrc_t rsrc; };
  auto open_group = [&](int qq, int sgm) -> Group {
    const int64_t slot = (int64_t)sgm * G + (int64_t)qq * GPW;
    const int64_t start = (int64_t)__builtin_amdgcn_readfirstlane(lstart[slot]);
    const int nh = __builtin_amdgcn_readfirstlane(lngroups[slot]);
    const int64_t gbytes = lidx_bytes - start * 128;
    // a half step (16 entries of 16 bits) is two 16-byte vectors per lane
    return Group{nh, max(nh * 2 - 1, 0),
                 __builtin_amdgcn_make_buffer_rsrc(
                     const_cast<Ent*>(reinterpret_cast<const Ent*>(lidx) + start * 8), 0,
                     (int)min(gbytes, (int64_t)0x7fffffff), 0x00020000)};
  };
  auto load_from = [&](const Group& gr, int p) -> Ent {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
        gr.rsrc, lane_off, min(p, gr.last) * (kWave * (int)sizeof(Ent)), kListLoadPolicy);
    return Ent{{v.x, v.y, v.z, v.w}};
  };
  const uint32_t lds0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) uint32_t*)tile_lds;
  const uint32_t three = 3u;                             // SDWA shift amount (an SGPR operand)
  const uint32_t* src = tiles + (int64_t)blockIdx.x * ((int64_t)nseg * kSegStride);
  // rounds and segments are block-uniform: every wavefront meets every barrier, whether or not
  // it has a wave group in this round
  const int rounds = (q_hi - q_lo + nwaves - 1) / nwaves;
  for (int rnd = 0; rnd < rounds; ++rnd) {
    const int q = q_lo + rnd * nwaves + wave;
    const bool active = q < q_hi;                                        // wave-uniform
    uint32_t c0[16], c1[16], c2[16], c3[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) c0[k] = c1[k] = c2[k] = c3[k] = 0u;
    for (int sgm = 0; sgm < nseg; ++sgm) {
      // the sub-list's first four index vectors are requested before the tile segment is
      // (re)loaded, so the walk does not open with an exposed L2 round trip (inactive
      // wavefronts read the block's last group: harmless)
      const Group cur = open_group(min(q, q_hi - 1), sgm);
      Ent ring[4] = {load_from(cur, 0), load_from(cur, 1), load_from(cur, 2), load_from(cur, 3)};
      __syncthreads();                       // the previous segment has been walked by everyone
      {
        const uint4* src4 = reinterpret_cast<const uint4*>(src + (int64_t)sgm * kSegStride);
        constexpr int n4 = kSegStride / 4;
        for (int i = wave * kWave; i < n4; i += nwaves * kWave)
          if (i + lane < n4) {
            uint32_t m0_saved;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved)
                         : "s"(lds0 + (uint32_t)i * 16u), "v"(lane_off), "s"(src4 + i)
                         : "memory");
          }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (active) {
        const int nhalf = cur.nhalf;
        const int nsuper = (nhalf + 1) >> 1;
        constexpr int kVecSub = 2;                     // eight 16-bit entries per index vector
#define SCOARY_READ4(H, X, E) read4x4_half<H>(X, E, three)
#include "scoary_list_walk.inc"
#undef SCOARY_READ4
      }
    }
    if (active) {
      const int lg = fresh_lane();
      const int slot = min(q * GPW + lg, G - 1);
      const bool have = q * GPW + lg < G;
      const uint2 cr = lcrit[(int64_t)t * G + slot];
      uint32_t valid[NW];                              // permutations of this tile that exist
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        const int64_t p_first = ((int64_t)tile * TW + w) * 32;
        valid[w] = p_first >= P ? 0u : (P - p_first >= 32 ? 0xffffffffu : ((1u << (P - p_first)) - 1u));
      }
      int cnt = __popc(region_bits<KC>(c0, cr.x, cr.y) & valid[0]) +
                __popc(region_bits<KC>(c1, cr.x, cr.y) & valid[1]);
      if (!have) cnt = 0;
      out[(int64_t)q * GPW + lg] = (uint16_t)cnt;                  // cnt <= 64
    }
  }
}

// r[t][gene of slot k] (+)= sum over the tiles of partial[t][tile][k]
// Block = 64 list slots x 4 tile quarters (round 6): a thread per slot walking all the tiles alone left a 25 000-gene
// shard with 1.5 wavefronts per CU and one 2-byte load in flight each -- 13-22 us for 2-16 MB; four threads per slot
// take every fourth tile, their sums meet in LDS.
__global__ __launch_bounds__(256) void k_lists_reduce(const uint16_t* __restrict__ partial,
                                                      int ntiles, int64_t gs, int G,
                                                      const int32_t* __restrict__ order,
                                                      int accumulate, uint32_t* __restrict__ r) {
  __shared__ uint32_t s_part[4][kWave];
  const int lane = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int k = blockIdx.x * kWave + lane;
  const int t = blockIdx.y;
  uint32_t sum = 0u;
  if (k < G) {
    const uint16_t* in = partial + (int64_t)t * ntiles * gs + k;
#pragma unroll 4
    for (int tile = part; tile < ntiles; tile += 4) sum += in[(int64_t)tile * gs];
  }
  s_part[part][lane] = sum;
  __syncthreads();
  if (part != 0 || k >= G) return;
  sum = s_part[0][lane] + s_part[1][lane] + s_part[2][lane] + s_part[3][lane];
  uint32_t* dst = r + (int64_t)t * G + order[k];
  *dst = accumulate ? *dst + sum : sum;
}

}  // namespace

extern "C" {

int64_t scoary_list_tiles_words(int64_t N, int64_t P, int64_t T) {
  const int TW = list_tw(N);
  if (!TW) return 0;
  const int64_t tile_perms = TW * 32;
  const int64_t ntiles = (P + tile_perms - 1) / tile_perms;
  return T * ntiles * list_tile_dwords_seg(N, TW);
}
int64_t scoary_list_tile_words(int64_t N) { return list_tw(N) ? list_tile_dwords_seg(N, list_tw(N)) : 0; }
int64_t scoary_list_max_isolates(void) { return kMaxListIsolates; }
int64_t scoary_list_segments(int64_t N) { return N < 1 ? 0 : list_segments(N); }
int scoary_list_params(int64_t N, int64_t* out5) {
  if (!out5) return SCOARY_ERR_ARG;
  const int TW = list_tw(N);
  out5[0] = TW;                     /* tile row width in dwords (0: N too large for LDS tiles) */
  out5[1] = TW * 4;                 /* LDS / tile row stride in bytes */
  out5[2] = TW ? kWave / list_lpg(TW) : 0;   /* genes per wavefront: lists padded to equal length */
  out5[3] = TW ? 64 / TW : 0;       /* residue classes of the isolate index (256-byte bank row) */
  out5[4] = TW ? (list_segments(N) > 1 ? kSegPiece : 4 * list_lpg(TW)) : 0;   /* interleave piece, entries (16-bit ones in segments) */
  return TW ? SCOARY_OK : SCOARY_ERR_SIZE;
}

extern "C++" {
// Bytes of index lists one k_permute_lists block walks against its LDS tile.  SCOARY_LIST_CHUNK_MB
// (1..64) overrides it for same-box A/B runs (tools/ab_chunk.sh); read once per process.
static int64_t list_chunk_bytes(int TW) {
  static const long env_mb = [] {
    const char* e = std::getenv("SCOARY_LIST_CHUNK_MB");
    const long mb = e ? std::strtol(e, nullptr, 10) : 0;
    return (mb >= 1 && mb <= 64) ? mb : 0L;
  }();
  // One lane per gene (TW <= 4, N > 5119): sixteen wave groups of 64 genes are ~10 MB of lists at
  // N = 10 000 whatever is asked for here, more than an L2 holds -- the lists of such a launch
  // stream from the Infinity Cache / HBM once per wave of concurrent blocks either way, and larger
  // chunks re-load the 160 KB tile less often: 16 MB is -1.7 % of kernel time on a cfg5-shaped
  // shard (profiles/r05_ab_cfg5_traffic.txt; round 3 measured -1.2 %).
  return (int64_t)(env_mb ? env_mb : (TW <= 4 ? 16 : kListChunkMB)) << 20;
}
// Launch geometry of k_permute_lists (also sizes the scratch)
struct ListGeom {
  int64_t ntiles, ngroups, gs, gpb, chunks;
};
static ListGeom list_geom(int num_cu, int64_t G, int64_t T, int64_t N, int64_t P, int64_t entries) {
  ListGeom g{};
  const int TW = list_tw(N), GPW = kWave / list_lpg(TW);
  const int64_t tile_perms = TW * 32;
  g.ntiles = (P + tile_perms - 1) / tile_perms;
  g.ngroups = (G + GPW - 1) / GPW;
  g.gs = g.ngroups * GPW;
  // enough blocks for >= 16 rounds over the CUs, and gene chunks whose index lists
  // (~2 MB) stay in an XCD's 4 MB L2 while the (trait, tile) blocks of the chunk run;
  // each chunk a multiple of 16 wave groups (one per wavefront)
  // (SCOARY_LIST_ROUNDS = 1..64 replaces the 16 for same-box A/B runs, tools/ab_rounds_small.sh)
  static const long env_rounds = [] {
    const char* e = std::getenv("SCOARY_LIST_ROUNDS");
    const long v = e ? std::strtol(e, nullptr, 10) : 0;
    return (v >= 1 && v <= 64) ? v : 16L;
  }();
  int64_t chunks = ((int64_t)num_cu * env_rounds + g.ntiles * T - 1) / (g.ntiles * T);
  const int64_t chunk_bytes = list_chunk_bytes(TW);
  const int64_t by_l2 = (entries * 4 + chunk_bytes - 1) / chunk_bytes;
  if (chunks < by_l2) chunks = by_l2;
  if (chunks > 65535) chunks = 65535;
  if (chunks < 1) chunks = 1;
  g.gpb = (g.ngroups + chunks - 1) / chunks;
  g.gpb = (g.gpb + 15) / 16 * 16;
  g.chunks = (g.ngroups + g.gpb - 1) / g.gpb;
  return g;
}

template <int TW, int KC>
static int launch_permute_lists(scoary_handle h, hipStream_t s, const uint32_t* d_tiles,
                                const uint32_t* d_lidx, int64_t entries, const int32_t* d_lstart,
                                const int32_t* d_lngroups, const int32_t* d_lorder,
                                const uint8_t* d_lflipped, const uint32_t* d_crit,
                                const uint32_t* d_lcrit_in, const int32_t* d_margins,
                                uint32_t* d_scratch, int64_t G, int64_t T, int64_t N, int64_t P,
                                uint32_t* d_r, int accumulate) {
  uint32_t* d_lcrit_sc = d_scratch;                    // [T][G][2]
  uint16_t* d_partial = reinterpret_cast<uint16_t*>(d_scratch + 2 * T * G);   // [T][ntiles][gs]
  const uint32_t* d_lcrit = d_lcrit_in ? d_lcrit_in : d_lcrit_sc;
  if (!d_lcrit_in) {               // regions in gene order (scoary_fisher): convert to slot order
    KernelTimer kt(h, s, "k_lists_crit");
    hipLaunchKernelGGL(k_lists_crit, dim3((unsigned)((G + 255) / 256), (unsigned)T), dim3(256), 0, s,
                       reinterpret_cast<const uint2*>(d_crit), d_margins, d_lorder, d_lflipped,
                       (int)G, reinterpret_cast<uint2*>(d_lcrit_sc));
  }
  const ListGeom g = list_geom(h->num_cu, G, T, N, P, entries);
  if (T * g.ntiles > 0x7fffffffLL || g.chunks > 65535)
    return fail(h, SCOARY_ERR_SIZE, "scoary_permute_lists: grid too large");
  const size_t lds = (size_t)list_tile_dwords(N, TW) * sizeof(uint32_t);
  const void* fn = reinterpret_cast<const void*>(&k_permute_lists<list_lpg(TW), list_nw(TW), KC>);
  if (!(h->lists_lds_optin & TW)) {   // once per handle (= per device) and tile width
    // list entries of the one-lane-per-gene kernels are absolute LDS addresses: the label
    // tile must be the kernel's only LDS object (dynamic LDS then starts at address 0)
    hipFuncAttributes attr;
    HIP_TRY(h, hipFuncGetAttributes(&attr, fn));
    if (attr.sharedSizeBytes != 0)
      return fail(h, SCOARY_ERR_SIZE, "scoary_permute_lists: k_permute_lists has static LDS; the "
                                      "label tile would not sit at LDS address 0");
    HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    h->lists_lds_optin |= TW;
  }
  {
    KernelTimer kt(h, s, "k_permute_lists");
    hipLaunchKernelGGL((k_permute_lists<list_lpg(TW), list_nw(TW), KC>),
                       dim3((unsigned)(T * g.ntiles), (unsigned)g.chunks), dim3(1024), lds, s, d_tiles,
                       d_lidx, d_lstart, d_lngroups, reinterpret_cast<const uint2*>(d_lcrit), (int)G,
                       (int)N, P, (int)g.ntiles, (int)g.gpb,
                       (entries + kListSlack) * (int64_t)sizeof(uint32_t), d_partial);
  }
  {
    KernelTimer kt(h, s, "k_lists_reduce");
    hipLaunchKernelGGL(k_lists_reduce, dim3((unsigned)((G + kWave - 1) / kWave), (unsigned)T), dim3(256), 0, s,
                       d_partial, (int)g.ntiles, g.gs, (int)G, d_lorder, accumulate, d_r);
  }
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}
// N > 20479: the segmented kernel (k_permute_seglists), same scratch layout and geometry
static int launch_permute_seglists(scoary_handle h, hipStream_t s, const uint32_t* d_tiles,
                                   const uint32_t* d_lidx, int64_t entries, const int32_t* d_lstart,
                                   const int32_t* d_lngroups, const int32_t* d_lorder,
                                   const uint8_t* d_lflipped, const uint32_t* d_crit,
                                   const uint32_t* d_lcrit_in, const int32_t* d_margins,
                                   uint32_t* d_scratch, int64_t G, int64_t T, int64_t N, int64_t P,
                                   uint32_t* d_r, int accumulate) {
  constexpr int KC = 16;
  uint32_t* d_lcrit_sc = d_scratch;
  uint16_t* d_partial = reinterpret_cast<uint16_t*>(d_scratch + 2 * T * G);
  const uint32_t* d_lcrit = d_lcrit_in ? d_lcrit_in : d_lcrit_sc;
  if (!d_lcrit_in) {
    KernelTimer kt(h, s, "k_lists_crit");
    hipLaunchKernelGGL(k_lists_crit, dim3((unsigned)((G + 255) / 256), (unsigned)T), dim3(256), 0, s,
                       reinterpret_cast<const uint2*>(d_crit), d_margins, d_lorder, d_lflipped,
                       (int)G, reinterpret_cast<uint2*>(d_lcrit_sc));
  }
  const ListGeom g = list_geom(h->num_cu, G, T, N, P, entries);
  if (T * g.ntiles > 0x7fffffffLL || g.chunks > 65535)
    return fail(h, SCOARY_ERR_SIZE, "scoary_permute_lists: grid too large");
  const size_t lds = (size_t)kSegStride * sizeof(uint32_t);
  const void* fn = reinterpret_cast<const void*>(&k_permute_seglists<KC>);
  constexpr int kOptinBit = 32;                         // next to the tile widths 16 / 8 / 4 / 2
  if (!(h->lists_lds_optin & kOptinBit)) {
    hipFuncAttributes attr;
    HIP_TRY(h, hipFuncGetAttributes(&attr, fn));
    if (attr.sharedSizeBytes != 0)
      return fail(h, SCOARY_ERR_SIZE, "scoary_permute_lists: k_permute_seglists has static LDS; the "
                                      "label tile would not sit at LDS address 0");
    HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    h->lists_lds_optin |= kOptinBit;
  }
  {
    KernelTimer kt(h, s, "k_permute_seglists");
    hipLaunchKernelGGL((k_permute_seglists<KC>), dim3((unsigned)(T * g.ntiles), (unsigned)g.chunks),
                       dim3(1024), lds, s, d_tiles, d_lidx, d_lstart, d_lngroups,
                       reinterpret_cast<const uint2*>(d_lcrit), (int)G, (int)N, P, (int)g.ntiles,
                       (int)g.gpb, (entries + kListSlack) * (int64_t)sizeof(uint32_t),
                       list_segments(N), d_partial);
  }
  {
    KernelTimer kt(h, s, "k_lists_reduce");
    hipLaunchKernelGGL(k_lists_reduce, dim3((unsigned)((G + kWave - 1) / kWave), (unsigned)T), dim3(256), 0, s,
                       d_partial, (int)g.ntiles, g.gs, (int)G, d_lorder, accumulate, d_r);
  }
  HIP_TRY(h, hipGetLastError());
  return SCOARY_OK;
}
}  // extern "C++"

int64_t scoary_permute_lists_scratch_bytes(int64_t G, int64_t T, int64_t N, int64_t P) {
  if (G < 1 || T < 1 || N < 1 || P < 1 || !list_tw(N)) return 0;
  const ListGeom g = list_geom(256, G, T, N, P, 0);
  return 2 * T * G * (int64_t)sizeof(uint32_t) + (T * g.ntiles * g.gs * (int64_t)sizeof(uint16_t) + 3) / 4 * 4;
}

int scoary_permute_lists(scoary_handle h, const uint32_t* d_tiles, const uint32_t* d_lidx,
                         int64_t entries, const int32_t* d_lstart, const int32_t* d_lngroups,
                         const int32_t* d_lorder, const uint8_t* d_lflipped,
                         const uint32_t* d_crit, const uint32_t* d_lcrit, const int32_t* d_margins,
                         void* d_scratch, int64_t G, int64_t T, int64_t N, int64_t P,
                         uint32_t* d_r, int accumulate, scoary_stream_t stream) {
  if (!h) return SCOARY_ERR_ARG;
  if (!d_tiles || !d_lidx || !d_lstart || !d_lngroups || !d_lorder || !d_lflipped ||
      (!d_crit && !d_lcrit) || (!d_lcrit && !d_margins) || !d_scratch || !d_r || G < 1 || T < 1 ||
      N < 1 || P < 1 || entries < 0)
    return fail(h, SCOARY_ERR_ARG, "scoary_permute_lists: bad argument");
  const int TW = list_tw(N);
  if (!TW)
    return fail(h, SCOARY_ERR_SIZE, "scoary_permute_lists: label tile does not fit in LDS for this N");
  if (T > 65535) return fail(h, SCOARY_ERR_SIZE, "scoary_permute_lists: T > 65535");
  DeviceGuard guard(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  uint32_t* sc = static_cast<uint32_t*>(d_scratch);
  if (list_segments(N) > 1)
    return launch_permute_seglists(h, s, d_tiles, d_lidx, entries, d_lstart, d_lngroups, d_lorder,
                                   d_lflipped, d_crit, d_lcrit, d_margins, sc, G, T, N, P, d_r,
                                   accumulate);
  // counter planes KC: lists hold <= N/2 entries, N/2 < 2^KC (and N + 1 < 2^(KC+1))
#define LAUNCH(TWV, KCV)                                                                        \
  return launch_permute_lists<TWV, KCV>(h, s, d_tiles, d_lidx, entries, d_lstart, d_lngroups,   \
                                        d_lorder, d_lflipped, d_crit, d_lcrit, d_margins, sc, G, \
                                        T, N, P, d_r, accumulate)
  if (TW == 16) LAUNCH(16, 11);
  if (TW == 8) LAUNCH(8, 12);
  if (TW == 4) LAUNCH(4, 13);
  LAUNCH(2, 14);
#undef LAUNCH
}

}  // extern "C"
