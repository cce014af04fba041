/*
 * scoary_hip.h -- C-ABI of the MI355X-native association engine for Scoary.
 *
 * Drop-in boundary for ONE path of AdmiralenOla/Scoary (v1.6.16): the per-gene
 * 2x2 contingency counting + Fisher exact test of Setup_results /
 * Perform_statistics and the --permute label-shuffling empirical-p loop.
 * The reference is pure Python and has no FFI of its own; each entry point
 * below names the reference code it replaces (paths relative to the reference
 * checkout) -- that call site is where a maintainer binds it (ctypes stub in
 * INTEGRATION.md).
 *
 * Conventions
 *   - every `d_` pointer is DEVICE memory on the handle's GPU, owned by the
 *     caller (e.g. a torch tensor's data_ptr()); nothing is allocated or freed
 *     behind the caller's back except a small per-handle scratch;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream) and never synchronises the device;
 *   - return value: 0 = ok, negative = error (see scoary_last_error); the
 *     library never calls exit();
 *   - one handle per GPU; a handle is used by one host thread at a time.
 *
 * Device layouts (all little-endian)
 *   rows64   : uint64 [R][W64], W64 = ceil(N/64); bit i of word w of a row is
 *              isolate 64*w+i; pad bits are zero.  (How the reference's
 *              genedic / traitsdic 0/1 values are bit-packed)
 *   tiled    : the gene matrix as the kernels read it: uint32 [Qp][Gp][4]
 *              ("word-quad-major"): the 16 bytes at ((q*Gp)+g)*16 hold 32-bit
 *              words 4q..4q+3 of gene g's row.  Qp = scoary_tiled_quads(N),
 *              Gp = scoary_tiled_genes(G); padding genes / words are zero.
 *              One lane owns one gene, so a wavefront's loads are 1 KiB
 *              coalesced and the per-trait operand is wave-uniform.
 *   vecrows  : trait / validity-mask / permuted-label vectors: uint32 [R][Wp],
 *              Wp = scoary_row_words(N) = 4*Qp >= 2*W64 (a rows64 row followed
 *              by zero padding).
 */
#ifndef SCOARY_HIP_H
#define SCOARY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCOARY_ABI_VERSION 9

/* error codes */
#define SCOARY_OK 0
#define SCOARY_ERR_ARG (-1)    /* bad argument (null pointer, negative size) */
#define SCOARY_ERR_HIP (-2)    /* a HIP runtime call failed */
#define SCOARY_ERR_SIZE (-3)   /* problem size outside what the kernels support */
#define SCOARY_ERR_DEVICE (-4) /* no such device / not a gfx950 code object */

typedef struct scoary_ctx *scoary_handle;
typedef void *scoary_stream_t; /* hipStream_t */

int scoary_abi_version(void);

/* Create / destroy the per-GPU context. */
int scoary_create(int device, scoary_handle *out);
void scoary_destroy(scoary_handle h);
/* Message of the last failing call on this handle ("" if none). */
const char *scoary_last_error(scoary_handle h);

/* Layout arithmetic (pure functions, usable without a GPU). */
int64_t scoary_tiled_quads(int64_t N);            /* Qp */
int64_t scoary_tiled_genes(int64_t G);            /* Gp */
int64_t scoary_tiled_bytes(int64_t G, int64_t N); /* 16*Qp*Gp */
int64_t scoary_row_words(int64_t N);              /* Wp */

/* ---- a1: genedic -> packed bits ---------------------------------------
 * Replaces the dict-of-dicts the reference builds in Csv_to_dic_Roary
 * (scoary/methods.py:445-491; presence rule :476-485 applied by the caller).
 * scoary_pack_dense : d_dense is uint8 [G][N], non-zero = present.
 * scoary_tile_rows  : d_rows64 is rows64 [G][W64].
 * Both write the full tiled buffer (scoary_tiled_bytes), padding included. */
int scoary_pack_dense(scoary_handle h, const uint8_t *d_dense, int64_t G,
                      int64_t N, uint32_t *d_tiled, scoary_stream_t stream);
int scoary_tile_rows(scoary_handle h, const uint64_t *d_rows64, int64_t G,
                     int64_t N, uint32_t *d_tiled, scoary_stream_t stream);

/* ---- a3: Perform_statistics (scoary/methods.py:930-982) ----------------
 * For every (trait, gene): tpgp, tpgn, tngp, tngn over the isolates that are
 * valid for that trait.  d_traits / d_masks are vecrows [T][Wp] (label bits,
 * validity bits; traits & ~masks must be 0).
 *   d_counts  : int32 [T][G][4]   (tpgp, tpgn, tngp, tngn)  -- bit-exact
 *   d_margins : int32 [T][2]      (npos, nval) per trait
 * Not for a stream that is capturing a hipGraph (SCOARY_ERR_ARG: it allocates its plan in
 * stream-ordered temporary memory): record scoary_counts_planned instead. */
int scoary_counts(scoary_handle h, const uint32_t *d_tiled,
                  const uint32_t *d_traits, const uint32_t *d_masks, int64_t G,
                  int64_t T, int64_t N, int32_t *d_counts, int32_t *d_margins,
                  scoary_stream_t stream);
/* The same in two parts (ABI 7), for callers that run more than one step per trait set: the
 * trait PLAN -- everything the counts need that depends on the traits alone -- is built once
 * (as the index lists are built once per gene matrix), and the per-step call is one kernel
 * that streams the gene matrix once per pass of up to 32 traits.
 *   scoary_trait_plan : d_margins int32 [T][2] = (npos, nval) per trait (the isolate loop of
 *       scoary/methods.py:940-965 sees exactly the valid isolates: :591-598); d_mask_class
 *       int32 [T] (may be NULL) = the smallest t' <= t IN THE SAME PASS (traits
 *       [k tb, (k + 1) tb), tb = scoary_counts_traits_per_pass(T)) whose validity row equals
 *       trait t's -- traits of one class share popc(gene & valid), counted once per class and
 *       pass (most traits of a real file have no missing values: one class per pass; ABI 8: the
 *       search no longer leaves the pass, at most 31 row compares per trait); d_plan = an opaque
 *       buffer of scoary_trait_plan_bytes(T, N) bytes: the classes as dense slots per pass and
 *       the label / validity rows gathered quad-major, one contiguous operand row per gene
 *       quad and pass (what k_counts reads with wide scalar loads).
 *   scoary_counts_planned : d_counts as scoary_counts; d_plan / d_margins from
 *       scoary_trait_plan of the same traits, masks, T and N.
 * scoary_counts(...) == scoary_trait_plan into stream-ordered temporary memory +
 * scoary_counts_planned. */
int64_t scoary_counts_traits_per_pass(int64_t T);   /* traits one pass over the matrix takes (<= 32) */
int64_t scoary_trait_plan_bytes(int64_t T, int64_t N);
int scoary_trait_plan(scoary_handle h, const uint32_t *d_traits, const uint32_t *d_masks,
                      int64_t T, int64_t N, int32_t *d_margins, int32_t *d_mask_class,
                      void *d_plan, scoary_stream_t stream);
int scoary_counts_planned(scoary_handle h, const uint32_t *d_tiled, const void *d_plan,
                          const int32_t *d_margins, int64_t G, int64_t T, int64_t N,
                          int32_t *d_counts, scoary_stream_t stream);

/* ---- a5: scipy.stats.fisher_exact(obs_table) at scoary/methods.py:854 ---
 * Two-sided Fisher exact p and sample odds ratio for M 2x2 tables
 * [[tpgp, tpgn], [tngp, tngn]] (d_tables int32 [M][4]); SciPy >= 1.7
 * semantics: (nan, 1.0) when a margin is zero, inf odds when tpgn*tngp == 0;
 * a support point counts as "as extreme as observed" iff its weight is
 * <= the observed weight * (1 + 1e-14) -- SciPy's own factor -- and that
 * comparison is decided exactly (double-double) whenever the fp64 weights agree
 * to 1e-9: tables whose two closest support points are 1e-12 apart (they
 * exist: tests/golden/near_ties.json) get SciPy's p, exact ties are ties.
 * Tolerance: |p - scipy| < 1e-12 for N <= 40 000 (measured <= 3e-15).  Above that SciPy
 * 1.15.3 ITSELF leaves the exact value of its rule (6.0e-12 on (5582, 3263, 70693,
 * 40462), N = 120 000); the kernel follows the exact value -- pinned by the oracle
 * against integer arithmetic at N = 50 000 ... 131 070 (tests/test_oracle_golden.py).
 *   d_p, d_or : double [M]
 *   d_crit    : uint32 [M][2] or NULL -- the rejection region of table m as
 *               (base, span): a permuted table with overlap count a' is "as
 *               or more extreme" iff (uint32)(a' - base) >= span.  Consumed by
 *               scoary_permute.  Tables the reference never tests
 *               (gene absent / present in all valid isolates,
 *               scoary/methods.py:804-814) get p = 1, odds = nan, span = 0. */
int scoary_fisher(scoary_handle h, const int32_t *d_tables, int64_t M,
                  double *d_p, double *d_or, uint32_t *d_crit,
                  scoary_stream_t stream);
/* scipy.stats.fisher_exact's OWN double for the tables of 171 ... scoary_fisher_scipy_max_isolates() (104 723)
 * isolates: d_p[m] is overwritten with the value SciPy >= 1.7 returns for table m, to the last bit (Boost.Math's
 * prime-factorised hypergeometric pmf, its tail recurrences and fisher_exact's binary search restated in fp64;
 * scoary_amd/csrc/scoary_scipy.hip).  Replaces, for what a caller PRINTS, the per-gene call the reference makes
 * (scoary/methods.py:854): with it the result files are the reference's bytes at any size, not only up to 170
 * isolates where scoary_fisher already returns SciPy's double.  Tables with an empty margin, with fewer than 171
 * isolates or with more than the maximum are left as they are; *d_skipped (uint64, may be NULL; zero it first)
 * counts the tables above the maximum.  About 50 x the work of scoary_fisher per table (15 pmf evaluations over the
 * primes up to N; 9.5 ms per 500 000 tables at N = 2000): an output-fidelity pass for the command line, not part of the association step bench.py times.
 * The first call on a device uploads the prime table (a synchronous 40 KB copy: not inside a graph capture). */
int64_t scoary_fisher_scipy_max_isolates(void);
int scoary_fisher_scipy(scoary_handle h, const int32_t *d_tables, int64_t M, double *d_p,
                        uint64_t *d_skipped, scoary_stream_t stream);
/* The same test for the list-driven permutation path, one launch fewer per step:
 * tables [T][G][4] are visited in LIST-SLOT order (slot k of trait t = gene
 * d_lorder[k]; slots are sorted by minority count, so the lanes of a wavefront walk
 * supports of similar length), results land at the gene's own index as in
 * scoary_fisher (d_crit may be NULL), and the rejection region is also written in
 * the form scoary_permute_lists consumes: d_lcrit uint32 [T][G][2] = (lo, hi1) of the
 * LIST count in slot order (ones-list: [base, base + span); zeros-list:
 * [npos - base - span + 1, npos - base + 1); span 0 -> (0, 0)).
 *   d_lorder / d_lflipped : from scoary_lists_plan */
int scoary_fisher_lists(scoary_handle h, const int32_t *d_tables, int64_t T, int64_t G,
                        const int32_t *d_lorder, const uint8_t *d_lflipped,
                        double *d_p, double *d_or, uint32_t *d_crit, uint32_t *d_lcrit,
                        scoary_stream_t stream);

/* ---- a8: PermuteGTC (scoary/methods.py:1371-1384) ----------------------
 * Label permutations pi = perm_base .. perm_base+P-1 of the T traits whose
 * rows are given (their global trait numbers are trait_base .. trait_base+T-1,
 * the number that enters the Philox counter): the trait's npos positive labels
 * placed on a uniformly random subset of its valid isolates.  The reference's
 * shuffle is unseeded; here the bits are a pure function of (seed, trait,
 * permutation) -- Philox4x32-10 keyed by `seed` -- by a sampler in which no isolate
 * depends on the one before it (spec S4 of DESIGN.md, ABI 8; rounds 1-4 used
 * sequential selection sampling): min(npos, nval - npos) marks; round 0 marks every
 * valid isolate with an 8-bit probability (eight words of counter (isolate, pi >> 5,
 * trait, "SCOB" / "SCOC") shared by 32 permutations, compared bit-sliced); the
 * surplus or deficit is removed / added at uniformly drawn positions (counter
 * (draw >> 2, pi, trait, "SCOD"), Lemire rejection, non-candidates rejected).
 * Exactly uniform under ideal random words; the CPU oracle (oracle/oracle.c)
 * regenerates the same bits.
 *   d_perms : vecrows [T][P][Wp]
 * N <= scoary_perm_max_isolates() (a block keeps its rows in LDS). */
int64_t scoary_perm_max_isolates(void);
int scoary_perm_generate(scoary_handle h, const uint32_t *d_masks,
                         const int32_t *d_margins, int64_t T, int64_t N,
                         int64_t P, int64_t perm_base, int64_t trait_base,
                         uint64_t seed, uint32_t *d_perms, scoary_stream_t stream);

/* ---- a7: Permute (scoary/methods.py:1314-1369), Fisher statistic --------
 * d_r[t][g] += #{ pi < P : popcount(gene_g & perm_{t,pi}) lies in the
 * rejection region d_crit[t][g] }.  The caller zeroes d_r (uint32 [T][G])
 * before the first chunk of permutations; Empirical_p = (r+1)/(P_total+1)
 * (scoary/methods.py:1365). */
int scoary_permute(scoary_handle h, const uint32_t *d_tiled,
                   const uint32_t *d_perms, const uint32_t *d_crit, int64_t G,
                   int64_t T, int64_t N, int64_t P, uint32_t *d_r,
                   scoary_stream_t stream);

/* ---- a7 with the reference's sequential early abort (scoary/methods.py:1348-1365) ----
 * Opt-in (--permute-early-abort): every (gene, trait) consumes the permutations in index
 * order, r counting the ones in the rejection region; from permutation index i >= 30 on, the
 * first i with r >= d_thr[i] stops that gene: d_nstop[t][g] = i + 1 and Empirical_p =
 * (r + 1) / (i + 2) (`emp_p = (r+1.0)/(i+2.0)`, :1362); genes that never stop keep
 * d_nstop = 0 and get (r + 1) / (P_total + 1) (:1365).  d_thr[i] = smallest r with
 * 1 - binom.cdf(r, i, 0.1) < 0.05 (:1361) for 30 <= i < P_total, 0xffffffff below 30
 * (uint32 [P_total], built by the caller).  Call once per batch of permutations
 * [perm_base, perm_base + P) in ascending order; the caller zeroes d_r and d_nstop (uint32
 * [T][G]) before the first batch.  d_perms: vecrows [T][P][Wp] (scoary_perm_generate). */
int scoary_permute_seq(scoary_handle h, const uint32_t *d_tiled, const uint32_t *d_perms,
                       const uint32_t *d_crit, const uint32_t *d_thr, int64_t G, int64_t T,
                       int64_t N, int64_t P, int64_t perm_base, uint32_t *d_r,
                       uint32_t *d_nstop, scoary_stream_t stream);

/* ---- a7/a8, list-driven variant -------------------------------------------
 * Same result as scoary_perm_generate + scoary_permute (d_r is bit-identical),
 * different data flow: genes as lists of the isolates that carry their
 * minority value (built once per dataset, on the device: scoary_lists_plan /
 * scoary_lists_fill below), permuted labels as isolate-major tiles of 512 / 256 /
 * 128 / 64 permutations (N <= 2559 / 5119 / 10239 / beyond) that live in LDS, overlap
 * counts as bit-sliced counters, 128 (64 for the last) permutations per lane.  Cost
 * is proportional to the list length, so sparse (or near-core) genes are
 * cheap.  One 64-permutation tile fits in LDS up to N = 20479; wider matrices, up to N =
 * scoary_list_max_isolates() = 131070, are cut into scoary_list_segments(N) = 2 ... 7
 * SEGMENTS of 20352 isolates: a block loads its tile one segment at a time, a gene's
 * list is one sub-list per segment, the counters live across the reloads (round 3; the
 * dense kernels took over at N = 40960 before).
 *   d_tiles : uint32 [scoary_list_tiles_words(N, P, T)]
 *   d_lidx / d_lstart / d_lngroups / d_lorder / d_lflipped : the index lists of the
 *             gene matrix, built on the device by scoary_lists_plan + scoary_lists_fill
 *             (below); `entries` = the entry count scoary_lists_plan returned
 *   d_scratch : scoary_permute_lists_scratch_bytes(G, T, N, P) bytes: the rejection
 *             regions in list order and one 16-bit exceedance count per (trait,
 *             tile, list slot) -- the kernel writes them with plain stores and a
 *             small second kernel (k_lists_reduce) sums the tiles into d_r; no
 *             atomics (a device-scope atomic is a 32-byte memory-side write)
 *   d_crit  : regions in gene order from scoary_fisher (converted by a small kernel,
 *             needs d_margins), or NULL when
 *   d_lcrit : regions in slot order from scoary_fisher_lists is given (d_margins unused)
 *   d_r     : uint32 [T][G]; accumulate != 0: += like scoary_permute (further batches of
 *             permutations); accumulate == 0: overwritten (no zero fill needed) */
int64_t scoary_list_tiles_words(int64_t N, int64_t P, int64_t T);
int64_t scoary_list_tile_words(int64_t N);   /* dwords per (trait, tile) */
int64_t scoary_list_max_isolates(void);
int64_t scoary_list_segments(int64_t N);     /* 1 for N <= 20479, else ceil(N / 20352); 0: too large */
/* out5 = { tile row width in dwords of 32 permutations (16 for N <= 2559, 8 for N <= 5119,
 * 4 for N <= 10239, 2 beyond), row stride in bytes, genes per wavefront, residue classes,
 * interleave piece } -- the last four are the arguments scoary_lists_build
 * wants.  Error if N is too large. */
int scoary_list_params(int64_t N, int64_t *out5);
/* Label tiles of the permutations perm_base .. perm_base + P - 1 (perm_base a multiple of 32):
 * d_tiles = uint32 [T][ntiles][scoary_list_tile_words(N)], the same labels as
 * scoary_perm_generate.  _range writes only the flat (trait, tile) indices first_tile ..
 * first_tile + n_tiles - 1 of that array, in place: the ranks of a multi-GPU run can each
 * generate one contiguous share and all-gather the rest (scoary_amd/dist.py). */
int scoary_perm_generate_tiles(scoary_handle h, const uint32_t *d_masks,
                               const int32_t *d_margins, int64_t T, int64_t N, int64_t P,
                               int64_t perm_base, int64_t trait_base, uint64_t seed,
                               uint32_t *d_tiles, scoary_stream_t stream);
int scoary_perm_generate_tiles_range(scoary_handle h, const uint32_t *d_masks,
                                     const int32_t *d_margins, int64_t T, int64_t N, int64_t P,
                                     int64_t perm_base, int64_t trait_base, uint64_t seed,
                                     int64_t first_tile, int64_t n_tiles, uint32_t *d_tiles,
                                     scoary_stream_t stream);
int64_t scoary_permute_lists_scratch_bytes(int64_t G, int64_t T, int64_t N, int64_t P);
int scoary_permute_lists(scoary_handle h, const uint32_t *d_tiles, const uint32_t *d_lidx,
                         int64_t entries, const int32_t *d_lstart, const int32_t *d_lngroups,
                         const int32_t *d_lorder, const uint8_t *d_lflipped,
                         const uint32_t *d_crit, const uint32_t *d_lcrit,
                         const int32_t *d_margins, void *d_scratch,
                         int64_t G, int64_t T, int64_t N, int64_t P, uint32_t *d_r,
                         int accumulate, scoary_stream_t stream);

/* ---- index lists of the list-driven kernel, built on the device -----------------
 * From the tiled gene matrix already in HBM (no host pass, no PCIe copy of the
 * lists): for every gene the positions of its MINORITY value (ones if popcount <=
 * N/2, else zeros: d_flipped[g] = 1), genes ordered by descending list length
 * (stable), lists padded to 16 entries (half a kernel step) and to the longest list of their wavefront
 * group, the lists of a group interleaved in pieces of `piece` entries, entries in
 * the bank-rotation order of spec S6 (the same layout
 * scoary_lists_build of include/scoary_io.h -- the checker -- produces on the host).
 *   scoary_lists_plan : popcount -> flip -> radix sort by length -> padded lengths
 *       and group bases (prefix sum).  Writes d_order (int32 [G], per list slot),
 *       d_start / d_ngroups (int32 [scoary_list_segments(N)][G]: per segment and list
 *       slot -- plain [G] for N <= 20479) and d_flipped (uint8 [G], per gene); *entries_out (HOST)
 *       = total number of 32-bit words of index array (= entries for N <= 20479).  Synchronises `stream` (the caller needs the
 *       count to allocate d_idx).
 *   scoary_lists_fill : d_idx = uint32 [entries + scoary_lists_slack_entries()],
 *       entry = position * row stride (the LDS byte offset of that isolate's label
 *       row), padding = N * row stride (the all-zero row).  Segmented lists (N > 20479,
 *       round 4): 16-BIT entries, two per word of d_idx -- entry = position - segment start
 *       (the kernel scales it to the LDS address), padding = the segment's own zero row (=
 *       the segment's row count); *entries_out then counts 32-bit WORDS of d_idx (half the
 *       number of entries), eight entries per lane and 16-byte index vector, the entries of a
 *       sub-list in grid-compaction order (scoary_listbuild.hip).
 *   d_scratch : scoary_lists_scratch_bytes(G, N) bytes, the SAME buffer in both calls
 *       (plan leaves the lengths and group bases in it). */
int64_t scoary_lists_scratch_bytes(int64_t G, int64_t N);
int64_t scoary_lists_slack_entries(void);
int scoary_lists_plan(scoary_handle h, const uint32_t *d_tiled, int64_t G, int64_t N,
                      void *d_scratch, int32_t *d_start, int32_t *d_ngroups, int32_t *d_order,
                      uint8_t *d_flipped, int64_t *entries_out, scoary_stream_t stream);
int scoary_lists_fill(scoary_handle h, const uint32_t *d_tiled, int64_t G, int64_t N,
                      const void *d_scratch, const int32_t *d_order, const uint8_t *d_flipped,
                      int64_t entries, uint32_t *d_idx, scoary_stream_t stream);

/* ======================================================================
 * Population-structure stage
 * ====================================================================== */

/* ---- pdist(zeroonesmatrix, 'hamming') at scoary/methods.py:627-628 -------
 * Pairwise Hamming COUNTS between the R rows of a tiled bit matrix (here the
 * rows are isolates and the N columns are the variable genes; build it with
 * scoary_tile_rows from the transposed presence matrix):
 *   d_out[i][j] = popcount(row_i XOR row_j)        int32 [R][R]
 * d_vecrows is the same matrix as vecrows [R][Wp] (the wave-uniform operand).
 * The caller divides by the number of columns to get the reference's
 * fractions. */
int scoary_hamming(scoary_handle h, const uint32_t *d_tiled, const uint32_t *d_vecrows,
                   int64_t R, int64_t N, int32_t *d_out, scoary_stream_t stream);

/* ---- UPGMA merge loop (scoary/methods.py:640-707, scoary/classes.py:68-196) ----
 * From the n x n Hamming counts (scoary_hamming) over ncols variable genes: the
 * reference's merge order -- distances count / ncols with the diagonal forced to
 * 1, repeatedly the cell with the smallest (value, position in the reference's
 * quad tree of 2x2 block minima), size-weighted averaging in the same fp64
 * operations.  Writes the n-1 merged index pairs (i keeps the new cluster, j is
 * retired) to d_merges[2*(n-1)] and 0 to d_status[0]; a non-zero status means
 * the degenerate case (a minimum >= 1 or on the diagonal, in which the
 * reference merges retired clusters): the caller then runs the host loop
 * (scoary_upgma_merges, include/scoary_io.h), which mirrors that.
 *   d_scratch : scoary_upgma_scratch_bytes(n) bytes */
int64_t scoary_upgma_scratch_bytes(int64_t n);
int scoary_upgma(scoary_handle h, const int32_t *d_counts, int64_t n, int64_t ncols, void *d_scratch,
                 int32_t *d_merges, int32_t *d_status, scoary_stream_t stream);

/* ---- bit gather: reorder the columns of bit rows --------------------------
 * d_out[r] bit k = d_rows[r] bit d_index[k]   (k < K; output rows are
 * uint32 [R][Wout], Wout = (K+31)/32, pad bits zero).  Used to bring gene rows
 * and permuted label rows into the tip order of a (pruned) tree. */
int scoary_gather_bits(scoary_handle h, const uint32_t *d_rows, int64_t R, int64_t Wsrc,
                       const int32_t *d_index, int64_t K, uint32_t *d_out,
                       scoary_stream_t stream);

/* ---- PhyloTree maxima: ConvertUPGMAtoPhyloTree, scoary/methods.py:1386-1402
 *      + PhyloTree/Tip, scoary/classes.py:199-592 ---------------------------
 * The binary tree is a stack program over its K tips (children order is
 * irrelevant to the result):  op >= 0: push tip `op`;  op == -1: merge the two
 * top entries;  op <= -2: merge the top entry with tip (-2 - op).
 * `stack_depth` = the deepest the stack gets (host computes it; <= 32).  K <= 32767
 * (pair counts are packed into 14-bit fields; more tips: SCOARY_ERR_SIZE).
 * Tip k of evaluation (g, l) is in state  (gene bit k of row g ? A : a) +
 * (label bit k of row l ? B : b); d_gene_bits uint32 [G][Wt], d_label_bits
 * uint32 [L][Wt], Wt = (K+31)/32.
 *   d_out : int32 [G][L][3] = max contrasting pairs, max supporting pairs,
 *           max opposing pairs (classes.py:246-249). */
int scoary_tree_pairs(scoary_handle h, const int32_t *d_ops, int64_t nops, int64_t stack_depth,
                      const uint32_t *d_gene_bits, const uint32_t *d_label_bits, int64_t G,
                      int64_t L, int64_t K, int32_t *d_out, scoary_stream_t stream);

/* ---- Permute with the reference's tree statistic, methods.py:1314-1369 ----
 * Same evaluation for L permuted label rows, reduced to the exceedance flag
 *   d_exceed[g][l] = Total_l > 0  &&  double(X_l)/double(Total_l) >= est_g,
 * X = supporting pairs if the OBSERVED tree has Pro >= Anti else opposing pairs
 * (methods.py:1333-1355); d_obs int32 [G][3] = observed (Total, Pro, Anti).
 * A permuted tree with Total == 0 (ZeroDivisionError in the reference) is 0.
 * The host applies the sequential estimator / early abort (methods.py:1360-1365)
 * to the flags. */
int scoary_tree_permute(scoary_handle h, const int32_t *d_ops, int64_t nops,
                        int64_t stack_depth, const uint32_t *d_gene_bits,
                        const uint32_t *d_label_bits, int64_t G, int64_t L, int64_t K,
                        const int32_t *d_obs, uint8_t *d_exceed, scoary_stream_t stream);

/* ---- --collapse: presence-pattern identity, scoary/methods.py:816-840, :981
 * 128-bit hash of every gene's presence pattern restricted to each trait's
 * valid isolates (row AND mask):  d_out uint64 [T][G][2].  Equal patterns give
 * equal hashes; the host groups by hash and confirms equality on the bit rows,
 * so a collision can cost time but never correctness. */
int scoary_row_hash(scoary_handle h, const uint32_t *d_tiled, const uint32_t *d_masks,
                    int64_t G, int64_t T, int64_t N, uint64_t *d_out, scoary_stream_t stream);

/* ---- result records for the one exchange step of the path --------------
 * The reference weaves per-gene result dicts back from its worker processes
 * (scoary/methods.py:1115-1122); here every rank packs its shard into fixed records and
 * one RCCL gather / all-gather moves them (scoary_amd/dist.py):
 *   d_rec[i] = { tpgp, tpgn, tngp, tngn, p lo, p hi, odds lo, odds hi, r, nstop }
 * (uint32 [M][10], bit patterns preserved) for the M = T * G_shard (trait, gene) pairs of
 * d_counts [M][4] / d_p / d_odds / d_r / d_nstop; d_r and d_nstop may be NULL (zeros). */
int scoary_pack_records(scoary_handle h, const int32_t *d_counts, const double *d_p,
                        const double *d_odds, const uint32_t *d_r, const uint32_t *d_nstop,
                        int64_t M, uint32_t *d_rec, scoary_stream_t stream);

/* ---- e: the exchange step itself, for hosts that do not go through torch.distributed ----------
 * (replaces the pickled result weave of scoary/methods.py:1115-1122.)  Gather to `root`:
 * every rank sends `bytes` bytes at d_send (its scoary_pack_records block); the root receives nranks
 * blocks into d_recv, rank r's at offset r * bytes (d_recv may be NULL elsewhere).  One RCCL group of
 * ncclSend / ncclRecv on `stream`, over xGMI inside a node; returns once the group is enqueued.
 *   comm    : the caller's ncclComm_t (as void*: this header does not include rccl.h)
 *   rccl_dl : dlopen handle of the RCCL instance `comm` was created with (a process can hold more
 *             than one librccl -- PyTorch ships its own); NULL: the one the loader already has,
 *             else "librccl.so.1".  libscoary_hip.so has no link-time dependency on RCCL.
 * scoary_amd itself issues the same exchange through torch.distributed (scoary_amd/dist.py). */
int scoary_gather(scoary_handle h, void *rccl_dl, void *comm, const void *d_send, void *d_recv,
                  int64_t bytes, int rank, int nranks, int root, scoary_stream_t stream);

/* ---- hipGraph capture ----------------------------------------------------------
 * Small workloads are launch-bound (BASELINE configs[1]: six kernels, 0.14 ms).
 * scoary_graph_begin puts `stream` into capture mode; every scoary_* call made on
 * that stream (and on streams forked from it by event waits) until
 * scoary_graph_end is recorded instead of executed.  The captured sequence must not
 * allocate, synchronise or read results back: use caller-owned buffers that stay
 * alive for as long as the graph is replayed (so: scoary_counts_planned with a plan
 * built beforehand, not the one-call scoary_counts, which takes temporary memory for
 * its plan).  scoary_graph_launch replays it.
 * Not available while per-kernel timing (scoary_set_timing) is on. */
typedef struct scoary_graph *scoary_graph_t;
int scoary_graph_begin(scoary_handle h, scoary_stream_t stream);
int scoary_graph_end(scoary_handle h, scoary_stream_t stream, scoary_graph_t *out);
int scoary_graph_launch(scoary_handle h, scoary_graph_t g, scoary_stream_t stream);
void scoary_graph_destroy(scoary_graph_t g);

/* Name + average device time (ms, hipEvent on `stream`) of the kernels the
 * last scoary_permute call launched; for bench.py's roofline line.  Costs a
 * stream sync; timing is recorded only after scoary_set_timing(h, 1). */
int scoary_set_timing(scoary_handle h, int enabled);
int scoary_last_kernel_ms(scoary_handle h, const char *kernel, double *ms_out);

#ifdef __cplusplus
}
#endif
#endif /* SCOARY_HIP_H */
