/*
 * scoary_io.h -- native input codec: gene presence/absence CSV -> bit rows.
 *
 * Replaces the per-cell Python loop of Csv_to_dic_Roary
 * (scoary/methods.py:335-508) for files too large for it:
 * one streaming pass over the memory-mapped file, Python-csv-compatible
 * tokenisation (excel dialect + skipinitialspace, as at scoary/methods.py:
 * 350-351), presence rule "cell not in {'', '0', '-'}" (:476-485), output
 * directly as rows64 (bit i of word w = kept strain column 64*w+i).
 * Host-only (no GPU); plain C ABI, bound with ctypes by scoary_amd/io_native.py.
 */
#ifndef SCOARY_IO_H
#define SCOARY_IO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct scoary_gpa *scoary_gpa_t;

/* Parse `path`.  Columns [0, startcol) are kept as text, columns >= startcol
 * are strains.  keep: NULL, or one byte per strain column (in file order,
 * length = header cells - startcol) -- non-zero = include that strain.
 * Because the header must be read before `keep` can be built, parsing is two
 * calls: scoary_gpa_open reads the header only; scoary_gpa_parse the body.
 * Return 0 or a negative error; message via scoary_gpa_error. */
int scoary_gpa_open(const char *path, char delimiter, int64_t startcol, scoary_gpa_t *out);
int scoary_gpa_parse(scoary_gpa_t g, const uint8_t *keep);
/* The same with an explicit thread count and minimum bytes per thread: the body is cut at
 * line ends into ranges parsed in parallel; if a cut turns out to lie inside a quoted
 * cell (a range does not stop where the next one starts) the body is parsed in one piece. */
int scoary_gpa_parse_mt(scoary_gpa_t g, const uint8_t *keep, int64_t threads, int64_t min_chunk);
/* One process of several (one rank per GPU under torchrun): the body is cut at line ends
 * into `nparts` byte ranges and only range `part` is parsed (with `threads` threads), so
 * that the ranks read a large table together instead of each reading all of it.  Rows come
 * back in file order within the part; the caller concatenates the parts in rank order.
 * Returns -6 if a part boundary turned out to lie inside a quoted multi-line cell: every
 * rank must then fall back to scoary_gpa_parse (the ranks have to agree on that). */
int scoary_gpa_parse_part(scoary_gpa_t g, const uint8_t *keep, int64_t part, int64_t nparts,
                          int64_t threads, int64_t min_chunk);
void scoary_gpa_close(scoary_gpa_t g);
const char *scoary_gpa_error(scoary_gpa_t g);

/* Header cells: count, total bytes; copy-out as lengths[] + concatenated bytes. */
int64_t scoary_gpa_header_cells(scoary_gpa_t g);
int64_t scoary_gpa_header_bytes(scoary_gpa_t g);
void scoary_gpa_header_copy(scoary_gpa_t g, int32_t *lengths, char *bytes);

/* After scoary_gpa_parse: rows, kept strains, words per row. */
int64_t scoary_gpa_rows(scoary_gpa_t g);
int64_t scoary_gpa_strains(scoary_gpa_t g);
int64_t scoary_gpa_words(scoary_gpa_t g);
/* rows64 [rows][words] */
void scoary_gpa_bits_copy(scoary_gpa_t g, uint64_t *out);
/* the text cells of columns [0, startcol) of every row, row-major:
 * lengths [rows*startcol], bytes concatenated in the same order */
int64_t scoary_gpa_meta_bytes(scoary_gpa_t g);
void scoary_gpa_meta_copy(scoary_gpa_t g, int32_t *lengths, char *bytes);

/* ---- minority index lists for the list-driven permutation kernel ----------
 * (scoary_permute_lists, include/scoary_hip.h).  For every gene row of a
 * rows64 matrix [G][W64] over N isolates: the positions of its MINORITY value
 * (ones if popcount <= N/2, else zeros; flipped[g] = 1 in the latter case),
 * padded with the value N (an all-zero row on the device) to a multiple of 16
 * entries (half a 32-entry step of the kernel) and to the longest list of its
 * wavefront group.  Lists are laid out
 * back to back in `order`: genes sorted by descending list length, so that the
 * `genes_per_wave` genes a wavefront processes together (slots w*gpw ..) have
 * similar lengths and, after padding, the same number of 32-entry groups.
 * Within a list (spec S6) the positions are ordered by
 * rank-within-class * classes + ((class - k) mod classes), class = position mod
 * classes, ascending position within a class, written without gaps: while every
 * class still has positions, entry e of slot k comes from residue class (k + e)
 * mod classes (LDS bank trick, see the kernel).  This host builder is the CHECKER
 * of the device builder (scoary_lists_plan / scoary_lists_fill, scoary_hip.h), which
 * the product path uses; both implement the spec independently.
 * With piece > 0 the lists of one wavefront group are interleaved in pieces of
 * `piece` entries -- entry e of the group's j-th gene sits at
 * group_base + ((e / piece) * genes_per_wave + j) * piece + e % piece -- so that
 * the wavefront's index loads are contiguous; start[] is then the group base
 * for every slot of the group and the last group is stored in full (missing
 * genes all padding).  piece = 0: every list contiguous, back to back.
 * row_stride / genes_per_wave / classes / piece come from scoary_list_params(N);
 * classes must be a power of two (1..64), piece must divide 32.
 *   first call  : scoary_lists_count -> total number of entries
 *   second call : scoary_lists_build fills
 *       idx     uint32 [total]   entry = position * row_stride (the LDS byte offset of
 *                                that isolate's label row in the list-driven kernel)
 *       start   int32  [G]       first entry of gene order[k]: in units of 32 entries
 *                                (piece > 0, the group base) or of 16 (piece = 0)
 *       ngroups int32  [G]       half-steps of 16 entries of gene order[k] (equal within a wave group)
 *       order   int32  [G]       gene id of slot k
 *       flipped uint8  [G]       per gene id */
int64_t scoary_lists_count(const uint64_t *rows64, int64_t G, int64_t N, int64_t genes_per_wave,
                           int64_t piece);
void scoary_lists_build(const uint64_t *rows64, int64_t G, int64_t N, int64_t row_stride,
                        int64_t genes_per_wave, int64_t classes, int64_t piece, uint32_t *idx,
                        int32_t *start, int32_t *ngroups, int32_t *order, uint8_t *flipped);

/* ---- vcf2scoary record lines (scoary/vcf2scoary.py:170-214) -------------------
 * Converts the variant lines of a VCF -- everything from byte `offset`, i.e. after
 * the #CHROM header line -- to rows of the Roary/Scoary-style table and APPENDS
 * them to out_path: one row per ALT allele, the nine fixed columns, DUMMY
 * ("False", or "True" for rows split out of a multi-allelic site), one genotype
 * cell per sample (first ':'-field; at multi-allelic sites "1" if it names this
 * allele, else "0", "." -> "0"), every cell in double quotes.  types: NULL, or
 * a comma-separated list of INFO TYPE= values to keep.
 * Returns the number of rows written; -1 on an I/O error; -2 if the file needs
 * the general reader (quote characters, lone carriage returns, short lines,
 * genotypes that are not plain digits, missing TYPE=): the caller then discards
 * the output and runs the Python implementation, which reproduces the
 * reference's behaviour for those inputs. */
int64_t scoary_vcf_convert(const char *vcf_path, int64_t offset, const char *out_path,
                           const char *types);

/* ---- a9: the per-trait results file (StoreTraitResult, scoary/methods.py:1003-1197) -------------
 * Writes `header` (the finished header line, newline included) and nrows rows to `path`.  A row is
 * ntext text cells followed by nnum numeric cells, every cell wrapped in double quotes
 * (methods.py:1052, :1195-1197), joined by `delimiter`, '\n' after the last cell.
 *   text column c : the cell of row r is bytes [text_off[c][k], text_off[c][k+1]) of text_blob[c]
 *                   with k = text_row[c][r] -- a column is a string table (e.g. the gene table's
 *                   identifiers, built once per table) and an index per written row, so a trait's
 *                   file costs no Python object per row;
 *   numeric col c : value cols[c][num_row[r]]; kind[c] = 0: int64, printed as a decimal;
 *                   kind[c] = 1: float64, printed as Python's repr(float) -- the shortest digits
 *                   that round-trip, fixed notation while 1e-4 <= |x| < 1e16 (".0" appended to
 *                   integers), else d.ddde[+-]XX with at least two exponent digits; "inf", "-inf",
 *                   "nan" -- which is what str() of the reference's numpy.float64 / float cells
 *                   gives.
 * Rows are formatted by `threads` OpenMP threads (<= 0: all) in blocks and written in order.
 * Returns the bytes written, or -1 if the file cannot be opened / written, -2 on a bad argument. */
int64_t scoary_results_write(const char *path, char delimiter, const char *header, int64_t header_len,
                             int64_t nrows, int32_t ntext, const char *const *text_blob,
                             const int64_t *const *text_off, const int64_t *const *text_row,
                             int32_t nnum, const int32_t *kind, const void *const *cols,
                             const int64_t *num_row, int64_t threads);
/* repr(float) of one value into buf (>= 32 bytes); returns the length (the formatter above, exposed
 * so that it can be checked against the interpreter value by value). */
int32_t scoary_format_float_repr(double x, char *buf);

/* ---- UPGMA merge order (scoary/methods.py:640-707, scoary/classes.py:68-196) ------
 * D: n x n float64 distances, row-major, diagonal already forced to 1
 * (methods.py:636-638).  Runs the reference's merge loop -- quad tree of 2x2
 * block minima, smallest (value, i, j) per block, size-weighted averaging -- and
 * writes the n-1 merged index pairs (i keeps the new cluster, j dies) to
 * merges[2*(n-1)].  Returns 0; -1 bad argument; -2 internal inconsistency. */
int scoary_upgma_merges(const double *D, int64_t n, int32_t *merges);

#ifdef __cplusplus
}
#endif
#endif
