// scoary_io.cpp -- streaming gene presence/absence CSV reader (see
// include/scoary_io.h).  Tokeniser = Python's csv "excel" dialect with
// skipinitialspace: quote char '"' opens a quoted field only at field start,
// "" inside quotes is a literal quote, text after a closing quote is appended,
// line ends \n, \r, \r\n outside quotes, newlines kept inside quotes.
#include "scoary_io.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#ifdef _OPENMP
#include <omp.h>
#endif
#include <algorithm>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

struct scoary_gpa {
  int fd = -1;
  const char* data = nullptr;
  size_t size = 0, pos = 0;
  char delim = ',';
  int64_t startcol = 14;
  std::string err;
  std::vector<std::string> header;
  // body
  int64_t rows = 0, strains = 0, words = 0;
  std::vector<uint64_t> bits;
  std::vector<int32_t> meta_len;
  std::string meta;
};

namespace {

// Reads one record starting at g->pos; calls sink(col, ptr, len) with the
// unquoted content of every cell.  Returns false at end of input.  A state
// machine equivalent to CPython's _csv reader for the excel dialect with
// skipinitialspace (text mode: \r\n and \r read as \n, also inside quotes).
struct Cursor {
  const char* data;
  size_t size, pos;
  char delim;
};

template <class Sink>
bool read_record(Cursor* g, std::string& scratch, Sink&& sink) {
  const char* d = g->data;
  const size_t n = g->size;
  size_t p = g->pos;
  if (p >= n) return false;
  enum { START_RECORD, START_FIELD, IN_FIELD, IN_QUOTED, QUOTE_IN_QUOTED } st = START_RECORD;
  int64_t col = 0;
  size_t fstart = 0;      // unquoted field: [fstart, p) is the content
  bool in_scratch = false;
  auto is_eol = [&](size_t q) { return d[q] == '\n' || d[q] == '\r'; };
  auto eat_eol = [&](size_t q) { return (d[q] == '\r' && q + 1 < n && d[q + 1] == '\n') ? q + 2 : q + 1; };
  auto save = [&](size_t end) {
    if (in_scratch) sink(col, scratch.data(), scratch.size());
    else sink(col, d + fstart, end - fstart);
    ++col;
    in_scratch = false;
  };
  for (;;) {
    if (p >= n) {                        // end of input acts as an end of line
      if (st == START_FIELD) { fstart = p; save(p); }
      else if (st == IN_FIELD || st == QUOTE_IN_QUOTED || st == IN_QUOTED) save(p);
      g->pos = p;
      return true;
    }
    const char c = d[p];
    switch (st) {
      case START_RECORD:
        if (is_eol(p)) {                 // empty line: a record with no cells
          g->pos = eat_eol(p);
          return true;
        }
        st = START_FIELD;
        continue;                        // re-dispatch the same character
      case START_FIELD:
        if (is_eol(p)) {
          fstart = p;
          save(p);
          g->pos = eat_eol(p);
          return true;
        } else if (c == '"') {
          scratch.clear();
          in_scratch = true;
          st = IN_QUOTED;
        } else if (c == ' ') {
          // skipinitialspace
        } else if (c == g->delim) {
          fstart = p;
          save(p);
        } else {
          fstart = p;
          st = IN_FIELD;
        }
        ++p;
        break;
      case IN_FIELD:
        if (is_eol(p)) {
          save(p);
          g->pos = eat_eol(p);
          return true;
        } else if (c == g->delim) {
          save(p);
          st = START_FIELD;
        } else if (in_scratch) {
          scratch.push_back(c);
        }
        ++p;
        break;
      case IN_QUOTED:
        if (c == '"') {
          st = QUOTE_IN_QUOTED;
          ++p;
        } else if (c == '\r') {
          scratch.push_back('\n');
          p = eat_eol(p);
        } else {
          scratch.push_back(c);
          ++p;
        }
        break;
      case QUOTE_IN_QUOTED:
        if (c == '"') {
          scratch.push_back('"');
          st = IN_QUOTED;
          ++p;
        } else if (c == g->delim) {
          save(p);
          st = START_FIELD;
          ++p;
        } else if (is_eol(p)) {
          save(p);
          g->pos = eat_eol(p);
          return true;
        } else {                         // non-strict: keep going as a plain field
          scratch.push_back(c);
          st = IN_FIELD;
          ++p;
        }
        break;
    }
  }
}

}  // namespace

extern "C" {

int scoary_gpa_open(const char* path, char delimiter, int64_t startcol, scoary_gpa_t* out) {
  if (!path || !out || startcol < 0) return -1;
  scoary_gpa* g = new scoary_gpa();
  *out = g;
  g->delim = delimiter;
  g->startcol = startcol;
  g->fd = ::open(path, O_RDONLY);
  if (g->fd < 0) {
    g->err = std::string("cannot open ") + path;
    return -2;
  }
  struct stat st;
  if (fstat(g->fd, &st) != 0) {
    g->err = "fstat failed";
    return -2;
  }
  g->size = (size_t)st.st_size;
  if (g->size == 0) {
    g->err = "empty file";
    return -3;
  }
  void* m = mmap(nullptr, g->size, PROT_READ, MAP_PRIVATE, g->fd, 0);
  if (m == MAP_FAILED) {
    g->err = "mmap failed";
    return -2;
  }
  madvise(m, g->size, MADV_SEQUENTIAL);
  g->data = static_cast<const char*>(m);
  std::string scratch;
  Cursor cur{g->data, g->size, g->pos, g->delim};
  const bool got_header =
      read_record(&cur, scratch, [&](int64_t, const char* p, size_t n) { g->header.emplace_back(p, n); });
  g->pos = cur.pos;
  if (!got_header) {
    g->err = "no header";
    return -3;
  }
  if ((int64_t)g->header.size() <= startcol) {
    g->err = "startcol beyond the header";
    return -4;
  }
  return 0;
}

namespace {
// Rows parsed from one byte range of the body.
struct Piece {
  std::vector<uint64_t> bits;
  std::vector<int32_t> meta_len;
  std::string meta;
  int64_t rows = 0;
  int64_t bad_cells = -1;   // >= 0: the row after `rows` has this many cells (< header cells)
  size_t stop = 0;          // start of the first record not consumed
};

// Parses the records that START in [begin, limit).
void parse_range(const scoary_gpa* g, const std::vector<int32_t>& slot, size_t begin, size_t limit,
                 Piece* out) {
  const int64_t ncols = (int64_t)g->header.size(), sc = g->startcol, nstr = ncols - sc;
  const size_t words = (size_t)g->words;
  Cursor cur{g->data, g->size, begin, g->delim};
  std::string scratch;
  while (cur.pos < limit) {
    const size_t base = out->bits.size();
    out->bits.resize(base + words, 0);
    const size_t meta_base = out->meta_len.size();
    out->meta_len.resize(meta_base + (size_t)sc, -1);
    int64_t cells = 0;
    uint64_t* row = out->bits.data() + base;
    const bool got = read_record(&cur, scratch, [&](int64_t col, const char* p, size_t n) {
      ++cells;
      if (col < sc) {
        out->meta_len[meta_base + (size_t)col] = (int32_t)n;
        out->meta.append(p, n);
      } else if (col - sc < nstr) {
        const int32_t b = slot[col - sc];
        if (b >= 0) {
          const bool absent = n == 0 || (n == 1 && (p[0] == '0' || p[0] == '-'));
          if (!absent) row[b >> 6] |= (uint64_t)1 << (b & 63);
        }
      }
    });
    if (!got || cells < ncols) {
      out->bits.resize(base);
      out->meta_len.resize(meta_base);
      if (got) out->bad_cells = cells;   // the reference indexes q[startcol + strain]: IndexError -> exit
      break;
    }
    ++out->rows;
  }
  out->stop = cur.pos;
}

// Record starts for `k` ranges of the body: the guessed cuts are moved to the
// character after the next line end.  A cut that falls inside a quoted cell is
// caught later: the range before it then does not stop exactly there.
std::vector<size_t> guess_cuts(const scoary_gpa* g, size_t body, size_t end, int64_t k) {
  std::vector<size_t> cuts{body};
  const size_t total = end - body;
  for (int64_t i = 1; i < k; ++i) {
    size_t q = body + (size_t)((double)total * (double)i / (double)k);
    while (q < end && g->data[q] != '\n' && g->data[q] != '\r') ++q;
    if (q >= end) break;
    q = (g->data[q] == '\r' && q + 1 < g->size && g->data[q + 1] == '\n') ? q + 2 : q + 1;
    if (q > cuts.back() && q < end) cuts.push_back(q);
  }
  cuts.push_back(end);
  return cuts;
}

// Parses the records of [lo, hi) (lo, hi at record starts) in up to `threads` ranges.  Returns
// false if a guessed cut -- an inner one, or `hi` itself when it is not the end of the file --
// turned out to lie inside a quoted cell (a range does not stop where the next one starts).
bool parse_span(const scoary_gpa* g, const std::vector<int32_t>& slot, size_t lo, size_t hi,
                int64_t threads, int64_t min_chunk, std::vector<Piece>* pieces) {
  if (min_chunk < 1) min_chunk = 1;
  int64_t k = (int64_t)((hi - lo) / (size_t)min_chunk);
  if (k > threads) k = threads;
  if (k < 1) k = 1;
  std::vector<size_t> cuts = guess_cuts(g, lo, hi, k);
  pieces->assign(cuts.size() - 1, Piece());
#pragma omp parallel for schedule(static, 1) num_threads((int)pieces->size())
  for (int64_t i = 0; i < (int64_t)pieces->size(); ++i)
    parse_range(g, slot, cuts[i], cuts[i + 1], &(*pieces)[i]);
  for (size_t i = 0; i < pieces->size(); ++i) {
    if ((*pieces)[i].bad_cells >= 0) return true;     // a malformed row: reported by the caller
    if ((*pieces)[i].stop != cuts[i + 1] && (i + 1 < pieces->size() || hi < g->size)) return false;
  }
  return true;
}

// range_start > 0: the pieces are one rank's byte range of the body -- the rows before it were
// parsed elsewhere, so the bad row is named by its position inside the range
int finish_pieces(scoary_gpa* g, std::vector<Piece>& pieces, size_t range_start = 0) {
  const int64_t ncols = (int64_t)g->header.size();
  for (auto& pc : pieces) {
    g->bits.insert(g->bits.end(), pc.bits.begin(), pc.bits.end());
    g->meta_len.insert(g->meta_len.end(), pc.meta_len.begin(), pc.meta_len.end());
    g->meta.append(pc.meta);
    g->rows += pc.rows;
    if (pc.bad_cells >= 0) {
      g->err = (range_start ? "data row " + std::to_string(g->rows + 1) + " of the byte range starting at offset " +
                                  std::to_string(range_start)
                            : "row " + std::to_string(g->rows + 2)) +
               " has " + std::to_string(pc.bad_cells) + " cells, header has " + std::to_string(ncols);
      return -5;
    }
  }
  return 0;
}

std::vector<int32_t> begin_parse(scoary_gpa* g, const uint8_t* keep) {
  const int64_t nstr = (int64_t)g->header.size() - g->startcol;
  std::vector<int32_t> slot(nstr, -1);  // strain column -> bit index
  int64_t kept = 0;
  for (int64_t c = 0; c < nstr; ++c)
    if (!keep || keep[c]) slot[c] = (int32_t)kept++;
  g->strains = kept;
  g->words = (kept + 63) / 64;
  g->rows = 0;
  g->bits.clear();
  g->meta_len.clear();
  g->meta.clear();
  return slot;
}
}  // namespace

int scoary_gpa_parse_mt(scoary_gpa_t g, const uint8_t* keep, int64_t threads, int64_t min_chunk) {
  if (!g || !g->data) return -1;
  std::vector<int32_t> slot = begin_parse(g, keep);
  const size_t body = g->pos;
  std::vector<Piece> pieces;
  // a range must stop exactly where the next one started (or at its own bad row);
  // otherwise a guessed cut was inside a quoted cell: parse in one piece instead
  if (!parse_span(g, slot, body, g->size, threads, min_chunk, &pieces)) {
    pieces.assign(1, Piece());
    parse_range(g, slot, body, g->size, &pieces[0]);
  }
  return finish_pieces(g, pieces);
}

int scoary_gpa_parse_part(scoary_gpa_t g, const uint8_t* keep, int64_t part, int64_t nparts,
                          int64_t threads, int64_t min_chunk) {
  if (!g || !g->data || nparts < 1 || part < 0 || part >= nparts) return -1;
  std::vector<int32_t> slot = begin_parse(g, keep);
  const size_t body = g->pos;
  const std::vector<size_t> outer = guess_cuts(g, body, g->size, nparts);
  const int64_t have = (int64_t)outer.size() - 1;        // small files give fewer parts
  if (part >= have) return 0;                            // nothing for this part
  std::vector<Piece> pieces;
  if (!parse_span(g, slot, outer[part], outer[part + 1], threads, min_chunk, &pieces)) {
    g->err = "a part boundary lies inside a quoted cell: parse the file in one piece";
    return -6;
  }
  return finish_pieces(g, pieces, part > 0 ? outer[part] : 0);
}

int scoary_gpa_parse(scoary_gpa_t g, const uint8_t* keep) {
  int64_t threads = 1;
#ifdef _OPENMP
  threads = omp_get_max_threads();
  if (threads > 32) threads = 32;
#endif
  return scoary_gpa_parse_mt(g, keep, threads, (int64_t)8 << 20);
}

void scoary_gpa_close(scoary_gpa_t g) {
  if (!g) return;
  if (g->data) munmap(const_cast<char*>(g->data), g->size);
  if (g->fd >= 0) ::close(g->fd);
  delete g;
}

const char* scoary_gpa_error(scoary_gpa_t g) { return g ? g->err.c_str() : "null handle"; }

int64_t scoary_gpa_header_cells(scoary_gpa_t g) { return (int64_t)g->header.size(); }
int64_t scoary_gpa_header_bytes(scoary_gpa_t g) {
  int64_t n = 0;
  for (auto& s : g->header) n += (int64_t)s.size();
  return n;
}
void scoary_gpa_header_copy(scoary_gpa_t g, int32_t* lengths, char* bytes) {
  size_t off = 0;
  for (size_t i = 0; i < g->header.size(); ++i) {
    lengths[i] = (int32_t)g->header[i].size();
    std::memcpy(bytes + off, g->header[i].data(), g->header[i].size());
    off += g->header[i].size();
  }
}
int64_t scoary_gpa_rows(scoary_gpa_t g) { return g->rows; }
int64_t scoary_gpa_strains(scoary_gpa_t g) { return g->strains; }
int64_t scoary_gpa_words(scoary_gpa_t g) { return g->words; }
void scoary_gpa_bits_copy(scoary_gpa_t g, uint64_t* out) {
  std::memcpy(out, g->bits.data(), g->bits.size() * sizeof(uint64_t));
}
int64_t scoary_gpa_meta_bytes(scoary_gpa_t g) { return (int64_t)g->meta.size(); }
void scoary_gpa_meta_copy(scoary_gpa_t g, int32_t* lengths, char* bytes) {
  std::memcpy(lengths, g->meta_len.data(), g->meta_len.size() * sizeof(int32_t));
  std::memcpy(bytes, g->meta.data(), g->meta.size());
}

static inline int64_t row_popcount(const uint64_t* r, int64_t W) {
  int64_t n = 0;
  for (int64_t w = 0; w < W; ++w) n += __builtin_popcountll(r[w]);
  return n;
}

// Padded length of every list: a multiple of kListPad entries, and equal for the
// `gpw` genes that share a wavefront (slots w*gpw .. w*gpw+gpw-1 of the
// length-sorted order), so that the kernel's loop count is wave-uniform.
static const int64_t kListPad = 16;         // half a 32-entry step of the kernel
static const int64_t kListStartUnit = 32;   // start[] of the interleaved layout counts in 32-entry units

static void list_plan(const uint64_t* rows64, int64_t G, int64_t N, int64_t gpw,
                      std::vector<int32_t>& len, std::vector<int32_t>& order,
                      std::vector<int32_t>& padded, uint8_t* flipped) {
  const int64_t W = (N + 63) / 64;
  len.assign(G, 0);
  order.assign(G, 0);
  padded.assign(G, 0);
  std::vector<uint8_t> fl(G);
  for (int64_t g = 0; g < G; ++g) {
    const int64_t n1 = row_popcount(rows64 + g * W, W);
    fl[g] = n1 * 2 <= N ? 0 : 1;
    len[g] = (int32_t)(fl[g] ? N - n1 : n1);
    if (flipped) flipped[g] = fl[g];
  }
  std::vector<int64_t> bucket(N + 2, 0);      // counting sort, descending length, stable
  for (int64_t g = 0; g < G; ++g) ++bucket[N - len[g] + 1];
  for (int64_t k = 1; k <= N + 1; ++k) bucket[k] += bucket[k - 1];
  for (int64_t g = 0; g < G; ++g) order[bucket[N - len[g]]++] = (int32_t)g;
  for (int64_t q = 0; q < G; q += gpw) {
    const int64_t L = (len[order[q]] + kListPad - 1) / kListPad * kListPad;  // longest of the group
    for (int64_t k = q; k < G && k < q + gpw; ++k) padded[k] = (int32_t)L;
  }
}

int64_t scoary_lists_count(const uint64_t* rows64, int64_t G, int64_t N, int64_t genes_per_wave,
                           int64_t piece) {
  std::vector<int32_t> len, order, padded;
  list_plan(rows64, G, N, genes_per_wave, len, order, padded, nullptr);
  int64_t total = 0;
  for (int64_t k = 0; k < G; ++k) total += padded[k];
  // interleaved layout: the last wave group is stored as a full group
  if (piece > 0 && G % genes_per_wave) total += (genes_per_wave - G % genes_per_wave) * padded[G - 1];
  return total;
}

void scoary_lists_build(const uint64_t* rows64, int64_t G, int64_t N, int64_t row_stride,
                        int64_t genes_per_wave, int64_t classes, int64_t piece, uint32_t* idx,
                        int32_t* start, int32_t* ngroups, int32_t* order_out, uint8_t* flipped) {
  const int64_t W = (N + 63) / 64;
  std::vector<int32_t> len, order, padded;
  list_plan(rows64, G, N, genes_per_wave, len, order, padded, flipped);
  // first entry of every slot (piece == 0) or of every wave group (piece > 0)
  const int64_t nwg = (G + genes_per_wave - 1) / genes_per_wave;
  std::vector<int64_t> base(piece > 0 ? nwg + 1 : G + 1, 0);
  if (piece > 0)
    for (int64_t q = 0; q < nwg; ++q) base[q + 1] = base[q] + genes_per_wave * padded[q * genes_per_wave];
  else
    for (int64_t k = 0; k < G; ++k) base[k + 1] = base[k] + padded[k];
  const uint32_t zero_row = (uint32_t)(N * row_stride);
  const int64_t cmask = classes - 1;                 // classes is a power of two
#pragma omp parallel
  {
    std::vector<uint32_t> tmp(N + 1), sorted(N + 1), one, grid;   // positions: ascending / by class / as emitted / on the grid
    const uint32_t kNoEntry = 0xffffffffu;
    std::vector<int64_t> first(classes + 1), taken(classes);
#pragma omp for schedule(dynamic, 8)
    for (int64_t q = 0; q < nwg; ++q) {
      for (int64_t j = 0; j < genes_per_wave; ++j) {
        const int64_t k = q * genes_per_wave + j;
        if (k >= G) {                                 // missing genes of the last group: all padding
          if (piece > 0) {
            const int64_t L = padded[q * genes_per_wave];
            for (int64_t pc = 0; pc < L / piece; ++pc)
              for (int64_t x = 0; x < piece; ++x)
                idx[base[q] + (pc * genes_per_wave + j) * piece + x] = zero_row;
          }
          continue;
        }
        const int64_t g = order[k];
        order_out[k] = (int32_t)g;
        ngroups[k] = (int32_t)(padded[k] / kListPad);
        start[k] = (int32_t)(piece > 0 ? base[q] / kListStartUnit : base[k] / kListPad);
        const uint64_t* r = rows64 + g * W;
        const uint64_t inv = flipped[g] ? ~(uint64_t)0 : 0;
        // Spec S6: the genes of one LDS lane group read the label tile in lockstep,
        // and a row's bank range is fixed by (isolate index mod classes).  Position (class c,
        // rank rho within the class, ascending) belongs on grid slot rho * classes + ((c - k) mod
        // classes): entry e of slot k then comes from class (k + e) mod classes and the genes of a
        // group sit on distinct bank slots.  The list has no gaps: positions whose grid slot lies
        // at or beyond the list length fill, in grid order, the holes below it (slots of classes
        // that ran dry), in hole order.  (The device builder k_lists_fill finds the same places
        // in closed form; here the grid is simply built and compacted.)
        std::fill(first.begin(), first.end(), 0);     // counting sort by class
        int64_t nt = 0;
        for (int64_t w = 0; w < W; ++w) {
          uint64_t bits = r[w] ^ inv;
          if (w == W - 1 && (N & 63)) bits &= (((uint64_t)1 << (N & 63)) - 1);
          while (bits) {
            const int b = __builtin_ctzll(bits);
            bits &= bits - 1;
            const uint32_t pos_i = (uint32_t)(w * 64 + b);
            tmp[nt++] = pos_i;
            ++first[(pos_i & cmask) + 1];
          }
        }
        for (int64_t c = 0; c < classes; ++c) first[c + 1] += first[c];
        std::copy(first.begin(), first.begin() + classes, taken.begin());
        for (int64_t i = 0; i < nt; ++i)
          sorted[taken[tmp[i] & cmask]++] = (uint32_t)(tmp[i] * row_stride);
        const int64_t L = padded[k];
        one.assign(L, zero_row);                      // padding -> the zero row
        int64_t maxcnt = 0;
        for (int64_t c = 0; c < classes; ++c) maxcnt = std::max<int64_t>(maxcnt, first[c + 1] - first[c]);
        grid.assign((size_t)(maxcnt * classes), kNoEntry);
        for (int64_t c = 0; c < classes; ++c) {
          const int64_t dc = (c - k) & cmask;
          for (int64_t rho = 0; rho < first[c + 1] - first[c]; ++rho)
            grid[(size_t)(rho * classes + dc)] = sorted[first[c] + rho];
        }
        int64_t hole = 0;                             // next hole below nt
        for (int64_t e = 0; e < (int64_t)grid.size(); ++e) {
          if (grid[(size_t)e] == kNoEntry) continue;
          if (e < nt) {
            one[e] = grid[(size_t)e];
          } else {                                    // overflow -> next hole
            while (grid[(size_t)hole] != kNoEntry) ++hole;
            one[hole++] = grid[(size_t)e];
          }
        }
        if (piece <= 0) {                             // gene-contiguous
          std::memcpy(idx + base[k], one.data(), (size_t)L * sizeof(uint32_t));
        } else {                                      // pieces of the group's genes interleaved
          for (int64_t pc = 0; pc < L / piece; ++pc)
            std::memcpy(idx + base[q] + (pc * genes_per_wave + j) * piece, one.data() + pc * piece,
                        (size_t)piece * sizeof(uint32_t));
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------
// vcf2scoary record lines (scoary/vcf2scoary.py:170-214): the per-variant loop
// ---------------------------------------------------------------------------
namespace {
// Output rows are assembled in one flat buffer; reserve() before every row keeps
// the cell writers free of bounds checks.
struct OutBuf {
  FILE* f;
  std::vector<char> buf;
  size_t len = 0;
  explicit OutBuf(FILE* f_) : f(f_), buf(1 << 23) {}
  bool reserve(size_t need) {
    if (len + need <= buf.size()) return true;
    if (len && fwrite(buf.data(), 1, len, f) != len) return false;
    len = 0;
    if (need > buf.size()) buf.resize(need);
    return true;
  }
  inline void put(char c) { buf[len++] = c; }
  inline void cell(const char* p, size_t n, bool first) {
    if (!first) put(',');
    put('"');
    memcpy(buf.data() + len, p, n);
    len += n;
    put('"');
  }
  bool finish() { return !len || fwrite(buf.data(), 1, len, f) == len; }
};
inline bool is_word(char c) {
  return (c >= '0' && c <= '9') || (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c == '_';
}
}  // namespace

int64_t scoary_vcf_convert(const char* vcf_path, int64_t offset, const char* out_path,
                           const char* types) {
  const int fd = open(vcf_path, O_RDONLY);
  if (fd < 0) return -1;
  struct stat st;
  if (fstat(fd, &st) != 0 || offset < 0 || offset > st.st_size) {
    close(fd);
    return -1;
  }
  const size_t n = (size_t)st.st_size;
  const char* d = n ? (const char*)mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0) : "";
  if (n && d == MAP_FAILED) {
    close(fd);
    return -1;
  }
  FILE* out = fopen(out_path, "ab");
  if (!out) {
    if (n) munmap((void*)d, n);
    close(fd);
    return -1;
  }
  std::vector<std::string> want;                    // wanted TYPE= values
  if (types) {
    const char* t = types;
    for (;;) {
      const char* c = strchr(t, ',');
      want.emplace_back(t, c ? (size_t)(c - t) : strlen(t));
      if (!c) break;
      t = c + 1;
    }
  }
  OutBuf ob(out);
  std::vector<int64_t> gt;                          // genotypes of a multi-allelic line (-1 = ".")
  int64_t rows = 0, rc = 0;
  size_t p = (size_t)offset;
  while (p < n && rc == 0) {
    const char* nl = (const char*)memchr(d + p, '\n', n - p);
    size_t end = nl ? (size_t)(nl - d) : n;
    const size_t next = nl ? end + 1 : n;
    if (end > p && d[end - 1] == '\r') --end;
    const char* L = d + p;
    const size_t len = end - p;
    p = next;
    // anything the plain split does not cover goes back to the Python reader:
    // quote characters, a lone carriage return, fewer than ten columns
    if (memchr(L, '"', len) || memchr(L, '\r', len)) { rc = -2; break; }
    size_t fs[10], fe[9];                            // the nine fixed fields; fs[9] = first sample
    size_t q = 0;
    int nf = 0;
    fs[0] = 0;
    while (nf < 9) {
      const char* tab = (const char*)memchr(L + q, '\t', len - q);
      if (!tab) break;
      fe[nf] = (size_t)(tab - L);
      q = fe[nf] + 1;
      fs[++nf] = q;
    }
    if (nf < 9) { rc = -2; break; }
    if (types) {                                    // re.search(r"TYPE=(\w+)", INFO).group(1) in types
      const char* info = L + fs[7];
      const size_t il = fe[7] - fs[7];
      bool found = false, keep = false;
      for (size_t i = 0; i + 6 <= il && !found; ++i) {
        if (memcmp(info + i, "TYPE=", 5) == 0 && is_word(info[i + 5])) {
          size_t e = i + 5;
          while (e < il && is_word(info[e])) ++e;
          found = true;
          for (const auto& w : want)
            if (w.size() == e - (i + 5) && memcmp(w.data(), info + i + 5, w.size()) == 0) keep = true;
        }
      }
      if (!found) { rc = -2; break; }               // Python raises here
      if (!keep) continue;
    }
    const char* alt = L + fs[4];
    const size_t al = fe[4] - fs[4];
    if (!memchr(alt, ',', al)) {
      if (!ob.reserve(3 * len + 64)) { rc = -1; break; }
      for (int c = 0; c < 9; ++c) ob.cell(L + fs[c], fe[c] - fs[c], c == 0);
      ob.cell("False", 5, false);
      for (size_t i = fs[9];;) {                     // first ':'-field of every sample cell
        ob.put(',');
        ob.put('"');
        while (i < len && L[i] != '\t' && L[i] != ':') ob.put(L[i++]);
        ob.put('"');
        while (i < len && L[i] != '\t') ++i;
        if (i >= len) break;
        ++i;
      }
      ob.put('\n');
      ++rows;
      continue;
    }
    gt.clear();
    for (size_t i = fs[9];;) {
      int64_t v = 0;
      size_t digits = 0;
      bool dot = false;
      while (i < len && L[i] != '\t' && L[i] != ':') {
        const char ch = L[i++];
        if (ch >= '0' && ch <= '9') { v = v * 10 + (ch - '0'); ++digits; }
        else if (ch == '.' && digits == 0 && !dot) dot = true;
        else rc = -2;                                // int() is laxer: the Python path decides
      }
      if (dot ? digits != 0 : (digits == 0 || digits > 9)) rc = -2;
      gt.push_back(dot ? -1 : v);
      while (i < len && L[i] != '\t') ++i;
      if (i >= len) break;
      ++i;
    }
    if (rc) break;
    int64_t k = 0;
    for (size_t a0 = 0;;) {
      const char* c = (const char*)memchr(alt + a0, ',', al - a0);
      const size_t a1 = c ? (size_t)(c - alt) : al;
      ++k;
      if (!ob.reserve(len + 4 * gt.size() + 64)) { rc = -1; break; }
      for (int cidx = 0; cidx < 4; ++cidx) ob.cell(L + fs[cidx], fe[cidx] - fs[cidx], cidx == 0);
      ob.cell(alt + a0, a1 - a0, false);
      for (int cidx = 5; cidx < 9; ++cidx) ob.cell(L + fs[cidx], fe[cidx] - fs[cidx], false);
      ob.cell("True", 4, false);
      for (const int64_t g : gt) {
        ob.put(',');
        ob.put('"');
        ob.put(g == k ? '1' : '0');
        ob.put('"');
      }
      ob.put('\n');
      ++rows;
      if (!c) break;
      a0 = a1 + 1;
    }
  }
  if (rc == 0 && !ob.finish()) rc = -1;
  if (fclose(out) != 0 && rc == 0) rc = -1;
  if (n) munmap((void*)d, n);
  close(fd);
  return rc ? rc : rows;
}

}  // extern "C"

// ---- UPGMA merge order (scoary/methods.py:640-707, scoary/classes.py:68-196) ---------
// The reference keeps the distance matrix in a quad tree of 2x2 block minima and takes,
// descending from the top, the smallest (value, i, j) of each block -- that order decides
// ties, so the levels are kept exactly as the reference has them (matrix padded to even
// size with BIG at every level, both triangles stored).
namespace {
struct MinQuadTree {
  static constexpr double BIG = 9223372036854775807.0;     // float(sys.maxsize)
  std::vector<std::vector<double>> lv;
  std::vector<int64_t> dim;
  std::vector<double> v, pair;
  MinQuadTree(const double* D, int64_t n) {
    int64_t m = n + n % 2;
    lv.emplace_back((size_t)(m * m), BIG);
    dim.push_back(m);
    for (int64_t r = 0; r < n; ++r) std::memcpy(&lv[0][(size_t)(r * m)], D + r * n, (size_t)n * sizeof(double));
    while (m > 2) {
      const int64_t h = m / 2, mm = h + h % 2;
      std::vector<double> nxt((size_t)(mm * mm), BIG);
      const std::vector<double>& cur = lv.back();
      for (int64_t r = 0; r < h; ++r)
        for (int64_t c = 0; c < h; ++c) {
          const double a = cur[(size_t)(2 * r * m + 2 * c)], b = cur[(size_t)(2 * r * m + 2 * c + 1)];
          const double e = cur[(size_t)((2 * r + 1) * m + 2 * c)], f = cur[(size_t)((2 * r + 1) * m + 2 * c + 1)];
          const double x = a < b ? a : b, y = e < f ? e : f;
          nxt[(size_t)(r * mm + c)] = x < y ? x : y;
        }
      lv.push_back(std::move(nxt));
      dim.push_back(mm);
      m = mm;
    }
    v.resize((size_t)dim[0]);
    pair.resize((size_t)dim[0]);
  }
  // row (or column) idx of level 0 := vec[0..n), then the block minima above it
  void set(int64_t idx, const double* vec, int64_t n, bool row) {
    int64_t len = n;
    std::memcpy(v.data(), vec, (size_t)n * sizeof(double));
    for (size_t l = 0; l < lv.size(); ++l) {
      const int64_t m = dim[l];
      std::vector<double>& cur = lv[l];
      for (int64_t k = len; k < m; ++k) v[(size_t)k] = BIG;
      const int64_t lo = idx & ~(int64_t)1, hi = idx | 1;
      if (row) {
        std::memcpy(&cur[(size_t)(idx * m)], v.data(), (size_t)m * sizeof(double));
        const double *a = &cur[(size_t)(lo * m)], *b = &cur[(size_t)(hi * m)];
        for (int64_t k = 0; k < m; ++k) pair[(size_t)k] = a[k] < b[k] ? a[k] : b[k];
      } else {
        // column lo/hi share a cache line per row; one strided pass, prefetched (the matrix is
        // far larger than the caches and this pass is the whole cost of a merge)
        for (int64_t k = 0; k < m; ++k) {
          if (k + 24 < m) __builtin_prefetch(&cur[(size_t)((k + 24) * m + lo)], 1);
          cur[(size_t)(k * m + idx)] = v[(size_t)k];
          const double a = cur[(size_t)(k * m + lo)], b = cur[(size_t)(k * m + hi)];
          pair[(size_t)k] = a < b ? a : b;
        }
      }
      len = m / 2;
      for (int64_t k = 0; k < len; ++k)
        v[(size_t)k] = pair[(size_t)(2 * k)] < pair[(size_t)(2 * k + 1)] ? pair[(size_t)(2 * k)] : pair[(size_t)(2 * k + 1)];
      idx /= 2;
    }
  }
  void argmin(int64_t& oi, int64_t& oj) const {
    int64_t i = 0, j = 0;
    for (size_t l = lv.size(); l-- > 0;) {
      const int64_t m = dim[l];
      const std::vector<double>& cur = lv[l];
      i *= 2;
      j *= 2;
      int64_t bi = i, bj = j;
      double bv = cur[(size_t)(i * m + j)];
      for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj) {
          if (!di && !dj) continue;
          const double x = cur[(size_t)((i + di) * m + j + dj)];
          if (x < bv) {                       // (value, i, j): candidates come in increasing (i, j)
            bv = x;
            bi = i + di;
            bj = j + dj;
          }
        }
      i = bi;
      j = bj;
    }
    oi = i;
    oj = j;
  }
};
}  // namespace

extern "C" int scoary_upgma_merges(const double* D, int64_t n, int32_t* merges) {
  if (!D || !merges || n < 1) return -1;
  if (n == 1) return 0;
  MinQuadTree qt(D, n);
  std::vector<double> size((size_t)n, 1.0), nd((size_t)n), dead((size_t)n, MinQuadTree::BIG);
  std::vector<char> alive((size_t)n, 1);
  const int64_t m0 = qt.dim[0];
  for (int64_t step = 0; step < n - 1; ++step) {
    int64_t i, j;
    qt.argmin(i, j);
    if (i >= n || j >= n || i == j || !alive[(size_t)i] || !alive[(size_t)j]) return -2;
    merges[2 * step] = (int32_t)i;
    merges[2 * step + 1] = (int32_t)j;
    const double si = size[(size_t)i], sj = size[(size_t)j], ns = si + sj;
    const double *ri = &qt.lv[0][(size_t)(i * m0)], *rj = &qt.lv[0][(size_t)(j * m0)];
    for (int64_t k = 0; k < n; ++k) {
      const double a = ri[k] * si, b = rj[k] * sj;      // no contraction: as numpy evaluates it
      const double sum = a + b;
      nd[(size_t)k] = alive[(size_t)k] ? sum / ns : 1.0;
    }
    nd[(size_t)i] = MinQuadTree::BIG;
    qt.set(i, nd.data(), n, true);
    qt.set(i, nd.data(), n, false);
    qt.set(j, dead.data(), n, true);
    qt.set(j, dead.data(), n, false);
    alive[(size_t)j] = 0;
    size[(size_t)i] = ns;
    size[(size_t)j] = 0.0;
  }
  return 0;
}


// ---- a9: results file writer (include/scoary_io.h) -----------------------------------------------
namespace {
// Python's repr(float): shortest round-trip digits (std::to_chars, scientific, no precision = shortest;
// the same digit string as the interpreter's dtoa mode 0) laid out by float_repr_style 'short':
// decpt = exponent + 1; exponent notation iff decpt <= -4 or decpt > 16 (CPython format_float_short, 'r').
inline int format_repr(double x, char* out) {
  if (std::isnan(x)) { std::memcpy(out, "nan", 3); return 3; }
  if (std::isinf(x)) {
    if (x < 0) { std::memcpy(out, "-inf", 4); return 4; }
    std::memcpy(out, "inf", 3);
    return 3;
  }
  char* o = out;
  if (std::signbit(x)) { *o++ = '-'; x = -x; }
  if (x == 0.0) { std::memcpy(o, "0.0", 3); return (int)(o - out) + 3; }
  char sci[40];
  const auto r = std::to_chars(sci, sci + sizeof sci, x, std::chars_format::scientific);
  // sci = d[.ddd]e[+-]XX
  char digits[24];
  int nd = 0;
  const char* p = sci;
  digits[nd++] = *p++;
  if (*p == '.') {
    ++p;
    while (*p != 'e') digits[nd++] = *p++;
  }
  ++p;                                               // 'e'
  int ex = 0;
  const bool neg = *p == '-';
  ++p;                                               // sign (always present)
  while (p < r.ptr) ex = ex * 10 + (*p++ - '0');
  if (neg) ex = -ex;
  const int decpt = ex + 1;
  if (decpt <= -4 || decpt > 16) {                   // exponent notation
    *o++ = digits[0];
    if (nd > 1) {
      *o++ = '.';
      std::memcpy(o, digits + 1, nd - 1);
      o += nd - 1;
    }
    *o++ = 'e';
    int e = decpt - 1;
    *o++ = e < 0 ? '-' : '+';
    if (e < 0) e = -e;
    char eb[8];
    int ne = 0;
    do { eb[ne++] = (char)('0' + e % 10); e /= 10; } while (e);
    if (ne < 2) eb[ne++] = '0';
    while (ne) *o++ = eb[--ne];
  } else if (decpt <= 0) {                           // 0.000ddd
    *o++ = '0';
    *o++ = '.';
    for (int k = 0; k < -decpt; ++k) *o++ = '0';
    std::memcpy(o, digits, nd);
    o += nd;
  } else if (decpt >= nd) {                          // ddd000.0
    std::memcpy(o, digits, nd);
    o += nd;
    for (int k = nd; k < decpt; ++k) *o++ = '0';
    *o++ = '.';
    *o++ = '0';
  } else {                                           // dd.ddd
    std::memcpy(o, digits, decpt);
    o += decpt;
    *o++ = '.';
    std::memcpy(o, digits + decpt, nd - decpt);
    o += nd - decpt;
  }
  return (int)(o - out);
}
inline int format_int(int64_t v, char* out) {
  const auto r = std::to_chars(out, out + 24, v);
  return (int)(r.ptr - out);
}
}  // namespace

extern "C" int32_t scoary_format_float_repr(double x, char* buf) { return buf ? format_repr(x, buf) : -2; }

extern "C" int64_t scoary_results_write(const char* path, char delimiter, const char* header,
                                        int64_t header_len, int64_t nrows, int32_t ntext,
                                        const char* const* text_blob, const int64_t* const* text_off,
                                        const int64_t* const* text_row, int32_t nnum,
                                        const int32_t* kind, const void* const* cols,
                                        const int64_t* num_row, int64_t threads) {
  if (!path || !header || header_len < 0 || nrows < 0 || ntext < 0 || nnum < 0 || ntext + nnum < 1 ||
      (ntext && (!text_blob || !text_off || !text_row)) || (nnum && (!kind || !cols)) ||
      (nrows && nnum && !num_row))
    return -2;
  FILE* f = std::fopen(path, "wb");
  if (!f) return -1;
  int64_t written = 0;
  bool ok = std::fwrite(header, 1, (size_t)header_len, f) == (size_t)header_len;
  written += header_len;
  const int64_t kBlock = 4096;                        // rows per formatting task
  const int64_t nblocks = (nrows + kBlock - 1) / kBlock;
  // never more threads than blocks of rows: a 6 000-row file is two blocks, and a team of
  // omp_get_max_threads() (the HOST's CPU count inside a container granted 16) only spins
  const int nth = (int)std::max<int64_t>(1, std::min<int64_t>(threads > 0 ? threads : omp_get_max_threads(),
                                                              std::max<int64_t>(nblocks, 1)));
  const int64_t kWindow = (int64_t)nth * 4;           // blocks formatted before they are written out
  std::vector<std::string> buf((size_t)std::min<int64_t>(kWindow, std::max<int64_t>(nblocks, 1)));
  for (int64_t b0 = 0; b0 < nblocks && ok; b0 += kWindow) {
    const int64_t b1 = std::min(nblocks, b0 + kWindow);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nth)
    for (int64_t b = b0; b < b1; ++b) {
      std::string& s = buf[(size_t)(b - b0)];
      s.clear();
      char tmp[40];
      const int64_t r1 = std::min(nrows, (b + 1) * kBlock);
      for (int64_t r = b * kBlock; r < r1; ++r) {
        bool first = true;
        for (int c = 0; c < ntext; ++c) {
          if (!first) s.push_back(delimiter);
          first = false;
          const int64_t k = text_row[c][r];
          const int64_t a = text_off[c][k], e = text_off[c][k + 1];
          s.push_back('"');
          s.append(text_blob[c] + a, (size_t)(e - a));
          s.push_back('"');
        }
        const int64_t nr = nnum ? num_row[r] : 0;
        for (int c = 0; c < nnum; ++c) {
          if (!first) s.push_back(delimiter);
          first = false;
          const int n = kind[c] == 0 ? format_int(static_cast<const int64_t*>(cols[c])[nr], tmp)
                                     : format_repr(static_cast<const double*>(cols[c])[nr], tmp);
          s.push_back('"');
          s.append(tmp, (size_t)n);
          s.push_back('"');
        }
        s.push_back('\n');
      }
    }
    for (int64_t b = b0; b < b1 && ok; ++b) {
      const std::string& s = buf[(size_t)(b - b0)];
      ok = std::fwrite(s.data(), 1, s.size(), f) == s.size();
      written += (int64_t)s.size();
    }
  }
  if (std::fclose(f) != 0) ok = false;
  return ok ? written : -1;
}
