"""ctypes binding of the native gene presence/absence reader (include/scoary_io.h,
scoary_amd/csrc/scoary_io.cpp).  Host-only input codec; the Python
csv path in methods.Csv_to_dic_Roary stays as the reference-shaped fallback for
handles that are not plain files."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SCOARY_IO_LIB: another build of the same source (tools/sanitize_io.sh: AddressSanitizer / UBSan on the CPU)
LIB_PATH = os.environ.get("SCOARY_IO_LIB") or os.path.join(_HERE, "csrc", "libscoary_io.so")
_lib = None


def available():
    return os.path.exists(LIB_PATH)


def _load():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(LIB_PATH)
        vp, i64 = ctypes.c_void_p, ctypes.c_int64
        L.scoary_gpa_open.argtypes = [ctypes.c_char_p, ctypes.c_char, i64, ctypes.POINTER(vp)]
        L.scoary_gpa_parse.argtypes = [vp, vp]
        L.scoary_gpa_parse_mt.argtypes = [vp, vp, i64, i64]
        L.scoary_gpa_close.argtypes = [vp]
        L.scoary_gpa_close.restype = None
        L.scoary_gpa_error.argtypes = [vp]
        L.scoary_gpa_error.restype = ctypes.c_char_p
        for name in ("header_cells", "header_bytes", "rows", "strains", "words", "meta_bytes"):
            f = getattr(L, "scoary_gpa_" + name)
            f.argtypes = [vp]
            f.restype = i64
        for name in ("header_copy", "meta_copy"):
            f = getattr(L, "scoary_gpa_" + name)
            f.argtypes = [vp, vp, vp]
            f.restype = None
        L.scoary_gpa_bits_copy.argtypes = [vp, vp]
        L.scoary_gpa_bits_copy.restype = None
        _lib = L
    return _lib


def vcf_convert(vcf_path, offset, out_path, types=None):
    """Variant lines of a VCF -> appended table rows (scoary_vcf_convert).
    Returns rows written, -1 on I/O errors, -2 if the file needs the Python loop."""
    L = _load()
    L.scoary_vcf_convert.argtypes = [ctypes.c_char_p, ctypes.c_int64, ctypes.c_char_p,
                                     ctypes.c_char_p]
    L.scoary_vcf_convert.restype = ctypes.c_int64
    t = None if types is None else ",".join(types).encode()
    return int(L.scoary_vcf_convert(os.fsencode(vcf_path), int(offset), os.fsencode(out_path), t))


def format_float_repr(x):
    """repr(float) as the native results writer prints it (scoary_format_float_repr)."""
    L = _load()
    L.scoary_format_float_repr.argtypes = [ctypes.c_double, ctypes.c_char_p]
    L.scoary_format_float_repr.restype = ctypes.c_int32
    buf = ctypes.create_string_buffer(40)
    n = L.scoary_format_float_repr(float(x), buf)
    return buf.raw[:n].decode()


class TextColumn:
    """A string table for scoary_results_write: the UTF-8 bytes of the strings back to back and
    their offsets.  Built once per gene table column (or per --collapse name list)."""

    def __init__(self, strings):
        enc = [str(x).encode("utf-8") for x in strings]
        self.blob = b"".join(enc)
        self.off = np.zeros(len(enc) + 1, dtype=np.int64)
        if enc:
            np.cumsum(np.fromiter((len(e) for e in enc), dtype=np.int64, count=len(enc)), out=self.off[1:])

    def __len__(self):
        return len(self.off) - 1


def results_write(path, delimiter, header_line, text_cols, text_rows, num_cols, num_row, threads=0):
    """Write a results file natively (scoary_results_write, include/scoary_io.h).
    text_cols: TextColumn per text column; text_rows: int64 array per text column (string index per
    written row); num_cols: numpy arrays, integer dtypes printed as decimals, floats as repr(float);
    num_row: int64 array, the index into the numeric columns per written row.  Returns bytes written."""
    L = _load()
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32
    L.scoary_results_write.argtypes = [ctypes.c_char_p, ctypes.c_char, ctypes.c_char_p, i64, i64, i32, vp, vp, vp,
                                       i32, vp, vp, vp, i64]
    L.scoary_results_write.restype = i64
    nrows = int(len(num_row))
    keep = []                                          # arrays that must stay alive across the call

    def as_i64(a):
        a = np.ascontiguousarray(a, dtype=np.int64)
        keep.append(a)
        return a.ctypes.data
    ntext = len(text_cols)
    blobs = (ctypes.c_char_p * max(ntext, 1))(*[c.blob for c in text_cols])
    offs = (vp * max(ntext, 1))(*[as_i64(c.off) for c in text_cols])
    trows = (vp * max(ntext, 1))(*[as_i64(r) for r in text_rows])
    for c, r in zip(text_cols, text_rows):
        if len(r) != nrows or (nrows and (np.min(r) < 0 or np.max(r) >= len(c))):
            raise ValueError("results_write: text row index out of range")
    kinds, ptrs = [], []
    for col in num_cols:
        col = np.asarray(col)
        if nrows and (np.min(num_row) < 0 or np.max(num_row) >= len(col)):
            raise ValueError("results_write: numeric row index out of range")
        if np.issubdtype(col.dtype, np.integer) or col.dtype == np.bool_:
            kinds.append(0)
            ptrs.append(as_i64(col))
        else:
            kinds.append(1)
            a = np.ascontiguousarray(col, dtype=np.float64)
            keep.append(a)
            ptrs.append(a.ctypes.data)
    nnum = len(num_cols)
    kind_arr = np.asarray(kinds or [0], dtype=np.int32)
    col_arr = (vp * max(nnum, 1))(*ptrs)
    hdr = header_line.encode("utf-8")
    n = L.scoary_results_write(os.fsencode(path), delimiter.encode("utf-8"), hdr, len(hdr), nrows, ntext,
                               ctypes.cast(blobs, vp), ctypes.cast(offs, vp), ctypes.cast(trows, vp), nnum,
                               kind_arr.ctypes.data, ctypes.cast(col_arr, vp), as_i64(num_row), int(threads))
    if n < 0:
        raise OSError("scoary_results_write(%s) failed: %d" % (path, n))
    return int(n)


def upgma_merges(D):
    """The reference's UPGMA merge loop on an (n, n) float64 distance matrix whose
    diagonal is already 1 (scoary_upgma_merges): (n-1, 2) int32 merged index pairs."""
    L = _load()
    D = np.ascontiguousarray(D, dtype=np.float64)
    n = D.shape[0]
    merges = np.zeros((max(n - 1, 0), 2), dtype=np.int32)
    L.scoary_upgma_merges.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    L.scoary_upgma_merges.restype = ctypes.c_int
    rc = L.scoary_upgma_merges(D.ctypes.data_as(ctypes.c_void_p), n,
                               merges.ctypes.data_as(ctypes.c_void_p))
    if rc != 0:
        raise RuntimeError("scoary_upgma_merges failed: %d" % rc)
    return merges


def build_lists(rows64, N, row_stride, genes_per_wave, classes, piece=0):
    """Minority index lists of every gene row (include/scoary_io.h,
    scoary_lists_build): dict of numpy arrays idx (uint32), start, ngroups,
    order (int32), flipped (uint8).  The four layout arguments come from
    scoary_list_params(N)."""
    L = _load()
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    L.scoary_lists_count.argtypes = [vp, i64, i64, i64, i64]
    L.scoary_lists_count.restype = i64
    L.scoary_lists_build.argtypes = [vp, i64, i64, i64, i64, i64, i64, vp, vp, vp, vp, vp]
    L.scoary_lists_build.restype = None
    rows64 = np.ascontiguousarray(rows64, dtype=np.uint64)
    G = rows64.shape[0]
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)        # noqa: E731
    total = L.scoary_lists_count(p(rows64), G, int(N), int(genes_per_wave), int(piece))
    # + 64 entries of slack: the kernel prefetches index vectors unconditionally
    out = {"idx": np.zeros(total + 64, dtype=np.uint32),
           "start": np.zeros(G, dtype=np.int32), "ngroups": np.zeros(G, dtype=np.int32),
           "order": np.zeros(G, dtype=np.int32), "flipped": np.zeros(G, dtype=np.uint8)}
    L.scoary_lists_build(p(rows64), G, int(N), int(row_stride), int(genes_per_wave), int(classes),
                         int(piece),
                         p(out["idx"]), p(out["start"]), p(out["ngroups"]), p(out["order"]),
                         p(out["flipped"]))
    out["entries"] = int(total)
    return out


def _cells(lengths, blob):
    out, off = [], 0
    for n in lengths:
        out.append(blob[off:off + n].decode("utf-8", errors="surrogateescape"))
        off += n
    return out


class GpaError(Exception):
    pass


class GpaPartBoundary(GpaError):
    """A part boundary of a sharded read lies inside a quoted cell (scoary_gpa_parse_part
    returned -6): every rank has to read the whole file instead."""


def read_gpa(path, delimiter, startcol, allowed=None, threads=None, min_chunk=8 << 20,
             need_cols=None, part=None):
    """-> (header, meta_rows, rows64, kept_strains): the file's header cells,
    for every data row the text of columns [0, startcol), the presence bits of
    the kept strain columns as rows64, and the kept strain names.  ``allowed``:
    None or a container of isolate names (methods.py:416-420, 473-475).
    ``threads`` / ``min_chunk``: parallel body parse (scoary_gpa_parse_mt); the
    default lets the library use its OpenMP thread count.  ``need_cols``: None
    (decode every text cell) or a callable header -> columns < startcol whose
    text is wanted; the other cells of meta_rows are left as "".  ``part`` = (k, n):
    parse only the k-th of n byte ranges of the body (scoary_gpa_parse_part; one rank of n
    under torchrun) -- raises GpaPartBoundary if the ranks must fall back to whole-file
    reads."""
    L = _load()
    h = ctypes.c_void_p()
    rc = L.scoary_gpa_open(os.fsencode(path), delimiter.encode()[0:1], int(startcol),
                           ctypes.byref(h))
    try:
        if rc != 0:
            raise GpaError((L.scoary_gpa_error(h) or b"open failed").decode())
        nh = L.scoary_gpa_header_cells(h)
        lens = np.zeros(nh, dtype=np.int32)
        blob = ctypes.create_string_buffer(max(1, L.scoary_gpa_header_bytes(h)))
        L.scoary_gpa_header_copy(h, lens.ctypes.data_as(ctypes.c_void_p), blob)
        header = _cells(lens, blob.raw)
        strains = header[startcol:]
        keep = None
        if allowed is not None:
            keep = np.array([1 if s in allowed else 0 for s in strains], dtype=np.uint8)
        kp = keep.ctypes.data_as(ctypes.c_void_p) if keep is not None else None
        if part is not None:
            L.scoary_gpa_parse_part.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                                ctypes.c_int64, ctypes.c_int64, ctypes.c_int64]
            L.scoary_gpa_parse_part.restype = ctypes.c_int
            rc = L.scoary_gpa_parse_part(h, kp, int(part[0]), int(part[1]),
                                         int(threads or min(32, os.cpu_count() or 1)),
                                         int(min_chunk))
            if rc == -6:
                raise GpaPartBoundary(L.scoary_gpa_error(h).decode())
        elif threads is None:
            rc = L.scoary_gpa_parse(h, kp)
        else:
            rc = L.scoary_gpa_parse_mt(h, kp, int(threads), int(min_chunk))
        if rc != 0:
            raise GpaError(L.scoary_gpa_error(h).decode())
        R, W = L.scoary_gpa_rows(h), L.scoary_gpa_words(h)
        bits = np.zeros((R, max(W, 0)), dtype=np.uint64)
        if R and W:
            L.scoary_gpa_bits_copy(h, bits.ctypes.data_as(ctypes.c_void_p))
        mlen = np.zeros(R * startcol, dtype=np.int32)
        mblob = ctypes.create_string_buffer(max(1, L.scoary_gpa_meta_bytes(h)))
        if R and startcol:
            L.scoary_gpa_meta_copy(h, mlen.ctypes.data_as(ctypes.c_void_p), mblob)
        if need_cols is None or not R or not startcol:
            flat = _cells(mlen, mblob.raw)
            meta = [flat[r * startcol:(r + 1) * startcol] for r in range(R)]
        else:
            raw = mblob.raw
            ends = np.cumsum(mlen, dtype=np.int64)
            begins = ends - mlen
            meta = [[""] * startcol for _ in range(R)]
            for c in sorted(set(int(c) for c in need_cols(header) if 0 <= c < startcol)):
                b, e = begins[c::startcol].tolist(), ends[c::startcol].tolist()
                for r in range(R):
                    meta[r][c] = raw[b[r]:e[r]].decode("utf-8", errors="surrogateescape")
        kept = [s for k, s in enumerate(strains) if keep is None or keep[k]]
        return header, meta, bits, kept
    finally:
        L.scoary_gpa_close(h)
