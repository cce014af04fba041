#!/usr/bin/env python3
"""Byte-level mutation fuzz of the native gene-table reader (scoary_amd/csrc/scoary_io.cpp through
scoary_amd.io_native.read_gpa) and the native VCF record loop, meant to run under AddressSanitizer /
UBSan (tools/sanitize_io.sh builds the instrumented library and sets SCOARY_IO_LIB).  No GPU.

    python tools/fuzz_reader_mutations.py [cases] [seed]

Every case: a small table text, mutated (quotes, delimiters, line ends of all three kinds, blanks,
truncation, duplicated and dropped spans, very long cells, bytes >= 0x80), parsed with 1 thread and
with several threads and tiny chunks.  What is held:
  * no crash and no sanitizer report (the run itself);
  * the same outcome from the one-thread and the many-thread parse (rows, bits, texts, or an error);
  * where Python's csv module (the reference's tokeniser: csv.reader(..., skipinitialspace=True) on a
    universal-newlines handle, scoary/methods.py:335-343) reads the text into rows that all have at
    least startcol + 1 cells ... the same cells and presence bits.
Rows shorter than the header are an error in both (the reference raises IndexError / exits)."""
import csv
import io
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scoary_amd import io_native  # noqa: E402

ABSENT = ("", "0", "-")


def base_table(rng, delim):
    ns, ng, sc = int(rng.integers(1, 9)), int(rng.integers(0, 7)), int(rng.integers(1, 5))
    rows = [["h%d" % c for c in range(sc)] + ["iso%d" % i for i in range(ns)]]
    for g in range(ng):
        meta = ["g%d" % g] + [rng.choice(["", "x", "a b", 'q"q', "c%sd" % delim, "multi\nline", " lead"])
                              for _ in range(sc - 1)]
        cells = [rng.choice(["", "0", "-", "1", "p_1", " 0", "0 ", '"', "--", "00"]) for _ in range(ns)]
        rows.append(meta + cells)
    buf = io.StringIO()
    w = csv.writer(buf, delimiter=delim, lineterminator=rng.choice(["\n", "\r\n"]),
                   quoting=csv.QUOTE_ALL if rng.random() < 0.3 else csv.QUOTE_MINIMAL)
    w.writerows(rows)
    return buf.getvalue().encode(), sc


def mutate(rng, data, delim):
    data = bytearray(data)
    tokens = [b'"', b'""', delim.encode(), b"\n", b"\r", b"\r\n", b" ", b"  ", b"\t", b"0", b"-", b"\xc3\xa9",
              b"\xff", b'",', b',"', b'"\n', b'\n"', b"x" * 300, b"\xef\xbb\xbf"]
    for _ in range(int(rng.integers(0, 5))):
        if not data:
            break
        op = int(rng.integers(0, 6))
        i = int(rng.integers(0, len(data) + 1))
        if op == 0:
            data[i:i] = tokens[int(rng.integers(0, len(tokens)))]
        elif op == 1 and i < len(data):
            del data[i:i + int(rng.integers(1, 6))]
        elif op == 2 and i < len(data):
            data[i] = int(rng.choice(list(b'",;\n\r 0-x')))
        elif op == 3:
            del data[i:]                                         # truncation (maybe inside a quoted cell)
        elif op == 4 and i < len(data):
            j = min(len(data), i + int(rng.integers(1, 40)))
            data[i:i] = data[i:j]                                # duplicated span
        elif op == 5:
            data[i:i] = b"\n" * int(rng.integers(1, 4))          # blank lines
    return bytes(data)


def python_rows(path, delim):
    try:
        with open(path, "r", newline=None, encoding="utf-8", errors="surrogateescape") as f:
            return list(csv.reader(f, skipinitialspace=True, delimiter=delim))
    except csv.Error:
        return None


def native(path, delim, sc, **kw):
    try:
        return "ok", io_native.read_gpa(path, delim, sc, **kw)
    except io_native.GpaError as e:
        return "err", str(e)


def same(a, b):
    if a[0] != b[0]:
        return False
    if a[0] == "err":
        return True                    # (messages name the same record; checked in the unit tests)
    x, y = a[1], b[1]
    return x[0] == y[0] and x[1] == y[1] and x[3] == y[3] and np.array_equal(x[2], y[2])


def vcf_text(rng, V, S):
    nl = "\r\n" if rng.random() < 0.3 else "\n"
    lines = ["##fileformat=VCFv4.2", '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
             '##INFO=<ID=TYPE,Number=A,Type=String,Description="type">',
             "\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"]
                       + ["s%d" % i for i in range(S)])]
    for v in range(V):
        nalt = int(rng.integers(1, 4))
        alt = ",".join("ACGT"[a] for a in range(nalt)) if rng.random() < 0.9 else "A,,T"[:2 * nalt - 1]
        g = rng.integers(0, nalt + 1, S).astype(str)
        g[rng.random(S) < 0.1] = "."
        sub = rng.random() < 0.5
        cells = [x + ":12" if sub else x for x in g]
        info = rng.choice(["DP=5;TYPE=snp", "TYPE=ins;AF=0.5", "TYPE=del", "DP=1", "TYPE=mnp,snp"])
        lines.append("\t".join(["chr1", str(100 + v), ".", "G", alt, "50", "PASS", info, "GT:DP" if sub else "GT"]
                               + cells))
    return (nl.join(lines) + nl).encode()


def vcf_phase(cases, rng, tmp):
    """The native VCF record loop (scoary_vcf_convert) against the Python loop on mutated record blocks: the same
    table, or the native loop hands the file over (-2) and convert_file ends like the Python loop ends."""
    from scoary_amd import vcf2scoary as v
    src, a, b = os.path.join(tmp, "in.vcf"), os.path.join(tmp, "a.csv"), os.path.join(tmp, "b.csv")
    same_out = handed = raised = 0
    bad = []
    for k in range(cases):
        data = vcf_text(rng, int(rng.integers(0, 12)), int(rng.integers(1, 7)))
        head = data.index(b"#CHROM")
        head = data.index(b"\n", head) + 1
        body = bytearray(data[head:])
        tokens = [b"\t", b"\t\t", b".", b",", b":", b"/", b"|", b"\n", b"\r\n", b'"', b" ", b"-1", b"+1", b"9", b"10",
                  b"x", b"TYPE=", b";", b"\xc3\xa9", b"\xff"]
        for _ in range(int(rng.integers(0, 4))):
            i = int(rng.integers(0, len(body) + 1))
            op = int(rng.integers(0, 4))
            if op == 0:
                body[i:i] = tokens[int(rng.integers(0, len(tokens)))]
            elif op == 1 and i < len(body):
                del body[i:i + int(rng.integers(1, 5))]
            elif op == 2:
                del body[i:]
            elif i < len(body):
                body[i] = int(rng.choice(list(b"\t.,:0123/|x\n")))
        with open(src, "wb") as f:
            f.write(data[:head] + bytes(body))
        types = rng.choice(["ALL", "snp", "ins,del"])
        types = "ALL" if types == "ALL" else types.split(",")
        py_exc = nat_exc = None
        try:
            with open(src, "r", newline=None, errors="surrogateescape") as f, open(a, "w", errors="surrogateescape") as o:
                n_py = v.convert(f, o, types, log=lambda *x: None)
        except BaseException as e:
            py_exc = type(e).__name__
        try:
            n_nat = v.convert_file(src, b, types, log=lambda *x: None)
        except BaseException as e:
            nat_exc = type(e).__name__
        if py_exc or nat_exc:
            raised += 1
            # convert_file opens its files with the default error handler: a UnicodeDecodeError there where the
            # surrogateescape handle of this harness got through is the same file being refused
            if (py_exc is None) != (nat_exc is None) and "Unicode" not in str(nat_exc) + str(py_exc):
                bad.append((k, "python %s / convert_file %s" % (py_exc, nat_exc), bytes(body)))
            continue
        ta, tb = open(a, "rb").read(), open(b, "rb").read()
        if n_py != n_nat or ta != tb:
            bad.append((k, "tables differ (%d / %d rows)" % (n_py, n_nat), bytes(body)))
        else:
            same_out += 1
    print("vcf: %d cases: %d identical tables, %d ended in an exception in both loops, %d findings"
          % (cases, same_out, raised, len(bad)))
    for k, why, data in bad[:12]:
        print("  vcf case %d: %s: %r" % (k, why, data[:300]))
    return len(bad)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    tmp = tempfile.mkdtemp(prefix="fuzzmut_")
    path = os.path.join(tmp, "t.csv")
    n_ok = n_err = n_cmp = 0
    bad = []
    for k in range(cases):
        delim = ";" if rng.random() < 0.25 else ","
        data, sc = base_table(rng, delim)
        data = mutate(rng, data, delim)
        with open(path, "wb") as f:
            f.write(data)
        one = native(path, delim, sc, threads=1)
        many = native(path, delim, sc, threads=int(rng.integers(2, 9)), min_chunk=int(rng.integers(1, 64)))
        dflt = native(path, delim, sc)
        if not (same(one, many) and same(one, dflt)):
            bad.append((k, "thread counts disagree", data))
            continue
        if one[0] == "err":
            n_err += 1
        else:
            n_ok += 1
        rows = python_rows(path, delim)
        if rows is None or b"\x00" in data:
            continue
        rows = [r for r in rows]
        if not rows:
            continue
        body = rows[1:]
        width = len(rows[0])
        py_ok = width > sc and all(len(r) >= width for r in body) if body else width > sc
        if one[0] == "ok" and py_ok:
            header, meta, bits, kept = one[1]
            want_bits = np.array([[c not in ABSENT for c in r[sc:width]] for r in body], dtype=np.uint8).reshape(
                len(body), width - sc)
            got = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :width - sc] if len(body) \
                else np.zeros((0, width - sc), dtype=np.uint8)
            if header != rows[0] or meta != [r[:sc] for r in body] or not np.array_equal(got, want_bits):
                bad.append((k, "differs from csv.reader", data))
            n_cmp += 1
        elif one[0] == "ok" and not py_ok:
            # the native reader took a table the reference's loop would fall over (a short row -> IndexError)
            # or the other way round: both are reported, neither is a memory-safety matter
            short = [i for i, r in enumerate(body) if len(r) < width]
            if short and one[1][2].shape[0] > short[0]:
                bad.append((k, "short row %d accepted" % (short[0] + 2), data))
    print("%d cases (seed %d): %d parsed, %d refused, %d compared cell by cell with csv.reader, %d findings"
          % (cases, seed, n_ok, n_err, n_cmp, len(bad)))
    for k, why, data in bad[:12]:
        print("  case %d: %s: %r" % (k, why, data[:400]))
    nv = vcf_phase(cases, rng, tmp)
    return 1 if bad or nv else 0


if __name__ == "__main__":
    sys.exit(main())
