"""Randomised differential corpus against the REAL reference (tests/golden/fuzz_corpus.json.gz, written by
tests/golden/make_fuzz.py in the build container): 200 small inputs with awkward shapes -- plain and Roary tables,
both delimiters, every spelling of absent cells and missing values, repeated identifiers, shuffled / missing
isolates, degenerate traits, -c / -p / -m / --collapse / -r / -w / --include_input_columns / --threads / pairwise
stage -- each run through ``scoary_amd.methods.main`` and compared with what the reference did with the same files
and flags: the result CSVs (bytes, or cell by cell with the float cells inside north_star's 1e-12), the tree file,
the reduced table, or the message the reference refused the input with."""
import csv
import gzip
import io
import json
import os
import sys

import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


def load_corpus():
    # SCOARY_FUZZ_CORPUS: another corpus of the same generator (tools/fuzz_report.py on one-off extra seeds)
    with gzip.open(os.environ.get("SCOARY_FUZZ_CORPUS") or os.path.join(GOLDEN, "fuzz_corpus.json.gz"), "rt") as f:
        return json.load(f)


CORPUS = load_corpus()


def load_args_corpus():
    """The flag-surface corpus (make_fuzz.py's second file): 200 small tables under odd command lines -- values out of
    range, words where numbers belong, repeated and contradictory flags, a -s inside the identifier cells -- and what
    the reference made of each: mostly the message it left with."""
    with gzip.open(os.path.join(GOLDEN, "fuzz_args_corpus.json.gz"), "rt") as f:
        return json.load(f)


ARGS_CORPUS = load_args_corpus()
FLOAT_TOL = 1e-12


def run_case(case, outdir):
    """-> {"status": "ok" | "exit", "message", "files", "tree", "reduced"} of our command line."""
    from scoary_amd import methods as m
    paths = {}
    for key in ("gpa", "traits", "restrict"):
        if case[key] is not None:
            paths[key] = os.path.join(outdir, key + ".csv")
            with open(paths[key], "w", newline="") as f:
                f.write(case[key])
    od = os.path.join(outdir, "out")
    os.mkdir(od)
    argv = ["-g", paths["gpa"], "-t", paths["traits"]] + \
           [paths["restrict"] if a == "RESTRICT" else a for a in case["argv"]] + ["-o", od, "--no-time"]
    old = sys.argv
    sys.argv = ["scoary"] + argv
    status, message = "ok", None
    try:
        try:
            m.main()
        except SystemExit as e:
            if e.code not in (0, None):
                status, message = "exit", str(e.code)
    finally:
        sys.argv = old
    out = {"status": status, "message": message, "files": {}, "tree": None, "reduced": None}
    if status == "ok":
        for fn in sorted(os.listdir(od)):
            p = os.path.join(od, fn)
            if fn.endswith(".results.csv"):
                with open(p, newline="") as f:
                    out["files"][fn] = f.read()
            elif fn == "Tree.nwk":
                with open(p) as f:
                    out["tree"] = f.read()
            elif fn == "gene_presence_absence_reduced.csv":
                with open(p, newline="") as f:
                    out["reduced"] = f.read()
    return out


def _is_float(s):
    try:
        float(s)
        return True
    except ValueError:
        return False


def _pcell(row, header, name="Naive_p"):
    """The cell of column `name`: a collapsed row of a plain table carries extra cells in FRONT of the counts
    (the reference splits the merged identifier at every "_|_", scoary/methods.py:1152-1153), so columns are
    counted from the right end of the row."""
    return row[header.index(name) + len(row) - len(header)]


SCIPY_DIGITS_MAX_N = 104723 if os.environ.get("SCOARY_FISHER_SCIPY", "1") != "0" else 170


METHOD_COLUMN = {"I": "Naive_p", "B": "Bonferroni_p", "BH": "Benjamini_H_p", "PW": "Best_pairwise_comp_p",
                 "EPW": "Worst_pairwise_comp_p"}


def compare_csv(got, want, delimiter, cutoffs=(("Naive_p", 0.05),), max_hits=False, woven=False,
                by_pairs=False):
    """None if equal -- bytes, or the same rows with float cells within FLOAT_TOL (relative / absolute) and the same
    order wherever the reference's sort key distinguishes two rows -- else a description of the first difference.
    Two places where SciPy's last bit decides in the reference and cannot be reproduced (p within
    1e-12, not to the last digit) are accepted for what they are:
      * a row whose p sits within FLOAT_TOL of a cut-off may be kept on one side and dropped on the other;
      * under -m the last rows may be other members of the tie that straddles the cut."""
    if got == want:
        return None
    g = list(csv.reader(io.StringIO(got), delimiter=delimiter))
    w = list(csv.reader(io.StringIO(want), delimiter=delimiter))
    if g[0] != w[0]:
        return "header %r != %r" % (g[0], w[0])
    header = g[0]

    def close(a, b):
        return abs(a - b) <= FLOAT_TOL + FLOAT_TOL * abs(b)

    def key(r):
        return delimiter.join(r[:len(r) - len(header) + 3])        # the identifier cells, whatever their number
    gi, wi = {}, {}
    for r in g[1:]:
        gi.setdefault(key(r), []).append(r)
    for r in w[1:]:
        wi.setdefault(key(r), []).append(r)
    only = [(k, r, "ours") for k in gi for r in gi[k][len(wi.get(k, [])):]] + \
           [(k, r, "reference") for k in wi for r in wi[k][len(gi.get(k, [])):]]
    if only:
        edge = None
        if max_hits and len(w) > 1:
            edge = max(float(_pcell(r, header)) for r in w[1:])       # the p of the tie the -m cut goes through
        n_cut = 0
        for k, r, side in only:
            # (a capped p of 1.0 never exceeds a cut-off of 1.0: no boundary there)
            at_cut = any(cut < 1.0 and c in header and _is_float(_pcell(r, header, c))
                         and close(float(_pcell(r, header, c)), cut) for c, cut in cutoffs)
            at_edge = edge is not None and close(float(_pcell(r, header)), edge)
            if not (at_cut or at_edge):
                return "row %r only in %s output (p %s)" % (k, side, _pcell(r, header))
            n_cut += at_cut
        if n_cut == 0 and len(g) != len(w):       # the -m tie exchanges rows, it does not add or remove any
            return "rows %d != %d" % (len(g) - 1, len(w) - 1)
    for k in wi:
        for gr, wr in zip(gi.get(k, []), wi[k]):
            if len(gr) != len(wr):
                return "row %r: %d cells != %d" % (k, len(gr), len(wr))
            for j, (a, b) in enumerate(zip(gr, wr)):
                if a == b:
                    continue
                if not (_is_float(a) and _is_float(b) and close(float(a), float(b))):
                    return "row %r cell %d: %r != %r" % (k, j, a, b)
    # order: ours sorted by p like the reference's, and rows that BOTH sides give one and the same p keep the
    # reference's order (its stable sort over the result dictionary -- merged genes move to the end,
    # scoary/methods.py:874-889).  Rows one side separates by an ulp and the other does not are SciPy's last
    # bit again (a gene and its complement get the same double from both sides beyond N = 170, k_fisher).
    if by_pairs:
        # only PW / EPW asked for: the rows come sorted by the pairwise p-values (exact binomial tails: no
        # last-bit noise), scoary/methods.py:1124-1128
        if [key(r) for r in g[1:]] != [key(r) for r in w[1:]]:
            return "rows (sorted by the pairwise p-values) come in another order"
        return None
    ps = [float(_pcell(r, header)) for r in g[1:]]
    if any(ps[i] > ps[i + 1] * (1 + FLOAT_TOL) + FLOAT_TOL for i in range(len(ps) - 1)):
        return "rows are not sorted by Naive_p"
    if woven:
        # pairwise stage on several workers: the order of tied rows follows which worker a row went to, i.e.
        # its position in the FIRST sort, where SciPy's last bit (N <= 170) moves rows between workers
        return None
    common = set(gi) & set(wi)
    pos, ours_p = {}, {}
    for i, r in enumerate(g[1:]):
        pos.setdefault(key(r), i)
        ours_p.setdefault(key(r), _pcell(r, header))
    groups = {}
    for r in w[1:]:
        if key(r) in common:
            groups.setdefault((_pcell(r, header), ours_p[key(r)]), []).append(key(r))
    for (pstr, _), keys in groups.items():
        seq = [pos[k] for k in keys]
        if seq != sorted(seq):
            return "rows with the p %s on both sides come in another order: %r" % (pstr, keys[:6])
    return None


def compare_case(case, got):
    """-> list of differences between our run and the reference's record of the case."""
    ref = case["ref"]
    diffs = []
    if got["status"] != ref["status"]:
        return ["status %s (%s) != reference %s (%s)" % (got["status"], got["message"], ref["status"], ref["message"])]
    if ref["status"] == "exit":
        if "--citation" in case["argv"]:
            return diffs              # both print a citation and leave; the text is this build's own
        if got["message"] != ref["message"]:
            diffs.append("exit message %r != %r" % (got["message"], ref["message"]))
        return diffs
    if sorted(got["files"]) != sorted(ref["files"]):
        return ["result files %s != %s" % (sorted(got["files"]), sorted(ref["files"]))]
    argv = case["argv"]
    delimiter = ";" if "--delimiter" in argv else ","
    methods, values = ["I"], [0.05]
    if "-c" in argv:
        methods = []
        for a in argv[argv.index("-c") + 1:]:
            if a not in METHOD_COLUMN:
                break
            methods.append(a)
    if "-p" in argv:
        values = []
        for a in argv[argv.index("-p") + 1:]:
            if not _is_float(a):
                break
            values.append(float(a))
    if len(values) == 1:
        values = values * len(methods)
    cutoffs = [(METHOD_COLUMN[m_], v) for m_, v in zip(methods, values)]
    for fn in sorted(ref["files"]):
        if got["files"][fn] != ref["files"][fn] and case["N"] <= SCIPY_DIGITS_MAX_N:
            # Up to 170 isolates k_fisher returns SciPy's own double (spec S3), above the command line passes what it
            # prints through scoary_fisher_scipy, and the pairwise stage's binomial cells are SciPy's too
            # (tree.binom_two_sided_many): the file is the reference's, byte for byte.
            diffs.append("%s: not byte-identical" % fn)
        d = compare_csv(got["files"][fn], ref["files"][fn], delimiter, cutoffs, "-m" in argv,
                        woven="--threads" in argv and "--no_pairwise" not in argv,
                        by_pairs="--no_pairwise" not in argv and not {"I", "B", "BH"} & set(methods))
        if d:
            diffs.append("%s: %s" % (fn, d))
    if ref["tree"] is not None and got["tree"] != ref["tree"]:
        diffs.append("Tree.nwk differs")
    if ref["reduced"] is not None and got["reduced"] != ref["reduced"]:
        diffs.append("gene_presence_absence_reduced.csv differs")
    return diffs


def test_corpus_is_what_the_generator_promises():
    assert CORPUS["kept"] == len(CORPUS["cases"]) >= 190
    ok = [c for c in CORPUS["cases"] if c["ref"]["status"] == "ok"]
    assert len(ok) >= 100 and len(CORPUS["cases"]) - len(ok) >= 20
    assert sum("--no_pairwise" not in c["argv"] for c in ok) >= 15        # the pairwise stage is in it
    assert sum(len(c["ref"]["files"]) for c in ok) >= 150


@pytest.mark.parametrize("k", range(len(CORPUS["cases"])), ids=lambda k: "case%03d" % CORPUS["cases"][k]["id"])
def test_fuzz_case_vs_reference(k, tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    case = CORPUS["cases"][k]
    got = run_case(case, str(tmp_path))
    diffs = compare_case(case, got)
    assert not diffs, "case %d %s (N=%d G=%d T=%d roary=%s): %s" % (
        case["id"], " ".join(case["argv"]), case["N"], case["G"], case["T"], case["roary"], "; ".join(diffs))


def test_args_corpus_is_what_the_generator_promises():
    cases = ARGS_CORPUS["cases"]
    assert ARGS_CORPUS["kept"] == len(cases) >= 190
    refused = [c for c in cases if c["ref"]["status"] == "exit"]
    assert len(refused) >= 120 and len(cases) - len(refused) >= 30
    assert len({c["ref"]["message"] for c in refused}) >= 12              # many different ways of being refused


@pytest.mark.parametrize("k", range(len(ARGS_CORPUS["cases"])),
                         ids=lambda k: "args%03d" % ARGS_CORPUS["cases"][k]["id"])
def test_odd_command_line_vs_reference(k, tmp_path):
    """The flag surface: the same outcome as the reference's command line -- its exit message (argparse's status 2
    included), or its files."""
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    case = ARGS_CORPUS["cases"][k]
    got = run_case(case, str(tmp_path))
    diffs = compare_case(case, got)
    assert not diffs, "case %d %s (N=%d G=%d T=%d roary=%s): %s" % (
        case["id"], " ".join(case["argv"]), case["N"], case["G"], case["T"], case["roary"], "; ".join(diffs))
