"""Randomised differential corpus against the REAL reference (tests/golden/fuzz_corpus.json.gz, written by
tests/golden/make_fuzz.py in the build container): 160 small inputs with awkward shapes -- plain and Roary tables,
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
    with gzip.open(os.path.join(GOLDEN, "fuzz_corpus.json.gz"), "rt") as f:
        return json.load(f)


CORPUS = load_corpus()
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


def compare_csv(got, want, delimiter):
    """None if equal (bytes, or same rows in the same order with float cells within FLOAT_TOL relative /
    absolute), else a description of the first difference."""
    if got == want:
        return None
    g = list(csv.reader(io.StringIO(got), delimiter=delimiter))
    w = list(csv.reader(io.StringIO(want), delimiter=delimiter))
    if g[0] != w[0]:
        return "header %r != %r" % (g[0], w[0])
    if len(g) != len(w):
        return "rows %d != %d" % (len(g) - 1, len(w) - 1)
    # rows may only change places where the reference's sort key (the p-value column order) ties to 1e-12:
    # compare as sets keyed by the first cell first, then the order of the keys that are not tied
    gi = {}
    for r in g[1:]:
        gi.setdefault(r[0], []).append(r)
    for wr in w[1:]:
        cand = gi.get(wr[0])
        if not cand:
            return "row %r missing" % wr[0]
        gr = cand.pop(0)
        if len(gr) != len(wr):
            return "row %r: %d cells != %d" % (wr[0], len(gr), len(wr))
        for k, (a, b) in enumerate(zip(gr, wr)):
            if a == b:
                continue
            if not (_is_float(a) and _is_float(b)):
                return "row %r column %s: %r != %r" % (wr[0], g[0][k], a, b)
            fa, fb = float(a), float(b)
            if not abs(fa - fb) <= FLOAT_TOL + FLOAT_TOL * abs(fb):
                return "row %r column %s: %r != %r" % (wr[0], g[0][k], a, b)
    if [r[0] for r in g] != [r[0] for r in w]:
        # an order difference is only legitimate between rows whose float cells agree to the tolerance
        # (a sort key tied in one implementation and 1 ulp apart in the other)
        wrow = {}
        for r in w[1:]:
            wrow.setdefault(r[0], r)
        for a, b in zip(g[1:], w[1:]):
            if a[0] == b[0]:
                continue
            ra, rb = wrow[a[0]], b
            keys = [k for k in range(len(ra)) if _is_float(ra[k]) and _is_float(rb[k]) and g[0][k].endswith("_p")]
            if not keys or any(abs(float(ra[k]) - float(rb[k])) > FLOAT_TOL * max(1.0, abs(float(rb[k])))
                               for k in keys[:1]):
                return "row order differs at %r / %r" % (a[0], b[0])
    return None


def compare_case(case, got):
    """-> list of differences between our run and the reference's record of the case."""
    ref = case["ref"]
    diffs = []
    if got["status"] != ref["status"]:
        return ["status %s (%s) != reference %s (%s)" % (got["status"], got["message"], ref["status"], ref["message"])]
    if ref["status"] == "exit":
        if got["message"] != ref["message"]:
            diffs.append("exit message %r != %r" % (got["message"], ref["message"]))
        return diffs
    if sorted(got["files"]) != sorted(ref["files"]):
        return ["result files %s != %s" % (sorted(got["files"]), sorted(ref["files"]))]
    delimiter = ";" if "--delimiter" in case["argv"] else ","
    for fn in sorted(ref["files"]):
        d = compare_csv(got["files"][fn], ref["files"][fn], delimiter)
        if d:
            diffs.append("%s: %s" % (fn, d))
    if ref["tree"] is not None and got["tree"] != ref["tree"]:
        diffs.append("Tree.nwk differs")
    if ref["reduced"] is not None and got["reduced"] != ref["reduced"]:
        diffs.append("gene_presence_absence_reduced.csv differs")
    return diffs


def test_corpus_is_what_the_generator_promises():
    assert CORPUS["kept"] == len(CORPUS["cases"]) >= 150
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
