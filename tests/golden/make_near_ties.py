#!/usr/bin/env python3
"""Near-tie census for the two-sided Fisher rule (spec S3).

Spec S3 counts a support point as "as extreme as observed" when its weight is <= the
observed weight x (1 + 1e-10); SciPy 1.15 uses 1 + 1e-14.  The two rules differ only on a
pair of support points whose weights are DIFFERENT but closer than 1e-10 (relative).  This
script measures how close such pairs get: tests/golden/near_tie_search.c walks every margin
pair of every population size N in the list below (exhaustively, up to the row / column
symmetries) and reports the closest non-equal pairs; each reported pair is then verified with
exact rational arithmetic (fractions.Fraction -- no floating point in the verdict), and the
closest ones per N are written to tests/golden/near_ties.json together with the exact
two-sided p-values under both rules.  tests/test_oracle_golden.py pins the oracle to them,
tests/test_gpu_parity.py the HIP kernel.

    python tests/golden/make_near_ties.py [--sizes 2000,1981,...] [--all-up-to 2100]
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
from fractions import Fraction
from math import comb

HERE = os.path.dirname(os.path.abspath(__file__))
SIZES = [100, 500, 1000, 1980, 1981, 1985, 1990, 1995, 2000, 4950, 5000, 10000]   # the BASELINE populations
BAND = "1e-8"
KEEP = 4            # closest pairs kept per N


def build():
    exe = os.path.join(tempfile.mkdtemp(), "near_tie_search")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", exe,
                           os.path.join(HERE, "near_tie_search.c"), "-lm"])
    return exe


def weight(n1, n2, n, x):
    """hypergeometric pmf numerator (exact integer): C(n1, x) * C(n2, n - x)"""
    return comb(n1, x) * comb(n2, n - x)


def exact_p(n1, n2, n, a, tie):
    """two-sided p of the table with a = top-left count: sum of the weights <= w(a) * (1 + tie)
    over the support, as an exact Fraction; tie is a Fraction (0 for the strict rule)."""
    lo, hi = max(0, n - n2), min(n, n1)
    ws = [weight(n1, n2, n, x) for x in range(lo, hi + 1)]
    thr = ws[a - lo] * (1 + tie)
    return Fraction(sum(w for w in ws if w <= thr), sum(ws))


def tables_for(case):
    """Exact two-sided p of the two tables of a pair (observed = x, observed = y) under the
    strict rule and under the tie windows 1e-14 (SciPy, = spec S3) and 1e-10 (spec S3 of
    round 1)."""
    n1, n2, n = case["n1"], case["n2"], case["n"]
    out = []
    for a in (case["x"], case["y"]):
        p_strict = exact_p(n1, n2, n, a, Fraction(0))
        p_10 = exact_p(n1, n2, n, a, Fraction(1, 10 ** 10))
        p_14 = exact_p(n1, n2, n, a, Fraction(1, 10 ** 14))
        out.append({"a": a, "b": n1 - a, "c": n - a, "d": n2 - n + a,
                    "p_strict": float(p_strict), "p_tie_1e-10": float(p_10),
                    "p_tie_1e-14": float(p_14), "rules_agree": p_10 == p_14})
    return out


def add_tables(out, explicit):
    """Tables for the closest pairs of the explicitly listed sizes (N <= 2100: the big-integer
    sums are cheap) and for EVERY pair inside the (1e-14, 1e-10) band, whatever its N."""
    for case in out["cases"]:
        inside = 1e-14 < abs(case["rel_gap"]) < 1e-10
        if inside or (case["N"] in explicit and case["N"] <= 2100):
            if not case["tables"]:
                case["tables"] = tables_for(case)
    out["pairs_inside_band"] = sum(1 for c in out["cases"] if 1e-14 < abs(c["rel_gap"]) < 1e-10)
    # keep the fixture small: the closest gap per population size for the record, the pairs
    # themselves only where they carry tables
    per_n = dict(out.get("closest_per_N", {}))
    for c in out["cases"]:
        k = str(c["N"])
        if k not in per_n or abs(c["rel_gap"]) < per_n[k]:
            per_n[k] = abs(c["rel_gap"])
    out["closest_per_N"] = per_n
    out["cases"] = [c for c in out["cases"] if c["tables"]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tables-only", action="store_true",
                    help="keep the census of an existing near_ties.json, (re)compute the exact tables")
    ap.add_argument("--sizes", default=",".join(str(s) for s in SIZES))
    ap.add_argument("--all-up-to", type=int, default=0,
                    help="additionally walk EVERY N from 4 to this value (minutes to hours)")
    args = ap.parse_args()
    explicit = set(int(s) for s in args.sizes.split(",") if s)
    if args.tables_only:
        with open(os.path.join(HERE, "near_ties.json")) as f:
            out = json.load(f)
        add_tables(out, explicit)
        with open(os.path.join(HERE, "near_ties.json"), "w") as f:
            json.dump(out, f, indent=1)
        print("tables:", sum(len(c["tables"]) for c in out["cases"]), "pairs inside the band:",
              out["pairs_inside_band"])
        return 0
    sizes = sorted(set(list(explicit) +
                       list(range(4, args.all_up_to + 1))))
    exe = build()
    out = {"_doc": "closest non-equal hypergeometric weight pairs on opposite sides of the mode, "
                   "exhaustive over all margins of each N (tests/golden/make_near_ties.py); "
                   "rel_gap = w(y)/w(x) - 1 exactly (as a float of the exact rational)",
           "band": float(BAND), "sizes_walked": sizes, "cases": []}
    overall = None
    for N in sizes:
        txt = subprocess.check_output([exe, str(N), BAND], text=True)
        rows = []
        for line in txt.splitlines():
            n1, n2, n, x, y, d = line.split()
            rows.append((abs(float(d)), int(n1), int(n2), int(n), int(x), int(y)))
        rows.sort()
        kept = 0
        for d, n1, n2, n, x, y in rows:
            wx, wy = weight(n1, n2, n, x), weight(n1, n2, n, y)
            if wx == wy:
                continue                      # an exact tie after all (never seen: the tool skips |d| < 1e-15)
            gap = Fraction(wy, wx) - 1
            g = float(gap)
            if overall is None or abs(g) < abs(overall[0]):
                overall = (g, N, n1, n2, n, x, y)
            if kept >= KEEP:
                continue
            kept += 1
            case = {"N": N, "n1": n1, "n2": n2, "n": n, "x": x, "y": y, "rel_gap": g, "tables": []}
            out["cases"].append(case)
        print("N=%d: %d candidate pairs below %s, closest %s" % (
            N, len(rows), BAND, ("%.3e" % rows[0][0]) if rows else "none"), flush=True)
    add_tables(out, explicit)
    if overall:
        out["closest_overall"] = dict(zip(("rel_gap", "N", "n1", "n2", "n", "x", "y"), overall))
    with open(os.path.join(HERE, "near_ties.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote near_ties.json;", "closest overall:", out.get("closest_overall"))


if __name__ == "__main__":
    sys.exit(main())
