#!/usr/bin/env python3
"""Randomised differential corpus: small inputs with the awkward shapes a drop-in has to survive,
run through the REAL reference command line (imported from /root/reference, as make_golden.py does).

    python tests/golden/make_fuzz.py            # rewrites tests/golden/fuzz_corpus.json.gz

Every case = the text of a gene presence/absence table, a traits table (and maybe a restrict-to
list), the flags, and what the reference did with them: the result CSVs (file name -> text), the
UPGMA tree if one was asked for, or the message it exited with.  Data only; the GPU test
(tests/test_gpu_fuzz.py) feeds the same files and flags to ``python -m scoary_amd``'s main().

What the generator varies (seeded; the case list is a pure function of SEED):
  * Roary-format tables (14 metadata columns) and plain tables (-s 4 ... ), ',' and ';' delimiters;
  * 4 .. 400 isolates, 1 .. 150 genes; uniform, rare, U-shaped and clade-structured genes; core and
    absent genes (the skip rule), duplicated patterns and complements (--collapse), repeated gene
    identifiers (the later row wins, scoary/methods.py:452);
  * every spelling of an absent cell ("", "0", "-"), quoted cells, cells with a leading blank
    (the reader's skipinitialspace);
  * 1 .. 4 traits; the top-left cell empty or "Name"; every spelling of a missing value; isolates
    in another order, isolates missing from the traits file, isolates the gene table does not have;
    traits with all values equal, with one positive, with everything missing but a few;
  * flags: --no_pairwise or the pairwise stage (-u), -c subsets with one or matching -p lists,
    -m, --collapse, -r (+ -w), --include_input_columns, --threads;
  * inputs the reference refuses (sys.exit with a message): kept with the message.
Every case also records what the reference's two READERS (Csv_to_dic_Roary, Csv_to_dic) return for its tables
written in five ways -- as they are, with Windows line ends, behind a byte-order mark, with a blank last line,
without a final newline -- as digests (tests/test_readers_fuzz.py, no GPU needed: the readers are host code).
Cases on which the reference itself raises anything but SystemExit are dropped (counted in the
manifest entry): there is no behaviour to be compatible with.
"""
import contextlib
import csv
import gzip
import io
import json
import os
import shutil
import sys
import tempfile
import warnings

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
# The committed corpora: a plain run writes fuzz_corpus.json.gz (SEED, 200 cases, NS) and fuzz_args_corpus.json.gz (the
# flag surface: ARGS_SEED, 200 odd command lines on small tables).  FUZZ_SEED / FUZZ_NCASES / FUZZ_NS / FUZZ_ARGMUT=1 /
# FUZZ_KEEP_CRASHES=1 / FUZZ_OUT: ONE one-off extra corpus instead (tools/round6_run.sh fuzzextra).
SEED = int(os.environ.get("FUZZ_SEED", "20261001"))
NCASES = int(os.environ.get("FUZZ_NCASES", "200"))
NS = [int(x) for x in os.environ.get("FUZZ_NS", "4,5,7,12,20,33,48,64,65,70,180,256,400").split(",")]   # isolates
DEFAULT_GS = [1, 2, 5, 17, 40, 40, 90, 90, 150]                                                           # genes
GS = [int(x) for x in os.environ["FUZZ_GS"].split(",")] if os.environ.get("FUZZ_GS") else DEFAULT_GS     # one-off: big tables
ARGS_SEED, ARGS_NS = 20261002, [4, 7, 12, 20, 33, 48, 70]
ARGMUT = os.environ.get("FUZZ_ARGMUT") == "1"          # set per corpus by build()
ODD_NAMES = os.environ.get("FUZZ_ODD_NAMES") == "1"    # one-off: non-ASCII / quoted / delimiter-holding isolate and gene names

sys.path.insert(0, REF)
import scipy.stats as ss  # noqa: E402

if not hasattr(ss, "binom_test"):       # removed in SciPy >= 1.12, still called at methods.py:1267
    ss.binom_test = lambda x, n, p: ss.binomtest(int(x), int(n), p).pvalue

warnings.filterwarnings("ignore")
import scoary.methods as rm  # noqa: E402

META = ["Gene", "Non-unique Gene name", "Annotation", "No. isolates", "No. sequences",
        "Avg sequences per isolate", "Genome Fragment", "Order within Fragment",
        "Accessory Fragment", "Accessory Order with Fragment", "QC", "Min group size nuc",
        "Max group size nuc", "Avg group size nuc"]


def gene_matrix(rng, G, N):
    kind = rng.choice(["uniform", "rare", "ushaped", "clade"])
    if kind == "uniform":
        f = rng.uniform(0.05, 0.95, size=(G, 1))
    elif kind == "rare":
        f = rng.beta(0.3, 3.0, size=(G, 1))
    elif kind == "ushaped":
        f = rng.beta(0.2, 0.2, size=(G, 1))
    else:
        k = int(rng.integers(2, 5))
        clade = np.sort(rng.integers(0, k, size=N))
        f = np.clip(rng.beta(0.4, 0.4, size=(G, 1)) + rng.normal(0, 0.3, size=(G, k))[:, clade], 0, 1)
    m = rng.random((G, N)) < f
    # the rows the skip rule, --collapse and the tie handling look at
    for _ in range(int(rng.integers(0, 4))):
        a, b = rng.integers(0, G, size=2)
        m[a] = m[b]
    for _ in range(3 if N > 170 else 1):
        if G > 3 and rng.random() < 0.5:
            a, b = rng.integers(0, G, size=2)
            m[a] = ~m[b]
    if rng.random() < 0.5:
        m[rng.integers(0, G)] = True
    if rng.random() < 0.5:
        m[rng.integers(0, G)] = False
    return m, str(kind)


def write_table(rows, delimiter, quote_all=False):
    buf = io.StringIO()
    w = csv.writer(buf, delimiter=delimiter, lineterminator="\n",
                   quoting=csv.QUOTE_ALL if quote_all else csv.QUOTE_MINIMAL)
    w.writerows(rows)
    return buf.getvalue()


def make_case(rng, k):
    N = int(rng.choice(NS))
    G = int(rng.choice(GS))
    if N > 170 and GS is DEFAULT_GS:
        # beyond SciPy's factorial table a gene and its complement get the SAME double (below: a coin flip),
        # so here the order of such rows in the reference's CSV is the dictionary's and can be held to it
        G = min(G, 40)
    roary = rng.random() < 0.7
    delimiter = ";" if rng.random() < 0.2 else ","
    m, kind = gene_matrix(rng, G, N)
    iso = ["iso%02d" % i if rng.random() < 0.8 else "s_%d.x" % i for i in range(N)]
    absent = ["", "0", "-"]
    nmeta = 14 if roary else int(rng.integers(3, 7))
    header = (META if roary else ["locus", "alias", "note", "x1", "x2", "x3"][:nmeta]) + iso
    names = ["g%04d" % g for g in range(G)]
    for _ in range(int(rng.integers(0, 3))):             # repeated identifiers: the later row wins
        if G > 2:
            a, b = rng.integers(0, G, size=2)
            names[a] = names[b]
    lead_blank = rng.random() < 0.3
    if ODD_NAMES:        # one-off corpora (no random draws: the committed corpora stay what they are)
        iso = ["isol\u00e9 %d" % i if i % 5 == 0 else "\u540d%d" % i if i % 7 == 0 else "st,%d" % i if i % 11 == 0
               else 'q"%d' % i if i % 13 == 0 else x for i, x in enumerate(iso)]
        header = header[:nmeta] + iso
        names = ["g\u00e9ne_%d" % g if g % 6 == 0 else x for g, x in enumerate(names)]
    rows = [header]
    for g in range(G):
        meta = [names[g], ("nm%d" % g) if g % 3 else "", "protein, putative %d" % g if g % 4 else "hyp %d" % g]
        if ODD_NAMES and g % 5 == 1:
            meta[2] = '\u00b5-protein "%d", 5\' end; x' % g
        meta += [str(int(m[g].sum()))] * (nmeta - 3)
        cells = []
        for i in range(N):
            if m[g, i]:
                c = "p%d" % g if rng.random() < 0.9 else "1"
            else:
                c = absent[int(rng.integers(0, 3))]
            if lead_blank and c and rng.random() < 0.1:
                c = " " + c                               # skipinitialspace strips it: still present
            cells.append(c)
        rows.append(meta + cells)
    gpa = write_table(rows, delimiter, quote_all=rng.random() < 0.25)

    T = int(rng.choice([1, 1, 2, 3, 4]))
    topleft = "" if rng.random() < 0.7 else "Name"
    tnames = ["trait%d" % t if rng.random() < 0.8 else "res (%d)" % t for t in range(T)]
    if ODD_NAMES:
        tnames = ["r\u00e9sistance, (%d)" % t if t % 2 == 0 else x for t, x in enumerate(tnames)]
    missing = ["NA", "-", ".", " ", ""]
    tv = np.empty((T, N), dtype=object)
    for t in range(T):
        style = rng.choice(["random", "random", "gene", "gene", "rare", "constant", "one", "mostly_missing"])
        if style == "gene" and G > 0:
            v = m[rng.integers(0, G)] ^ (rng.random(N) < 0.1)
        elif style == "rare":
            v = rng.random(N) < 0.1
        elif style == "constant":
            v = np.full(N, rng.random() < 0.5)
        elif style == "one":
            v = np.zeros(N, dtype=bool)
            v[rng.integers(0, N)] = True
        else:
            v = rng.random(N) < rng.uniform(0.2, 0.8)
        tv[t] = [str(int(x)) for x in v]
        pm = 0.6 if style == "mostly_missing" else (0.1 if rng.random() < 0.4 else 0.0)
        for i in range(N):
            if rng.random() < pm:
                tv[t, i] = missing[int(rng.integers(0, 5))]
    order = rng.permutation(N) if rng.random() < 0.5 else np.arange(N)
    drop = set(rng.choice(N, size=int(rng.integers(1, max(2, N // 6))), replace=False).tolist()) \
        if rng.random() < 0.3 else set()
    trows = [[topleft] + tnames]
    for i in order:
        if int(i) in drop:
            continue
        trows.append([iso[i]] + [tv[t, i] for t in range(T)])
    if rng.random() < 0.04:                               # isolates the gene table does not have: refused
        for j in range(int(rng.integers(1, 4))):
            trows.insert(int(rng.integers(1, len(trows) + 1)),
                         ["extra_%d" % j] + [str(int(rng.random() < 0.5)) for _ in range(T)])
    bad = None
    r = rng.random()
    if r < 0.04:                                          # inputs the reference refuses
        trows[min(2, len(trows) - 1)][1] = "2"
        bad = "value"
    elif r < 0.07:
        trows[0][0] = "isolate"
        bad = "topleft"
    traits = write_table(trows, delimiter)

    argv = []
    if not roary:
        argv += ["-s", str(nmeta + 1)]
    elif rng.random() < 0.1:
        argv += ["-s", "15"]
    if delimiter != ",":
        argv += ["--delimiter", delimiter]
    pairwise = rng.random() < 0.3 and N >= 5
    corr_pool = ["I", "B", "BH"] + (["PW", "EPW"] if pairwise else [])
    if rng.random() < 0.6:
        nc = int(rng.integers(1, len(corr_pool) + 1))
        corr = [str(c) for c in rng.choice(corr_pool, size=nc, replace=False)]
        argv += ["-c"] + corr
    else:
        corr = ["I"]
    pvals = [1.0, 0.9, 0.5, 0.2, 0.05]
    if rng.random() < 0.5:
        argv += ["-p", str(pvals[int(rng.integers(0, 5))])]
    elif rng.random() < 0.6:
        argv += ["-p"] + [str(pvals[int(rng.integers(0, 4))]) for _ in corr]
    elif rng.random() < 0.1:
        argv += ["-p", "0.5", "0.5", "0.5", "0.5", "0.5", "0.5", "0.5"]   # more cut-offs than methods: refused
    if not pairwise:
        argv += ["--no_pairwise"]
    elif rng.random() < 0.7:
        argv += ["-u"]
    if rng.random() < 0.2:
        argv += ["-m", str(int(rng.integers(1, 12)))]
    if rng.random() < 0.25:
        argv += ["--collapse"]
    if rng.random() < 0.15:
        argv += ["--threads", str(int(rng.integers(2, 4)))]
    restrict = None
    if rng.random() < 0.2 and N >= 7:
        keep = [iso[i] for i in range(N) if rng.random() < 0.7]
        if rng.random() < 0.3:
            keep.append("not_an_isolate")
        half = len(keep) // 2
        restrict = ",".join(keep[:half]) + "\n" + ",".join(keep[half:]) + "\n" if rng.random() < 0.3 \
            else ",".join(keep) + "\n"
        argv += ["-r", "RESTRICT"]
        if rng.random() < 0.4:
            argv += ["-w"]
    elif rng.random() < 0.03:
        argv += ["-w"]                                    # -w without -r: refused
    if roary and rng.random() < 0.15:
        argv += ["--include_input_columns", str(rng.choice(["4", "4,6-7", "ALL", "5-6"]))]
    if ARGMUT:                                             # the flag surface (validation, exits)
        argv = mutate_argv(rng, argv)
    return {"id": k, "kind": kind, "roary": bool(roary), "N": N, "G": G, "T": T, "bad": bad,
            "gpa": gpa, "traits": traits, "restrict": restrict, "argv": argv}


def mutate_argv(rng, argv):
    """Odd but plausible command lines: values out of range, words where numbers belong, repeated and contradictory
    flags.  (No --permute >= 10: the reference's permutations are unseeded.)"""
    pool = [["-m", "0"], ["-m", "-1"], ["-m", "abc"], ["-m", "1"], ["-p", "1.5"], ["-p", "-0.1"], ["-p", "abc"],
            ["-p", "1e-400"], ["-p", "0"], ["-p", "1"], ["-p", "1.0", "0.05"], ["-c", "XYZ"], ["-c", "I", "I"],
            ["-c", "bh"], ["-c", "P"], ["-c", "EPW"], ["-c", "PW", "EPW", "BH", "B", "I"], ["-c"],
            ["--threads", "0"], ["--threads", "-2"], ["--threads", "40"], ["--threads", "x"],
            ["-s", "1"], ["-s", "2"], ["-s", "0"], ["-s", "999"], ["-s", "-3"], ["-s", "4"], ["-s", "16"],
            ["-e", "5"], ["-e", "0"], ["-e", "-1"], ["-e", "1.5"], ["--delimiter", "\t"], ["--delimiter", ";;"],
            ["--delimiter", ""], ["--delimiter", ","], ["--include_input_columns", "0"],
            ["--include_input_columns", "99"], ["--include_input_columns", "3-1"], ["--include_input_columns", "abc"],
            ["--include_input_columns", "1,2,3"], ["--include_input_columns", "2-"], ["-u"], ["--no_pairwise"], ["-w"],
            ["--collapse"], ["--collapse", "--collapse"], ["-r", "/nonexistent/file.csv"], ["--version"], ["--citation"],
            ["--bogus"], ["-n", "/nonexistent/tree.nwk"], ["--no-time"], ["-o", "/nonexistent_dir/x"], ["-t"], ["-g"]]
    out = list(argv)
    for _ in range(int(rng.integers(1, 3))):
        extra = pool[int(rng.integers(0, len(pool)))]
        if rng.random() < 0.5 and extra[0] in out and extra[0] not in ("-r",):
            i = out.index(extra[0])                           # replace the flag's values instead of repeating the flag
            j = i + 1
            while j < len(out) and not (out[j].startswith("-") and not _is_number(out[j])):
                j += 1
            out[i:j] = extra
        else:
            at = int(rng.integers(0, len(out) + 1))
            while 0 < at < len(out) and not out[at].startswith("-"):
                at += 1                                       # never between a flag and its values
            out[at:at] = extra
    return out


def _is_number(x):
    try:
        float(x)
        return True
    except ValueError:
        return False


@contextlib.contextmanager
def quiet():
    old_out, old_err = sys.stdout, sys.stderr
    sys.stdout = sys.stderr = io.StringIO()
    try:
        yield
    finally:
        sys.stdout, sys.stderr = old_out, old_err


def run_reference(case):
    tmp = tempfile.mkdtemp(prefix="fuzz_")
    try:
        paths = {}
        for key in ("gpa", "traits", "restrict"):
            if case[key] is not None:
                paths[key] = os.path.join(tmp, key + ".csv")
                with open(paths[key], "w", newline="") as f:
                    f.write(case[key])
        od = os.path.join(tmp, "out")
        os.mkdir(od)
        argv = ["-g", paths["gpa"], "-t", paths["traits"]] + \
               [paths["restrict"] if a == "RESTRICT" else a for a in case["argv"]] + ["-o", od, "--no-time"]
        old = sys.argv
        sys.argv = ["scoary.py"] + argv
        # the reference keeps its log handlers and call counters in module state
        for hdl in list(rm.log.handlers):
            rm.log.removeHandler(hdl)
        status, message = "ok", None
        try:
            with quiet():
                try:
                    rm.main()
                except SystemExit as e:
                    if e.code not in (0, None):
                        status, message = "exit", str(e.code)
        except BaseException as e:      # the reference's own crash: nothing to be compatible with
            return {"status": "crash", "message": "%s: %s" % (type(e).__name__, e)}
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
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


READER_VARIANTS = ("plain", "crlf", "bom", "trailing_blank", "no_final_newline")


def reader_variant(text, variant):
    """The same table as another file: Windows line ends, a byte-order mark, a blank last line, no final
    newline.  (The command line opens its inputs in text mode with universal newlines, methods.py:192-193.)"""
    if variant == "crlf":
        return text.replace("\n", "\r\n")
    if variant == "bom":
        return "\ufeff" + text
    if variant == "trailing_blank":
        return text + "\n"
    if variant == "no_final_newline":
        return text.rstrip("\n")
    return text


def reader_args(case):
    """(delimiter, startcol, grabcols, allowed isolates) as main() derives them from the flags."""
    argv = case["argv"]
    delimiter = ";" if "--delimiter" in argv else ","
    startcol = (int(argv[argv.index("-s") + 1]) if "-s" in argv else 15) - 1
    grab = []
    if "--include_input_columns" in argv:
        grab = rm.grabcoltype(argv[argv.index("--include_input_columns") + 1])
        if grab == "ALL":
            grab = [-999]
    allowed = None
    if case["restrict"]:
        allowed = {iso: "all" for line in case["restrict"].splitlines(True) for iso in line.rstrip().split(",")}
    return delimiter, startcol, grab, allowed


def normalise_gene_table(res):
    """What Csv_to_dic_Roary returns, as plain JSON-able data."""
    g = res["Roarydic"]
    return {"ids": list(g.keys()), "strains": list(res["Strains"]), "extracols": list(res["Extracols"]),
            "first": list(res["Firstcolnames"]), "zom": [[int(x) for x in r] for r in res["Zero_ones_matrix"]],
            "rows": {k: {a: (int(b) if not isinstance(b, str) else b) for a, b in dict(g[k]).items()}
                     for k in g.keys()}}


def digest(obj):
    import hashlib
    return hashlib.sha256(json.dumps(obj, sort_keys=True).encode()).hexdigest()[:24]


def call_reader(fn, path, *a, **k):
    """-> ("ok", value) | ("exit", message) | ("crash", exception name)"""
    try:
        with open(path, "r") as f, quiet():
            return "ok", fn(f, *a, **k)
    except SystemExit as e:
        return "exit", str(e.code)
    except BaseException as e:
        return "crash", type(e).__name__


def reader_records(case, methods_module):
    """{variant: {"genes": digest | ["exit", message] | None, "traits": ...}} -- what the two readers of
    `methods_module` (the reference's here; scoary_amd's in tests/test_readers_fuzz.py) make of the case's
    tables written in each variant.  None = the reference itself crashed (nothing to compare)."""
    delimiter, startcol, grab, allowed = reader_args(case)
    out = {}
    tmp = tempfile.mkdtemp(prefix="fuzzrd_")
    try:
        for v in READER_VARIANTS:
            pg, pt = os.path.join(tmp, v + "_g.csv"), os.path.join(tmp, v + "_t.csv")
            with open(pg, "w", newline="", encoding="utf-8") as f:
                f.write(reader_variant(case["gpa"], v))
            with open(pt, "w", newline="", encoding="utf-8") as f:
                f.write(reader_variant(case["traits"], v))
            st, val = call_reader(methods_module.Csv_to_dic_Roary, pg, delimiter, list(grab), startcol=startcol,
                                  allowed_isolates=allowed)
            rec = {"genes": None, "traits": None}
            if st == "ok":
                table = normalise_gene_table(val)
                rec["genes"] = digest(table)
                st2, val2 = call_reader(methods_module.Csv_to_dic, pt, delimiter, allowed, table["strains"])
                if st2 == "ok":
                    rec["traits"] = digest([{k: dict(x) for k, x in dict(val2[0]).items()},
                                            {k: list(x) for k, x in dict(val2[1]).items()}])
                elif st2 == "exit":
                    rec["traits"] = ["exit", val2]
            elif st == "exit":
                rec["genes"] = ["exit", val]
            out[v] = rec
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def build(seed, ncases, ns, argmut, path, keep_crashes=False):
    global NS, ARGMUT
    NS, ARGMUT = list(ns), bool(argmut)
    rng = np.random.default_rng(seed)
    cases, crashed, crash_cases = [], [], []
    k = 0
    while len(cases) < ncases and k < 4 * ncases:
        case = make_case(rng, k)
        k += 1
        ref = run_reference(case)
        if ref["status"] == "crash":
            crashed.append({"id": case["id"], "message": ref["message"][:200]})
            if keep_crashes:                              # one-off corpora: what does OUR command line do there?
                case["ref"] = ref
                crash_cases.append(case)
            continue
        case["ref"] = ref
        # (odd command lines: reader_args would trip over the very flags under test; the readers have their corpus)
        case["readers"] = reader_records(case, rm) if not argmut else {}
        cases.append(case)
    doc = {"seed": seed, "generated": k, "kept": len(cases), "reference_crashes": crashed, "cases": cases}
    if crash_cases:
        doc["crash_cases"] = crash_cases
    with open(path, "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as g:
            g.write(json.dumps(doc, sort_keys=True).encode())
    n_ok = sum(c["ref"]["status"] == "ok" for c in cases)
    print("%s: %d cases kept of %d generated (%d ok, %d refused by the reference, %d reference crashes dropped); %d bytes"
          % (os.path.basename(path), len(cases), k, n_ok, len(cases) - n_ok, len(crashed), os.path.getsize(path)))
    for c in crashed[:20]:
        print("  dropped", c)


def main():
    one_off = any(os.environ.get(k) for k in ("FUZZ_SEED", "FUZZ_NCASES", "FUZZ_NS", "FUZZ_ARGMUT", "FUZZ_OUT",
                                              "FUZZ_KEEP_CRASHES", "FUZZ_ODD_NAMES", "FUZZ_GS"))
    if one_off:
        build(SEED, NCASES, NS, ARGMUT, os.environ.get("FUZZ_OUT") or os.path.join(HERE, "fuzz_corpus.json.gz"),
              keep_crashes=os.environ.get("FUZZ_KEEP_CRASHES") == "1")
        return
    build(SEED, NCASES, NS, False, os.path.join(HERE, "fuzz_corpus.json.gz"))
    build(ARGS_SEED, 200, ARGS_NS, True, os.path.join(HERE, "fuzz_args_corpus.json.gz"))


if __name__ == "__main__":
    main()
