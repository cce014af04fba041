#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container, where the reference checkout is mounted
read-only at /root/reference (it never travels to the GPU box).  It imports
the reference's own ``scoary.methods`` and records *data only*: inputs the
reference's tests ship (exampledata) and the outputs the reference computes
for them.  No reference source text is stored.

    python tests/golden/make_golden.py            # rewrites every fixture

Fixtures written (all consumed by tests/ and by oracle pinning):

  exampledata/*.gz                 the reference's example inputs (test data)
  setup_results_exampledata.npz    Setup_results() rows for both traits
  setup_results_collapse.npz       same with collapse=True (Tetracycline)
  setup_results_restrict.npz       same under -r Restrict_to.csv
  setup_results_vcf.npz            non-Roary (vcf2scoary output) GPA path
  csv_no_pairwise/*.csv.gz         `--no_pairwise -p 1.0 --no-time` CSVs
  csv_no_pairwise_default/*.csv.gz `--no_pairwise` (default -p 0.05) CSVs
  fisher_grid.npz                  scipy.stats.fisher_exact (SciPy 1.15.3,
                                   the arithmetic behind methods.py:854) on a
                                   table grid incl. ties / zero cells / zero
                                   margins
  permute_tree_seeded.json         seeded reference Permute() (tree statistic;
                                   SURVEY D1 -- pinned for the "next" row)
  synth2/*                         a second, synthetic data set (clade-structured
                                   genes, NA / absent isolates, "0" and "-" cells,
                                   quoted cells) with the reference's CSVs for
                                   --no_pairwise, default mode, --collapse and the
                                   tree it builds
"""
import contextlib
import gzip
import hashlib
import io
import json
import os
import random
import shutil
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
EX = os.path.join(REF, "scoary", "exampledata")

sys.path.insert(0, REF)
import scipy.stats as ss  # noqa: E402

# scipy.stats.binom_test was removed in SciPy >= 1.12; the reference still
# calls it (methods.py:1267).  Patched here, outside the reference tree.
if not hasattr(ss, "binom_test"):
    ss.binom_test = lambda x, n, p: ss.binomtest(int(x), int(n), p).pvalue

import scoary.methods as rm  # noqa: E402


def gz_copy(src, dst):
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(src, "rb") as f, open(dst, "wb") as raw:
        # mtime=0 => byte-stable output
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as g:
            shutil.copyfileobj(f, g)


def gz_write(text, dst):
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    with open(dst, "wb") as raw:
        with gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as g:
            g.write(text.encode())


@contextlib.contextmanager
def quiet():
    old = sys.stdout
    sys.stdout = io.StringIO()
    try:
        yield
    finally:
        sys.stdout = old


def load_inputs(gpa, traits, startcol=14, allowed=None, delimiter=","):
    with open(gpa, "r") as g:
        gd = rm.Csv_to_dic_Roary(g, delimiter, [], startcol=startcol,
                                 allowed_isolates=allowed)
    with open(traits, "r") as t:
        td, prune = rm.Csv_to_dic(t, delimiter, allowed, gd["Strains"])
    return gd, td, prune


def dump_setup_results(path, genedic, traitsdic, collapse):
    with quiet():
        res = rm.Setup_results(genedic, traitsdic, collapse)
    out = {}
    traits = list(res["Results"].keys())
    out["traits"] = np.array(json.dumps(traits))
    out["all_genes"] = np.array(json.dumps(list(genedic.keys())))
    for ti, trait in enumerate(traits):
        rows = res["Results"][trait]
        genes = list(rows.keys())            # insertion order == file order
        out["t%d_genes" % ti] = np.array(json.dumps(genes))
        out["t%d_nugn" % ti] = np.array(json.dumps([rows[g]["NUGN"] for g in genes]))
        out["t%d_annotation" % ti] = np.array(json.dumps([rows[g]["Annotation"] for g in genes]))
        out["t%d_counts" % ti] = np.array(
            [[rows[g]["tpgp"], rows[g]["tpgn"], rows[g]["tngp"], rows[g]["tngn"]]
             for g in genes], dtype=np.int32)
        for k in ("sens", "spes", "OR", "p_v", "B_p", "BH_p"):
            out["t%d_%s" % (ti, k)] = np.array([float(rows[g][k]) for g in genes],
                                               dtype=np.float64)
        # the per-isolate AB/Ab/aB/ab map of a few genes (Gene_trait_combinations)
        gtc = res["Gene_trait_combinations"][trait]
        pick = genes[:3] + genes[-2:]
        out["t%d_gtc" % ti] = np.array(json.dumps({g: gtc[g] for g in pick}))
    np.savez_compressed(path, **out)
    return res


def run_cli(argv, outdir):
    """Run the reference CLI in-process; returns {filename: text}."""
    old_argv = sys.argv
    sys.argv = ["scoary.py"] + argv + ["-o", outdir, "--no-time"]
    try:
        with quiet():
            try:
                rm.main()
            except SystemExit as e:
                if e.code not in (0, None):
                    raise
    finally:
        sys.argv = old_argv
    out = {}
    for fn in sorted(os.listdir(outdir)):
        if fn.endswith(".results.csv"):
            with open(os.path.join(outdir, fn)) as f:
                out[fn] = f.read()
    return out


def fisher_grid(path):
    rng = np.random.default_rng(20260926)
    tabs = []
    # every table with total N in {10, 37}
    for N in (10, 37):
        for a in range(N + 1):
            for b in range(N + 1 - a):
                for c in range(N + 1 - a - b):
                    d = N - a - b - c
                    if N == 37 and (a + b + c) % 3:   # thin N=37 to ~1/3
                        continue
                    tabs.append((a, b, c, d))
    # random margins at the survey's sizes (near-null and associated)
    for N in (100, 500, 2000, 5000, 10000):
        for _ in range(1000):
            n1 = int(rng.integers(1, N))           # trait positives
            n = int(rng.integers(1, N))            # gene margin
            lo, hi = max(0, n - (N - n1)), min(n, n1)
            if rng.random() < 0.6:                 # near the null expectation
                mu = n * n1 / N
                sd = max(1.0, np.sqrt(mu * (1 - n1 / N) * (1 - n / N)))
                a = int(np.clip(round(rng.normal(mu, 2.5 * sd)), lo, hi))
            else:
                a = int(rng.integers(lo, hi + 1))
            tabs.append((a, n1 - a, n - a, N - n1 - n + a))
    # exact ties: symmetric margins (n1 == N/2 makes pmf(x) == pmf(n - x))
    for N in (20, 100, 500, 2000, 5000, 10000):
        for _ in range(300):
            n1 = N // 2
            n = int(rng.integers(2, N - 1))
            lo, hi = max(0, n - (N - n1)), min(n, n1)
            a = int(rng.integers(lo, hi + 1))
            tabs.append((a, n1 - a, n - a, N - n1 - n + a))
    # zero cells (inf / 0 odds ratios) and zero margins (nan, 1.0)
    for a, b, c, d in [(5, 0, 0, 7), (0, 5, 7, 0), (3, 0, 4, 9), (3, 4, 0, 9),
                       (0, 4, 5, 9), (7, 4, 5, 0), (0, 0, 4, 5), (4, 5, 0, 0),
                       (0, 4, 0, 5), (4, 0, 5, 0), (1, 0, 0, 0), (0, 0, 0, 0),
                       (29, 8, 3, 60), (1, 1, 1, 1), (1000, 0, 0, 1000),
                       (5000, 0, 0, 5000), (2500, 2500, 2500, 2500)]:
        tabs.append((a, b, c, d))
    tabs = np.array(tabs, dtype=np.int64)
    p = np.empty(len(tabs))
    orr = np.empty(len(tabs))
    for i, (a, b, c, d) in enumerate(tabs):
        r = ss.fisher_exact([[a, b], [c, d]])
        orr[i], p[i] = r[0], r[1]
    np.savez_compressed(path, tables=tabs, p=p, odds=orr,
                        scipy_version=np.array(__import__("scipy").__version__))
    return len(tabs)


def tree_goldens(gd, td, prune, res, manifest):
    """Population-structure stage: UPGMA trees, PhyloTree
    maxima, the default-mode (pairwise) CSVs, and Permute() driven by OUR
    counter-based label permutations (random.shuffle patched so the reference
    consumes exactly the labels spec S4 generates)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from oracle import oracle as orc
    rng = np.random.default_rng(20260927)
    gpa = os.path.join(EX, "Gene_presence_absence.csv")
    tr = os.path.join(EX, "Tetracycline_resistance.csv")

    # -- UPGMA: exampledata + random matrices with many tied distances --------
    def ref_upgma(zom, names):
        TDM = rm.CreateTriangularDistanceMatrix(zom, names)
        return rm.upgma(rm.PopulateQuadTreeWithDistances(TDM))
    cases = []
    for n, g in [(2, 5), (3, 4), (4, 6), (5, 8), (7, 10), (8, 16), (9, 7), (12, 20), (16, 12),
                 (17, 30), (23, 40), (31, 25), (33, 64), (40, 90)]:
        for rep in range(3):
            X = (rng.random((n, g)) < rng.uniform(0.2, 0.8)).astype(int)
            if rep == 2 and n > 4:               # duplicate strains => zero distances / ties
                X[1] = X[0]
                X[n - 1] = X[n // 2]
            keep = (X.sum(0) > 0) & (X.sum(0) < n)
            if keep.sum() == 0:
                continue
            X = X[:, keep]
            names = ["s%d" % i for i in range(n)]
            cases.append({"matrix": X.tolist(), "names": names,
                          "newick": str(ref_upgma(X.tolist(), names))})
    tree = ref_upgma(gd["Zero_ones_matrix"], gd["Strains"])
    with open(os.path.join(EX, "ExampleTree.nwk")) as f:
        shipped = f.read().strip()
    ours = str(tree).replace("[", "(").replace("]", ")") + ";"
    manifest["upgma_example_equals_shipped_tree"] = (ours == shipped)
    with open(os.path.join(HERE, "upgma_cases.json"), "w") as f:
        json.dump({"cases": cases, "exampledata_tree": str(tree)}, f)

    # -- PhyloTree maxima on random trees / tip states -----------------------
    def rand_tree(tips):
        if len(tips) == 1:
            return tips[0]
        k = int(rng.integers(1, len(tips)))
        if rng.random() < 0.3:
            k = 1                                  # caterpillar-ish
        return [rand_tree(tips[:k]), rand_tree(tips[k:])]
    pcases = []
    for n in [2, 2, 3, 3, 4, 5, 6, 8, 11, 16, 25, 40, 64, 100]:
        for rep in range(6):
            tips = ["t%d" % i for i in range(n)]
            t = rand_tree(tips)
            states = rng.choice(["AB", "Ab", "aB", "ab"], size=n,
                                p=rng.dirichlet([1, 1, 1, 1])).tolist()
            gtc = dict(zip(tips, states))
            pcases.append({"tree": t, "gtc": gtc,
                           "result": rm.ConvertUPGMAtoPhyloTree(t, gtc)})
    with open(os.path.join(HERE, "phylotree_cases.json"), "w") as f:
        json.dump(pcases, f)

    # -- default (pairwise) mode CSVs + tree file -----------------------------
    for sub, extra in (("csv_pairwise_default", ["-u"]),
                       ("csv_pairwise_epw", ["-c", "I", "EPW", "-p", "0.05", "0.05"]),
                       ("csv_pairwise_bh_pw", ["-c", "BH", "PW", "-p", "0.9", "0.05", "-m", "300"]),
                       # the order of equal-p rows depends on --threads (result weave, :1115-1122)
                       ("csv_pairwise_threads3", ["--threads", "3"])):
        od = tempfile.mkdtemp()
        files = run_cli(["-g", gpa, "-t", tr] + extra, od)
        for fn, text in files.items():
            gz_write(text, os.path.join(HERE, sub, fn + ".gz"))
        if "-u" in extra:
            with open(os.path.join(od, "Tree.nwk")) as f:
                manifest["tree_nwk_equals_shipped"] = (f.read().strip() == shipped)

    # -- Permute() fed with spec-S4 permutations -------------------------------
    strains = gd["Strains"]
    col = {s: j for j, s in enumerate(strains)}
    out = {}
    for ti, trait in enumerate(td):
        valid = np.array([1 if s in td[trait] else 0 for s in strains], dtype=np.uint8)
        lab = np.array([1 if td[trait].get(s) == "1" else 0 for s in strains], dtype=np.uint8)
        mb = orc.pack_rows(valid[None])[0]
        npos = int(lab.sum())
        ptree = tree if not [x for x in prune[trait] if x is not None] else \
            rm.PruneForMissing(tree, prune[trait])
        gtcs = res["Gene_trait_combinations"][trait]
        R = res["Results"][trait]
        ranked = sorted(R, key=lambda g: R[g]["p_v"])
        picks = ranked[:6] + ranked[40:43] + ranked[300:302] + ranked[-2:]
        for P, seed in ((100, 7), (600, 12345)):
            for gene in picks:
                state = {"pi": 0}

                def fake_shuffle(lst, _state=state, _gene=gene):
                    bits = orc.perm_labels(seed, ti, _state["pi"], mb, npos, len(strains))
                    b = np.unpackbits(bits.view(np.uint8), bitorder="little")
                    order = list(gtcs[_gene].keys())        # isolates in GTC order
                    want = ["B" if b[col[s]] else "b" for s in order]
                    lst[:] = want[::-1]                     # PermuteGTC pops from the end
                    _state["pi"] += 1
                real = rm.random.shuffle
                rm.random.shuffle = fake_shuffle
                try:
                    with quiet():
                        emp = rm.Permute(tree=ptree, GTC=gtcs[gene], permutations=P,
                                         cutoffs={"I": 0.05})
                except ZeroDivisionError:
                    emp = None
                finally:
                    rm.random.shuffle = real
                obs = rm.ConvertUPGMAtoPhyloTree(ptree, gtcs[gene])
                out.setdefault(trait, []).append(
                    {"gene": gene, "P": P, "seed": seed, "empirical_p": emp,
                     "perms_consumed": state["pi"], "observed": obs})
    with open(os.path.join(HERE, "permute_tree_s4.json"), "w") as f:
        json.dump(out, f, indent=0)


def synth2(manifest):
    """Second data set: 260 genes x 52 isolates in 4 clades, 3 traits with NA /
    missing isolates.  Inputs are generated here (numpy seed below) and stored
    next to the reference's outputs."""
    rng = np.random.default_rng(20260928)
    G, N = 260, 52
    clade = np.repeat(np.arange(4), N // 4)
    base = rng.beta(0.35, 0.35, size=(G, 1))                       # U-shaped frequencies
    shift = rng.normal(0, 0.25, size=(G, 4))[:, clade]             # clade structure
    dense = rng.random((G, N)) < np.clip(base + shift, 0.0, 1.0)
    dense[3] = dense[2]                                            # identical patterns (--collapse)
    dense[11] = dense[2]
    dense[20] = ~dense[19]                                         # complement
    dense[30] = True                                               # core gene   (skip rule)
    dense[31] = False                                              # absent gene (skip rule)
    iso = ["iso_%02d" % i for i in range(N)]
    meta = ["Gene", "Non-unique Gene name", "Annotation", "No. isolates", "No. sequences",
            "Avg sequences per isolate", "Genome Fragment", "Order within Fragment",
            "Accessory Fragment", "Accessory Order with Fragment", "QC", "Min group size nuc",
            "Max group size nuc", "Avg group size nuc"]
    lines = [",".join('"%s"' % h for h in meta + iso)]
    for g in range(G):
        cells = []
        for i in range(N):
            if dense[g, i]:
                cells.append('"%s_%05d"' % (iso[i], g) if (g + i) % 7 else "%s_%05d" % (iso[i], g))
            else:
                cells.append(["", "", '""', "0", "-"][(g * 3 + i) % 5])   # all absent spellings
        lines.append(",".join(['"group_%d"' % g, '"nm%d"' % g if g % 3 else '""',
                               '"hypothetical, protein %d"' % g] + ['"1"'] * 11 + cells))
    gpa_text = "\n".join(lines) + "\n"
    t0 = dense[5] ^ (rng.random(N) < 0.08)                         # gene-driven
    t1 = (clade >= 2) ^ (rng.random(N) < 0.15)                     # clade-driven
    t2 = rng.random(N) < 0.3
    rows = [",driven,lineage,sparse"]
    for i in range(N):
        if i == 17:
            continue                                               # isolate absent from the traits file
        v = [str(int(t0[i])), str(int(t1[i])), str(int(t2[i]))]
        if i in (4, 9, 33, 40):
            v[1] = "NA"
        if i == 50:
            v[2] = "NA"
        rows.append(iso[i] + "," + ",".join(v))
    tr_text = "\n".join(rows) + "\n"
    tmp = tempfile.mkdtemp()
    gpa, tr = os.path.join(tmp, "gpa.csv"), os.path.join(tmp, "traits.csv")
    with open(gpa, "w") as f:
        f.write(gpa_text)
    with open(tr, "w") as f:
        f.write(tr_text)
    gz_write(gpa_text, os.path.join(HERE, "synth2", "gpa.csv.gz"))
    gz_write(tr_text, os.path.join(HERE, "synth2", "traits.csv.gz"))
    for sub, extra in (("no_pairwise", ["--no_pairwise", "-p", "1.0"]),
                       ("collapse", ["--no_pairwise", "--collapse", "-c", "I", "B", "-p", "0.5", "1.0"]),
                       ("pairwise", ["-u", "-c", "I", "EPW", "-p", "0.3", "1.0"])):
        od = tempfile.mkdtemp()
        files = run_cli(["-g", gpa, "-t", tr] + extra, od)
        assert len(files) == 3, files.keys()
        for fn, text in files.items():
            gz_write(text, os.path.join(HERE, "synth2", sub, fn + ".gz"))
            manifest.setdefault("synth2_rows", {})[sub + "/" + fn] = text.count("\n") - 1
        if "-u" in extra:
            with open(os.path.join(od, "Tree.nwk")) as f:
                gz_write(f.read(), os.path.join(HERE, "synth2", "Tree.nwk.gz"))
    # -- less common flags: -r/-w (reduced set written), --include_input_columns,
    #    --delimiter on ';'-separated copies of the same inputs ------------------
    restrict = os.path.join(tmp, "restrict.csv")
    keep = [iso[i] for i in range(N) if i % 3 != 1]
    with open(restrict, "w") as f:
        f.write(",".join(keep) + "\n")
    gz_write(",".join(keep) + "\n", os.path.join(HERE, "synth2", "restrict.csv.gz"))
    od = tempfile.mkdtemp()
    files = run_cli(["-g", gpa, "-t", tr, "--no_pairwise", "-p", "1.0", "-r", restrict, "-w",
                     "--include_input_columns", "4,6-7"], od)
    for fn, text in files.items():
        gz_write(text, os.path.join(HERE, "synth2", "restrict_w", fn + ".gz"))
    with open(os.path.join(od, "gene_presence_absence_reduced.csv")) as f:
        gz_write(f.read(), os.path.join(HERE, "synth2", "restrict_w",
                                        "gene_presence_absence_reduced.csv.gz"))
    import csv as _csv
    semi = {}
    for name, path in (("gpa", gpa), ("traits", tr)):
        with open(path, newline="") as f:
            rows = list(_csv.reader(f))
        buf = io.StringIO()
        _csv.writer(buf, delimiter=";", lineterminator="\n").writerows(rows)
        semi[name] = os.path.join(tmp, name + "_semi.csv")
        with open(semi[name], "w") as f:
            f.write(buf.getvalue())
        gz_write(buf.getvalue(), os.path.join(HERE, "synth2", name + "_semi.csv.gz"))
    od = tempfile.mkdtemp()
    files = run_cli(["-g", semi["gpa"], "-t", semi["traits"], "--no_pairwise", "-p", "0.2",
                     "--delimiter", ";", "-m", "40"], od)
    for fn, text in files.items():
        gz_write(text, os.path.join(HERE, "synth2", "semicolon", fn + ".gz"))


def neartie(manifest):
    """Third data set: the closest near-tie the census found (tests/golden/near_ties.json): 1972
    isolates, trait A with 750 positives, trait B with 756.  Under trait A the genes nt_x / nt_y
    (756 carriers, 282 / 293 of them positive) sit on two support points whose hypergeometric
    weights differ by 1.7e-12 without being equal; under trait B the genes nt_u / nt_v (750
    carriers) are the row/column-swapped pair.  SciPy -- hence the reference -- tells the points
    apart; a tie window wider than 1.7e-12 does not (p 0.6334 instead of 0.6002).  Plus a few
    ordinary genes.  The reference's --no_pairwise CSVs are the golden output."""
    N = 1972
    iso = ["s%04d" % i for i in range(N)]
    tA = np.zeros(N, dtype=int); tA[:750] = 1
    tB = np.zeros(N, dtype=int); tB[:756] = 1

    def gene(pos_in, neg_in, trait):
        g = np.zeros(N, dtype=int)
        pos, neg = np.nonzero(trait == 1)[0], np.nonzero(trait == 0)[0]
        g[pos[:pos_in]] = 1
        g[neg[:neg_in]] = 1
        return g
    genes = {"nt_x": gene(282, 474, tA), "nt_y": gene(293, 463, tA),
             "nt_u": gene(282, 468, tB), "nt_v": gene(293, 457, tB)}
    rng = np.random.default_rng(1972)
    for k in range(6):
        genes["bg_%d" % k] = (rng.random(N) < rng.uniform(0.1, 0.9)).astype(int)
    meta = ["Gene", "Non-unique Gene name", "Annotation", "No. isolates", "No. sequences",
            "Avg sequences per isolate", "Genome Fragment", "Order within Fragment",
            "Accessory Fragment", "Accessory Order with Fragment", "QC", "Min group size nuc",
            "Max group size nuc", "Avg group size nuc"]
    lines = [",".join(meta + iso)]
    for name, g in genes.items():
        lines.append(",".join([name, "", "near-tie census"] + ["1"] * 11 + ["x" if v else "" for v in g]))
    gpa_text = "\n".join(lines) + "\n"
    tr_text = ",A,B\n" + "".join("%s,%d,%d\n" % (iso[i], tA[i], tB[i]) for i in range(N))
    tmp = tempfile.mkdtemp()
    gpa, tr = os.path.join(tmp, "gpa.csv"), os.path.join(tmp, "traits.csv")
    with open(gpa, "w") as f:
        f.write(gpa_text)
    with open(tr, "w") as f:
        f.write(tr_text)
    gz_write(gpa_text, os.path.join(HERE, "neartie", "gpa.csv.gz"))
    gz_write(tr_text, os.path.join(HERE, "neartie", "traits.csv.gz"))
    od = tempfile.mkdtemp()
    files = run_cli(["-g", gpa, "-t", tr, "--no_pairwise", "-p", "1.0"], od)
    for fn, text in files.items():
        gz_write(text, os.path.join(HERE, "neartie", fn + ".gz"))
    manifest["neartie_files"] = sorted(files)


def vcf_cli(manifest):
    """The non-Roary path through the command line: vcf2scoary output + `-s`."""
    tmp = tempfile.mkdtemp()
    mpa = os.path.join(tmp, "mpa.csv")
    with gzip.open(os.path.join(HERE, "exampledata", "mutations_presence_absence.csv.gz"), "rt") as f, \
            open(mpa, "w") as o:
        o.write(f.read())
    od = tempfile.mkdtemp()
    files = run_cli(["-g", mpa, "-t", os.path.join(EX, "ExampleVCFTrait.csv"), "--no_pairwise",
                     "-p", "1.0", "-s", str(manifest["vcf_startcol_1based"])], od)
    for fn, text in files.items():
        gz_write(text, os.path.join(HERE, "csv_vcf_cli", fn + ".gz"))
    manifest["vcf_cli_files"] = sorted(files)


def main():
    os.makedirs(HERE, exist_ok=True)
    manifest = {}
    synth2(manifest)

    # -- inputs the reference's own tests ship --------------------------------
    for fn in ("Gene_presence_absence.csv", "Tetracycline_resistance.csv",
               "Restrict_to.csv", "ExampleVCFTrait.csv", "Example.vcf",
               "ExampleTree.nwk"):
        gz_copy(os.path.join(EX, fn), os.path.join(HERE, "exampledata", fn + ".gz"))

    gpa = os.path.join(EX, "Gene_presence_absence.csv")
    tr = os.path.join(EX, "Tetracycline_resistance.csv")

    # -- Setup_results on exampledata -----------------------------------------
    gd, td, prune = load_inputs(gpa, tr)
    res = dump_setup_results(os.path.join(HERE, "setup_results_exampledata.npz"),
                             gd["Roarydic"], td, False)
    top = res["Results"]["Tetracycline_resistance"]["TetRCG"]
    manifest["TetRCG"] = {k: (float(v) if k not in ("NUGN", "Annotation") else v)
                          for k, v in top.items()}
    manifest["prune"] = {k: [x for x in v if x is not None] for k, v in prune.items()}
    manifest["strains"] = gd["Strains"]

    # -- collapse -------------------------------------------------------------
    gd2, td2, _ = load_inputs(gpa, tr)
    dump_setup_results(os.path.join(HERE, "setup_results_collapse.npz"),
                       gd2["Roarydic"], td2, True)

    # -- restrict_to ----------------------------------------------------------
    with open(os.path.join(EX, "Restrict_to.csv")) as f:
        allowed = {iso: "all" for line in f for iso in line.rstrip().split(",")}
    gd3, td3, _ = load_inputs(gpa, tr, allowed=allowed)
    dump_setup_results(os.path.join(HERE, "setup_results_restrict.npz"),
                       gd3["Roarydic"], td3, False)
    manifest["restrict_strains"] = gd3["Strains"]

    # -- vcf2scoary output -> non-Roary GPA path -----------------------------
    tmp = tempfile.mkdtemp()
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        import scoary.vcf2scoary as v2s
        old = sys.argv
        sys.argv = ["vcf2scoary", "--force", os.path.join(EX, "Example.vcf")]
        with quiet():
            try:
                v2s.main()
            except SystemExit:
                pass
        sys.argv = old
        mpa = os.path.join(tmp, "mutations_presence_absence.csv")
        gz_copy(mpa, os.path.join(HERE, "exampledata", "mutations_presence_absence.csv.gz"))
        with open(mpa) as f:
            header = f.readline().rstrip("\n").split(",")
        startcol = header.index("DUMMY") + 1 if "DUMMY" in header else 9
        manifest["vcf_startcol_1based"] = startcol + 1
        gd4, td4, _ = load_inputs(mpa, os.path.join(EX, "ExampleVCFTrait.csv"),
                                  startcol=startcol)
        dump_setup_results(os.path.join(HERE, "setup_results_vcf.npz"),
                           gd4["Roarydic"], td4, False)
        manifest["vcf_firstcolnames"] = gd4["Firstcolnames"]
        manifest["vcf_strains"] = gd4["Strains"]
        # every FILE row enters the tree-building matrix, also rows whose identifier repeats
        # (8 rows -> 5 identifiers here): scoary/methods.py:445-497
        manifest["vcf_zero_ones_matrix"] = gd4["Zero_ones_matrix"]
        manifest["vcf_file_rows"] = sum(1 for _ in open(mpa)) - 1
    finally:
        os.chdir(cwd)

    # -- CLI CSVs (the byte-level output contract, StoreTraitResult) ----------
    for sub, extra in (("csv_no_pairwise", ["-p", "1.0"]),
                       ("csv_no_pairwise_default", []),
                       ("csv_no_pairwise_collapse_bh",
                        ["--collapse", "-c", "I", "BH", "-p", "0.05", "0.01", "-m", "50"])):
        od = tempfile.mkdtemp()
        files = run_cli(["-g", gpa, "-t", tr, "--no_pairwise"] + extra, od)
        for fn, text in files.items():
            gz_write(text, os.path.join(HERE, sub, fn + ".gz"))
            manifest.setdefault("csv_sha256", {})[sub + "/" + fn] = \
                hashlib.sha256(text.encode()).hexdigest()
    od = tempfile.mkdtemp()
    files = run_cli(["-g", gpa, "-t", tr, "--no_pairwise", "-r",
                     os.path.join(EX, "Restrict_to.csv")], od)
    for fn, text in files.items():
        gz_write(text, os.path.join(HERE, "csv_no_pairwise_restrict", fn + ".gz"))

    # -- third-party arithmetic: scipy.stats.fisher_exact grid ----------------
    manifest["fisher_grid_n"] = fisher_grid(os.path.join(HERE, "fisher_grid.npz"))

    # -- seeded tree-statistic Permute (pinned for the "next" row, D1) --------
    gtc = res["Gene_trait_combinations"]["Tetracycline_resistance"]
    TDM = rm.CreateTriangularDistanceMatrix(gd["Zero_ones_matrix"], gd["Strains"])
    QT = rm.PopulateQuadTreeWithDistances(TDM)
    tree = rm.upgma(QT)
    obs = rm.ConvertUPGMAtoPhyloTree(tree, gtc["TetRCG"])
    random.seed(0)
    with quiet():
        emp = rm.Permute(tree=tree, GTC=gtc["TetRCG"], permutations=100,
                         cutoffs={"I": 0.05})
    with open(os.path.join(HERE, "permute_tree_seeded.json"), "w") as f:
        json.dump({"gene": "TetRCG", "observed": obs, "seed": 0,
                   "permutations": 100, "empirical_p": emp}, f, indent=1)

    tree_goldens(gd, td, prune, res, manifest)
    vcf_cli(manifest)
    neartie(manifest)

    with open(os.path.join(HERE, "manifest.json"), "w") as f:
        json.dump(manifest, f, indent=1, sort_keys=True)
    print("golden vectors written to", HERE)


if __name__ == "__main__":
    main()
