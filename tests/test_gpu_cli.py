"""Drop-in boundary on the GPU: the scoary command line (-g/-t/-p/-c/--permute/
--no_pairwise/--collapse/-r/-m/--no-time) and the Setup_results / Permute call
sites, against files and dictionaries captured from the real reference."""
import csv
import gzip
import io
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, golden_text, read_dense

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")


def run_cli(argv, outdir):
    from scoary_amd import methods as m
    old = sys.argv
    sys.argv = ["scoary"] + argv + ["-o", str(outdir), "--no-time"]
    try:
        with pytest.raises(SystemExit) as e:
            m.main()
        assert e.value.code in (0, None), e.value.code
    finally:
        sys.argv = old
    out = {}
    for fn in sorted(os.listdir(outdir)):
        if fn.endswith(".results.csv"):
            with open(os.path.join(outdir, fn), newline="") as f:
                out[fn] = f.read()
    return out


def _inputs(exampledir):
    return ["-g", os.path.join(exampledir, "Gene_presence_absence.csv"),
            "-t", os.path.join(exampledir, "Tetracycline_resistance.csv")]


def _assert_csv_equal(got, want, p_tol=1e-12):
    """Byte-identical where possible; otherwise identical structure, integer and
    text cells exact, and float cells within the north_star tolerance (the GPU
    Fisher p may differ from SciPy's in the last bits)."""
    if got == want:
        return "bytes"
    g = list(csv.reader(io.StringIO(got)))
    w = list(csv.reader(io.StringIO(want)))
    assert g[0] == w[0], "header differs"
    assert len(g) == len(w), "row count differs: %d vs %d" % (len(g), len(w))
    gi = {r[0]: r for r in g[1:]}
    for wr in w[1:]:
        gr = gi[wr[0]]
        for k, (a, b) in enumerate(zip(gr, wr)):
            if a == b:
                continue
            fa, fb = float(a), float(b)
            tol = p_tol * (6000 if g[0][k] in ("Bonferroni_p", "Benjamini_H_p") else 1)
            assert abs(fa - fb) <= tol + 1e-11 * abs(fb), (wr[0], g[0][k], a, b)
    # row ORDER must agree wherever the sort keys are distinguishable
    gp = [float(r[g[0].index("Naive_p")]) for r in g[1:]]
    assert all(gp[i] <= gp[i + 1] * (1 + 1e-9) for i in range(len(gp) - 1))
    return "tolerance"


@pytest.mark.parametrize("sub,extra", [
    ("csv_no_pairwise", ["-p", "1.0"]),
    ("csv_no_pairwise_default", []),
    ("csv_no_pairwise_collapse_bh", ["--collapse", "-c", "I", "BH", "-p", "0.05", "0.01", "-m", "50"]),
    ("csv_no_pairwise_restrict", ["-r", "RESTRICT"]),
])
def test_cli_results_csv_vs_reference(exampledir, tmp_path, sub, extra):
    extra = [os.path.join(exampledir, "Restrict_to.csv") if x == "RESTRICT" else x for x in extra]
    files = run_cli(_inputs(exampledir) + ["--no_pairwise"] + extra, tmp_path)
    assert sorted(files) == ["Bogus_trait.results.csv", "Tetracycline_resistance.results.csv"]
    modes = []
    for fn, text in files.items():
        want = golden_text(os.path.join(sub, fn + ".gz"))
        modes.append(_assert_csv_equal(text, want))
    print(sub, modes)
    # the log file exists and is named as in the reference
    assert os.path.exists(os.path.join(tmp_path, "scoary.log"))


def test_cli_top_hit_is_the_reference_travis_row(exampledir, tmp_path):
    """tests/test_scoary_output.py:12-14 of the reference, applied to our CSV."""
    files = run_cli(_inputs(exampledir) + ["--no_pairwise"], tmp_path)
    rows = list(csv.reader(io.StringIO(files["Tetracycline_resistance.results.csv"])))
    d = rows[1]
    assert d[0] == "TetRCG" and d[1] == "" and d[2].startswith("A fictitious gene")
    assert [int(x) for x in d[3:7]] == [29, 8, 3, 60]
    assert abs(float(d[7]) - 90.625) < 0.01 and abs(float(d[8]) - 88.2352941176) < 0.01
    assert abs(float(d[9]) - 72.5) < 0.1
    assert abs(float(d[10]) - 1.08621066108E-014) < 1e-15
    assert abs(float(d[11]) - 6.45209132679E-011) < 1e-12
    assert abs(float(d[12]) - 6.45209132679E-011) < 1e-12


def _load(exampledir, allowed=None):
    from scoary_amd import methods as m
    with open(os.path.join(exampledir, "Gene_presence_absence.csv")) as f:
        gd = m.Csv_to_dic_Roary(f, ",", [], startcol=14, allowed_isolates=allowed)
    with open(os.path.join(exampledir, "Tetracycline_resistance.csv")) as f:
        td, prune = m.Csv_to_dic(f, ",", allowed, gd["Strains"])
    return gd, td


@pytest.mark.parametrize("fixture,collapse", [("setup_results_exampledata.npz", False),
                                              ("setup_results_collapse.npz", True)])
def test_setup_results_vs_reference_dump(exampledir, fixture, collapse):
    from scoary_amd import methods as m
    gd, td = _load(exampledir)
    res = m.Setup_results(gd["Roarydic"], td, collapse)
    z = np.load(os.path.join(GOLDEN, fixture))
    assert list(res["Results"]) == json.loads(str(z["traits"]))
    for t, trait in enumerate(res["Results"]):
        R = res["Results"][trait]
        assert list(R) == json.loads(str(z["t%d_genes" % t]))      # names AND order
        assert R.nugn == json.loads(str(z["t%d_nugn" % t]))
        assert R.annotation == json.loads(str(z["t%d_annotation" % t]))
        want = z["t%d_counts" % t]                                  # tpgp,tpgn,tngp,tngn
        got = np.stack([R.column(k) for k in ("tpgp", "tpgn", "tngp", "tngn")], 1)
        assert np.array_equal(got, want)
        assert np.array_equal(R.column("sens"), z["t%d_sens" % t])
        assert np.array_equal(R.column("spes"), z["t%d_spes" % t])
        go = z["t%d_OR" % t]
        fin = np.isfinite(go)
        assert np.array_equal(R.column("OR")[fin], go[fin])
        assert np.array_equal(np.isinf(R.column("OR")), np.isinf(go))
        gp = z["t%d_p_v" % t]
        assert np.max(np.abs(R.column("p_v") - gp)) < 1e-12
        assert np.max(np.abs(R.column("B_p") - z["t%d_B_p" % t])) < 1e-12 * R.number_of_tests
        assert np.max(np.abs(R.column("BH_p") - z["t%d_BH_p" % t])) < 1e-12 * R.number_of_tests
        # row dict has the reference's keys
        row = R[list(R)[0]]
        assert set(row) == {"NUGN", "Annotation", "tpgp", "tngp", "tpgn", "tngn", "sens",
                            "spes", "OR", "p_v", "B_p", "BH_p"}
        # Gene_trait_combinations for the genes the fixture sampled
        gtc = json.loads(str(z["t%d_gtc" % t]))
        for g, want_map in gtc.items():
            assert res["Gene_trait_combinations"][trait][g] == want_map
    if collapse:
        assert res["Results"]["Tetracycline_resistance"].number_of_tests == 5925


def test_setup_results_accepts_reference_style_dicts(exampledir):
    """The internal call site: plain dict-of-dicts genedic as the reference's
    reader builds it (methods.py:445-491)."""
    from scoary_amd import methods as m
    ids, strains, genes, names, traits = read_dense(
        golden_text("exampledata/Gene_presence_absence.csv.gz"),
        golden_text("exampledata/Tetracycline_resistance.csv.gz"))
    pick = [ids.index("TetRCG")] + list(range(0, len(ids), 90))
    genedic = {ids[i]: dict({"Non-unique Gene name": "", "Annotation": "x"},
                            **{s: int(genes[i, j]) for j, s in enumerate(strains)}) for i in pick}
    traitsdic = {names[0]: {s: str(int(traits[0, j])) for j, s in enumerate(strains)}}
    res = m.Setup_results(genedic, traitsdic, False)
    row = res["Results"][names[0]]["TetRCG"]
    assert (row["tpgp"], row["tngp"], row["tpgn"], row["tngn"]) == (29, 8, 3, 60)
    assert abs(row["p_v"] - 1.0862106610751687e-14) < 1e-15


def test_perform_statistics_call_site(exampledir):
    from scoary_amd import methods as m
    gd, td = _load(exampledir)
    st = m.Perform_statistics(td["Tetracycline_resistance"], gd["Roarydic"]["TetRCG"])
    assert st["statistics"] == {"tpgp": 29, "tpgn": 3, "tngp": 8, "tngn": 60}
    assert sorted(set(st["gene_trait"].values())) == ["AB", "Ab", "aB", "ab"]
    assert len(st["gene_trait"]) == 100


def test_cli_permute_fisher_empirical_p(exampledir, tmp_path):
    """--permute with --no_pairwise (documented extension): the
    Empirical_p column follows Benjamini_H_p and equals (r+1)/(P+1) with r from
    the oracle on the same (seed, trait, permutation) counters."""
    from oracle import oracle as orc
    from scoary_amd.engine import pack_bits_rows
    P, seed = 200, 1234
    files = run_cli(_inputs(exampledir) + ["--no_pairwise", "-e", str(P), "--seed", str(seed),
                                           "-c", "I", "P", "-p", "0.05", "1.0"], tmp_path)
    ids, strains, genes, names, traits = read_dense(
        golden_text("exampledata/Gene_presence_absence.csv.gz"),
        golden_text("exampledata/Tetracycline_resistance.csv.gz"))
    gb = orc.pack_rows(genes)
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    r = orc.permute_r(gb, tb, mb, len(strains), P, seed)
    for t, trait in enumerate(names):
        rows = list(csv.reader(io.StringIO(files[trait + ".results.csv"])))
        assert rows[0][12] == "Benjamini_H_p" and rows[0][13] == "Empirical_p"
        assert len(rows) > 10
        for d in rows[1:]:
            want = (float(r[ids.index(d[0]), t]) + 1.0) / (P + 1.0)
            assert d[13] == repr(want), (trait, d[0], d[13], want)


def test_cli_permute_early_abort_flag(exampledir, tmp_path):
    """--permute-early-abort: the Fisher-statistic Empirical_p column follows the
    reference's sequential rule (scoary/methods.py:1360-1363) -- genes whose running count
    crosses the binomial bound stop early and get the coarse (r+1)/(i+2), the significant
    ones still get (r+1)/(P+1); without the flag nothing stops."""
    P, seed = 300, 99
    common = _inputs(exampledir) + ["--no_pairwise", "-e", str(P), "--seed", str(seed),
                                    "-c", "I", "-p", "1.0"]
    plain = run_cli(common, tmp_path / "a")
    abort = run_cli(common + ["--permute-early-abort"], tmp_path / "b")
    rows_p = list(csv.reader(io.StringIO(plain["Tetracycline_resistance.results.csv"])))
    rows_a = list(csv.reader(io.StringIO(abort["Tetracycline_resistance.results.csv"])))
    assert rows_p[0] == rows_a[0] and rows_a[0][13] == "Empirical_p"
    ep = {d[0]: float(d[13]) for d in rows_p[1:]}
    ea = {d[0]: float(d[13]) for d in rows_a[1:]}
    assert set(ep) == set(ea)
    grid = {(r + 1.0) / (P + 1.0) for r in range(P + 1)}
    assert all(v in grid for v in ep.values())                    # fixed-P estimator
    coarse = [g for g, v in ea.items() if v not in grid]
    assert len(coarse) > 100                                       # most genes stop early
    assert all(ea[g] > 0.1 for g in coarse)                        # only unpromising ones do
    top = rows_p[1][0]                                             # the best hit never stops
    assert ea[top] == ep[top] == 1.0 / (P + 1.0)
    for d in rows_a[1:]:                                           # every other column unchanged
        assert d[:13] == next(x for x in rows_p[1:] if x[0] == d[0])[:13]


def test_permute_call_site_matches_oracle(exampledir):
    from oracle import oracle as orc
    from scoary_amd import methods as m
    from scoary_amd.engine import pack_bits_rows
    gd, td = _load(exampledir)
    res = m.Setup_results(gd["Roarydic"], td, False)
    gtc = res["Gene_trait_combinations"]["Bogus_trait"]["gene_8004"]
    assert len(gtc) == 97                                  # missing isolates are not in GTC
    emp = m.Permute(tree=None, GTC=gtc, permutations=150, cutoffs={"I": 0.05}, seed=9)
    strains = list(gtc)
    g = np.array([[1 if gtc[s][0] == "A" else 0 for s in strains]], dtype=np.uint8)
    t = np.array([[1 if gtc[s][1] == "B" else 0 for s in strains]], dtype=np.uint8)
    r = orc.permute_r(orc.pack_rows(g), pack_bits_rows(t), pack_bits_rows(np.ones_like(t)),
                      len(strains), 150, 9)
    assert emp == (float(r[0, 0]) + 1.0) / 151.0
    assert m.Permute(None, gtc, 5, {}) is None


def test_collapse_device_hash_equals_exact_grouping(monkeypatch):
    """--collapse grouping from the device hash (scoary_row_hash) == exact
    grouping on the bit rows, incl. chains of 3+ identical genes, genes that
    differ only at missing isolates, and the reference's ordering rules."""
    from scoary_amd import methods as m
    rng = np.random.default_rng(3)
    G, N = 400, 70
    genes = (rng.random((G, N)) < 0.5).astype(np.uint8)
    genes[10] = genes[3]; genes[200] = genes[3]; genes[57] = genes[56]
    genes[300] = genes[299]; genes[300, 5] ^= 1        # differs only at isolate 5
    strains = ["s%d" % i for i in range(N)]
    genedic = {"g%d" % i: dict({"Non-unique Gene name": "n%d" % i, "Annotation": "a%d" % i},
                               **{s: int(genes[i, j]) for j, s in enumerate(strains)})
               for i in range(G)}
    t0 = {s: str(int(rng.random() < 0.4)) for s in strains}
    t1 = {s: v for s, v in t0.items() if s != "s5"}    # isolate 5 missing for trait 1
    res = m.Setup_results(genedic, {"full": t0, "holes": t1}, True)
    names0 = list(res["Results"]["full"])
    assert "g3--g10--g200" in names0 and "g56--g57" in names0 and "g299--g300" not in names0
    assert names0[-1] == "g3--g10--g200" or names0.index("g3--g10--g200") > names0.index("g199")
    names1 = list(res["Results"]["holes"])
    assert "g299--g300" in names1                       # identical once isolate 5 is masked
    r = res["Results"]["full"]["g3--g10--g200"]
    assert r["NUGN"] == "n3--n10--n200" and r["Annotation"] == "a3--a10--a200"
    # exact grouping (no hashes) gives the same thing
    monkeypatch.setattr(m, "_pattern_groups",
                        lambda table, maskrow, idx, hashes, _f=m._pattern_groups:
                        _f(table, maskrow, idx, None))
    res2 = m.Setup_results(genedic, {"full": t0, "holes": t1}, True)
    for tr in ("full", "holes"):
        assert list(res2["Results"][tr]) == list(res["Results"][tr])
        assert np.array_equal(res2["Results"][tr].column("BH_p"), res["Results"][tr].column("BH_p"))
        assert res2["Results"][tr].number_of_tests == res["Results"][tr].number_of_tests


def test_cli_custom_newick_tree_equals_upgma_run(exampledir, tmp_path):
    """-n <tree> (the reference's Travis "Test4" flag): feeding the shipped
    ExampleTree.nwk -- which IS the UPGMA tree of exampledata -- must reproduce
    the default-mode files."""
    a = tmp_path / "a"
    b = tmp_path / "b"
    fa = run_cli(_inputs(exampledir), a)
    fb = run_cli(_inputs(exampledir) + ["-n", os.path.join(exampledir, "ExampleTree.nwk")], b)
    assert fa.keys() == fb.keys() and len(fa) == 2
    for k in fa:
        assert fa[k] == fb[k]
    top = list(csv.reader(io.StringIO(fb["Tetracycline_resistance.results.csv"])))[1]
    assert top[0] == "TetRCG" and [int(x) for x in top[13:16]] == [25, 25, 1]


@pytest.fixture(scope="module")
def synth2dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("synth2")
    for fn in ("gpa.csv", "traits.csv"):
        with open(os.path.join(d, fn), "w", newline="") as f:
            f.write(golden_text("synth2/%s.gz" % fn))
    return str(d)


@pytest.mark.parametrize("sub,extra", [
    ("no_pairwise", ["--no_pairwise", "-p", "1.0"]),
    ("collapse", ["--no_pairwise", "--collapse", "-c", "I", "B", "-p", "0.5", "1.0"]),
    ("pairwise", ["-u", "-c", "I", "EPW", "-p", "0.3", "1.0"]),
])
def test_cli_second_dataset_vs_reference(synth2dir, tmp_path, sub, extra):
    """tests/golden/synth2: clade-structured genes, NA / absent isolates, quoted
    cells; --no_pairwise, --collapse and default (pairwise) mode against the CSVs
    and the Tree.nwk the reference wrote for the same files."""
    files = run_cli(["-g", os.path.join(synth2dir, "gpa.csv"),
                     "-t", os.path.join(synth2dir, "traits.csv")] + extra, tmp_path)
    assert sorted(files) == ["driven.results.csv", "lineage.results.csv", "sparse.results.csv"]
    if "-u" in extra:
        with open(os.path.join(tmp_path, "Tree.nwk")) as f:
            assert f.read().strip() == golden_text("synth2/Tree.nwk.gz").strip()
    for fn, text in files.items():
        want = golden_text("synth2/%s/%s.gz" % (sub, fn))
        if sub != "pairwise":
            _assert_csv_equal(text, want)
            continue
        g = list(csv.reader(io.StringIO(text)))
        w = list(csv.reader(io.StringIO(want)))
        assert g[0] == w[0] and len(g) == len(w), (fn, len(g), len(w))
        wi = {r[0]: r for r in w[1:]}
        for r in g[1:]:
            x = wi[r[0]]
            assert r[:7] == x[:7] and r[13:16] == x[13:16], r[0]   # counts and pair counts: exact
            for k in (7, 8, 9, 10, 11, 12, 16, 17):
                if r[k] != x[k]:
                    assert abs(float(r[k]) - float(x[k])) <= 1e-9 * abs(float(x[k])) + 1e-12 * 300, \
                        (r[0], g[0][k], r[k], x[k])


def test_cli_restrict_write_reduced_and_input_columns(synth2dir, tmp_path):
    """-r/-w/--include_input_columns: results and the reduced gene table written by
    -w equal the reference's files."""
    restrict = os.path.join(synth2dir, "restrict.csv")
    with open(restrict, "w") as f:
        f.write(golden_text("synth2/restrict.csv.gz"))
    files = run_cli(["-g", os.path.join(synth2dir, "gpa.csv"), "-t", os.path.join(synth2dir, "traits.csv"),
                     "--no_pairwise", "-p", "1.0", "-r", restrict, "-w",
                     "--include_input_columns", "4,6-7"], tmp_path)
    for fn, text in files.items():
        want = golden_text("synth2/restrict_w/%s.gz" % fn)
        assert text.splitlines()[0] == want.splitlines()[0]         # incl. the three extra columns
        _assert_csv_equal(text, want)
    with open(os.path.join(tmp_path, "gene_presence_absence_reduced.csv"), newline="") as f:
        got = list(csv.reader(f))
    want = list(csv.reader(io.StringIO(
        golden_text("synth2/restrict_w/gene_presence_absence_reduced.csv.gz"))))
    assert got == want


def test_cli_semicolon_delimiter_and_max_hits(synth2dir, tmp_path):
    for name in ("gpa_semi.csv", "traits_semi.csv"):
        with open(os.path.join(synth2dir, name), "w", newline="") as f:
            f.write(golden_text("synth2/%s.gz" % name))
    files = run_cli(["-g", os.path.join(synth2dir, "gpa_semi.csv"),
                     "-t", os.path.join(synth2dir, "traits_semi.csv"), "--no_pairwise", "-p", "0.2",
                     "--delimiter", ";", "-m", "40"], tmp_path)
    assert len(files) == 3
    for fn, text in files.items():
        want = golden_text("synth2/semicolon/%s.gz" % fn)
        assert text.splitlines()[0] == want.splitlines()[0] and '";"' in text.splitlines()[0]
        g = list(csv.reader(io.StringIO(text), delimiter=";"))
        w = list(csv.reader(io.StringIO(want), delimiter=";"))
        assert len(g) == len(w) <= 41 and g[0] == w[0]
        wi = {r[0]: r for r in w[1:]}               # rows with (near-)equal p may swap places
        for a in g[1:]:
            b = wi[a[0]]
            assert a[:7] == b[:7]
            for x, y in zip(a[7:], b[7:]):
                assert x == y or abs(float(x) - float(y)) <= 1e-12 * 300 + 1e-11 * abs(float(y))
        ps = [float(r[10]) for r in g[1:]]
        assert all(ps[i] <= ps[i + 1] * (1 + 1e-9) for i in range(len(ps) - 1))


def test_cli_vcf_table_with_start_col(exampledir, tmp_path, manifest):
    """vcf2scoary output through the command line (-s, non-Roary identifiers)."""
    files = run_cli(["-g", os.path.join(exampledir, "mutations_presence_absence.csv"),
                     "-t", os.path.join(exampledir, "ExampleVCFTrait.csv"), "--no_pairwise",
                     "-p", "1.0", "-s", str(manifest["vcf_startcol_1based"])], tmp_path)
    assert sorted(files) == manifest["vcf_cli_files"]
    for fn, text in files.items():
        want = golden_text("csv_vcf_cli/%s.gz" % fn)
        g, w = text.splitlines(), want.splitlines()
        assert g[0] == w[0] and len(g) == len(w)
        # all rows tie at p = 1 / 0.5 here; compare as sets of rows
        assert sorted(g[1:]) == sorted(w[1:])


def test_cli_near_tie_dataset_vs_reference(tmp_path):
    """tests/golden/neartie: the closest near-tie of the census as a data set, run through the
    REFERENCE (SciPy 1.15.3) by make_golden.py.  Under trait A the genes nt_x / nt_y sit on two
    support points whose weights differ by 1.7e-12: SciPy -- the reference -- gives them
    different p-values (0.6002 / 0.6334); a tie window wider than that gap would give both
    0.6334.  Every cell of both CSVs within the usual tolerance."""
    d = tmp_path / "in"
    os.makedirs(d)
    for fn in ("gpa.csv", "traits.csv"):
        with open(d / fn, "w", newline="") as f:
            f.write(golden_text("neartie/%s.gz" % fn))
    files = run_cli(["-g", str(d / "gpa.csv"), "-t", str(d / "traits.csv"), "--no_pairwise",
                     "-p", "1.0"], tmp_path / "out")
    for trait, pair in (("A", ("nt_x", "nt_y")), ("B", ("nt_u", "nt_v"))):
        got, want = files[trait + ".results.csv"], golden_text("neartie/%s.results.csv.gz" % trait)
        _assert_csv_equal(got, want)
        rows = {r[0]: r for r in csv.reader(io.StringIO(got))}
        px, py = float(rows[pair[0]][10]), float(rows[pair[1]][10])
        assert abs(px - 0.6002069089625) < 1e-12 and abs(py - 0.6333635751133) < 1e-12


def test_cli_cfg2_sized_table_every_row_vs_oracle(tmp_path):
    """The command line on a table of BASELINE configs[1]'s shape (10 000 genes x 500
    isolates, 1000 permutations, here with 3 traits and missing values): every
    output row -- counts, sensitivity / specificity, Naive_p, corrected p and
    Empirical_p -- against the oracle fed by the independent reader, and the row
    order against the reference's sort keys.  Exercises the native reader, the
    list-driven permutation path and the results writer at a size the example data
    does not reach."""
    from oracle import oracle as orc
    from scoary_amd.engine import pack_bits_rows
    rng = np.random.default_rng(20)
    G, N, T, P, seed = 10_000, 500, 3, 1000, 77
    dense = rng.random((G, N)) < rng.beta(0.5, 0.5, (G, 1))
    dense[11] = rng.random(N) < 0.4
    lab = np.where(rng.random((T, N)) < 0.45, "1", "0").astype(object)
    lab[0] = np.where(dense[11] ^ (rng.random(N) < 0.04), "1", "0")
    lab[1, rng.random(N) < 0.1] = "NA"
    iso = ["s%03d" % i for i in range(N)]
    meta = ["Gene", "Non-unique Gene name", "Annotation", "No. isolates", "No. sequences",
            "Avg sequences per isolate", "Genome Fragment", "Order within Fragment",
            "Accessory Fragment", "Accessory Order with Fragment", "QC", "Min group size nuc",
            "Max group size nuc", "Avg group size nuc"]
    cells = np.where(dense, "x", "")
    gtxt = ",".join(meta + iso) + "\n" + "".join(
        "g%05d,,\"syn, thetic\",%s,%s\n" % (g, ",".join(["1"] * 11), ",".join(cells[g]))
        for g in range(G))
    ttxt = "," + ",".join("tr%d" % t for t in range(T)) + "\n" + "".join(
        iso[i] + "," + ",".join(lab[:, i]) + "\n" for i in range(N))
    (tmp_path / "g.csv").write_text(gtxt)
    (tmp_path / "t.csv").write_text(ttxt)
    files = run_cli(["-g", str(tmp_path / "g.csv"), "-t", str(tmp_path / "t.csv"),
                     "--no_pairwise", "-e", str(P), "--seed", str(seed), "-p", "1.0"],
                    tmp_path / "out")
    ids, strains, genes, names, traits = read_dense(gtxt, ttxt)
    assert strains == iso and len(ids) == G
    gb = orc.pack_rows(genes)
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    r = orc.permute_r(gb, tb, mb, N, P, seed)
    cnt_all = orc.counts_packed(gb, tb, mb)
    for t, trait in enumerate(names):
        rows = list(csv.reader(io.StringIO(files[trait + ".results.csv"])))
        col = {c: k for k, c in enumerate(rows[0])}
        cnt = cnt_all[:, t]                            # tpgp, tpgn, tngp, tngn
        odds, p = orc.fisher_many(cnt)
        keep = [g for g in range(G) if cnt[g, 0] + cnt[g, 2] and cnt[g, 1] + cnt[g, 3]]
        assert len(rows) - 1 == len(keep)             # all-present / all-absent genes drop out
        seen = set()
        for d in rows[1:]:
            g = int(d[0][1:])
            seen.add(g)
            assert d[2] == "syn, thetic"
            assert [int(d[col[c]]) for c in
                    ("Number_pos_present_in", "Number_neg_present_in",
                     "Number_pos_not_present_in", "Number_neg_not_present_in")] == \
                [cnt[g, 0], cnt[g, 2], cnt[g, 1], cnt[g, 3]]
            assert abs(float(d[col["Naive_p"]]) - p[g]) <= 1e-12 + 1e-11 * p[g]
            assert d[col["Empirical_p"]] == repr((float(r[g, t]) + 1.0) / (P + 1.0))
        assert seen == set(keep)
        naive = [float(d[col["Naive_p"]]) for d in rows[1:]]
        assert all(a <= b * (1 + 1e-9) for a, b in zip(naive, naive[1:]))
        bonf = [float(d[col["Bonferroni_p"]]) for d in rows[1:]]
        assert all(abs(b - min(1.0, q * len(keep))) <= 1e-9 * max(b, 1e-300)
                   for b, q in zip(bonf, naive))
    top = list(csv.reader(io.StringIO(files["tr0.results.csv"])))[1]
    assert top[0] == "g00011"                          # the planted gene wins its trait


@pytest.mark.parametrize("collapse", [False, True])
def test_p_order_from_the_device_equals_numpy(exampledir, monkeypatch, collapse):
    """Large results take the stable p order of Benjamini-Hochberg (scoary/methods.py:903-925) from one
    batched torch.argsort on the device records instead of numpy on the host: forced on for the
    example data, every column and the writer's row order must be what the host path gives
    (ties -- the example data has many equal p -- included)."""
    from scoary_amd import methods as m
    gd, td = _load(exampledir)
    monkeypatch.setattr(m, "DEVICE_SORT_MIN_PAIRS", 1 << 60)
    host = m.Setup_results(gd["Roarydic"], td, collapse)["Results"]
    monkeypatch.setattr(m, "DEVICE_SORT_MIN_PAIRS", 0)
    dev = m.Setup_results(gd["Roarydic"], td, collapse)["Results"]
    for trait in host:
        assert list(dev[trait]) == list(host[trait])
        for k in ("p_v", "B_p", "BH_p", "tpgp", "sens"):
            assert np.array_equal(dev[trait].column(k), host[trait].column(k)), (trait, k)
        if not collapse:
            assert np.array_equal(dev[trait].p_order, host[trait].p_order)


@pytest.mark.parametrize("collapse", [False, True])
def test_per_trait_statistics_on_worker_threads_equal_the_loop(exampledir, monkeypatch, collapse):
    """Round 6: the per-trait host statistics (skip rule, B / BH, columns; scoary/methods.py:781-925)
    of a large run go through a thread pool, one trait per worker.  Forced on for the example data:
    the same rows, columns and p order as the plain loop, in trait order."""
    from scoary_amd import methods as m
    gd, td = _load(exampledir)
    monkeypatch.setattr(m, "HOST_THREADS_MIN_PAIRS", 1 << 60)
    loop = m.Setup_results(gd["Roarydic"], td, collapse)["Results"]
    monkeypatch.setattr(m, "HOST_THREADS_MIN_PAIRS", 0)
    monkeypatch.setattr(m, "_usable_cpus", lambda: 4)
    pool = m.Setup_results(gd["Roarydic"], td, collapse)["Results"]
    assert list(pool) == list(loop)
    for trait in loop:
        assert list(pool[trait]) == list(loop[trait]) and pool[trait].number_of_tests == loop[trait].number_of_tests
        for k in ("p_v", "B_p", "BH_p", "tpgp", "tngn", "sens", "spes", "OR"):
            assert np.array_equal(pool[trait].column(k), loop[trait].column(k), equal_nan=True), (trait, k)


def test_cli_vcf_pipeline_every_row_vs_oracle(tmp_path):
    """BASELINE configs[3] as it is literally defined -- "VCF-derived" -- at 20 000 sites x 1000
    isolates: a synthetic haploid VCF (rare variants, ~3 % multi-allelic sites, a few missing
    calls; tools/e2e_vcf.py writes the same file at 200 000 x 5000) -> scoary_amd.vcf2scoary ->
    the command line with -s 11 (nine fixed columns + DUMMY), --no_pairwise --permute, and every
    output row against the oracle fed by an INDEPENDENT reading of the VCF that applies the
    reference's rules itself: one table row per ALT allele (scoary/vcf2scoary.py:186-214), the
    identifier CHROM_|_POS_|_ID (scoary/methods.py:437-460) -- so the rows of a multi-allelic
    site collide and the last allele's row replaces the earlier ones in place (dict overwrite) --
    and a missing call "." that is written through at bi-allelic sites and therefore READS AS
    PRESENT, but is "0" at multi-allelic ones."""
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN), "..", "tools"))
    import e2e_vcf
    from oracle import oracle as orc
    from scoary_amd import vcf2scoary
    from scoary_amd.engine import pack_bits_rows
    sites, N, P, seed = 20_000, 1000, 500, 31
    rng = np.random.default_rng(9)
    vcf, table = str(tmp_path / "in.vcf"), str(tmp_path / "mutations.csv")
    nrows, names = e2e_vcf.write_vcf(vcf, sites, N, rng, multi_frac=0.03, missing_frac=0.001)
    lab = rng.random(N) < 0.3
    with open(tmp_path / "traits.csv", "w") as f:
        f.write(",resistance\n" + "".join("%s,%d\n" % (s, v) for s, v in zip(names, lab)))
    assert vcf2scoary.convert_file(vcf, table, log=lambda *a: None) == nrows > sites
    files = run_cli(["-g", table, "-t", str(tmp_path / "traits.csv"), "-s", "11", "--no_pairwise",
                     "-e", str(P), "--seed", str(seed), "-p", "1.0"], tmp_path / "out")
    # independent reading of the VCF
    dense = np.zeros((sites, N), dtype=np.uint8)
    ident = []
    with open(vcf) as f:
        k = 0
        for line in f:
            if line.startswith("#"):
                continue
            q = line.rstrip("\n").split("\t")
            alts = q[4].split(",")
            if len(alts) == 1:
                dense[k] = [0 if g in ("", "0", "-") else 1 for g in q[9:]]       # "." reads as present
            else:
                last = len(alts)                                                  # the last allele's row survives
                dense[k] = [1 if g == str(last) else 0 for g in q[9:]]
            ident.append((q[0], q[1], q[2]))
            k += 1
    assert k == sites
    gb = orc.pack_rows(dense)
    tb = pack_bits_rows(lab[None].astype(np.uint8))
    mb = pack_bits_rows(np.ones((1, N), dtype=np.uint8))
    r = orc.permute_r(gb, tb, mb, N, P, seed)
    cnt = orc.counts_packed(gb, tb, mb)[:, 0]
    _, p = orc.fisher_many(cnt)
    rows = list(csv.reader(io.StringIO(files["resistance.results.csv"])))
    col = {c: i for i, c in enumerate(rows[0])}
    assert rows[0][:3] == ["#CHROM", "POS", "ID"]
    keep = [g for g in range(sites) if cnt[g, 0] + cnt[g, 2] and cnt[g, 1] + cnt[g, 3]]
    assert len(rows) - 1 == len(keep) > sites // 2
    seen = set()
    for d in rows[1:]:
        g = int(d[2])                                    # the ID column is the site number
        seen.add(g)
        assert (d[0], d[1], d[2]) == ident[g]
        assert [int(d[col[c]]) for c in
                ("Number_pos_present_in", "Number_neg_present_in",
                 "Number_pos_not_present_in", "Number_neg_not_present_in")] == \
            [cnt[g, 0], cnt[g, 2], cnt[g, 1], cnt[g, 3]]
        assert abs(float(d[col["Naive_p"]]) - p[g]) <= 1e-12 + 1e-11 * p[g]
        assert d[col["Empirical_p"]] == repr((float(r[g, 0]) + 1.0) / (P + 1.0))
    assert seen == set(keep)
    naive = [float(d[col["Naive_p"]]) for d in rows[1:]]
    assert all(a <= b * (1 + 1e-9) for a, b in zip(naive, naive[1:]))


def test_cli_wide_table_takes_the_segmented_list_path(tmp_path, caplog):
    """The command line on a table WIDER than one LDS label tile (41 000 isolates > 20 479, and
    past the 40 959 at which the dense kernels used to take over; round 3): the list-driven
    permutation path stays in charge -- three isolate segments, k_permute_seglists -- and every output row (counts,
    Naive_p, Empirical_p) equals the oracle's.  Also the native reader on 41 014-column rows."""
    import logging
    from oracle import oracle as orc
    from scoary_amd import methods as M
    from scoary_amd.engine import pack_bits_rows
    rng = np.random.default_rng(41)
    G, N, T, P, seed = 60, 41_000, 2, 96, 5
    dense = rng.random((G, N)) < rng.beta(0.5, 0.5, (G, 1))
    dense[3] = rng.random(N) < 0.3
    lab = np.where(rng.random((T, N)) < 0.4, "1", "0").astype(object)
    lab[0] = np.where(dense[3] ^ (rng.random(N) < 0.05), "1", "0")
    lab[1, rng.random(N) < 0.02] = "NA"
    iso = ["s%05d" % i for i in range(N)]
    meta = ["Gene", "Non-unique Gene name", "Annotation", "No. isolates", "No. sequences",
            "Avg sequences per isolate", "Genome Fragment", "Order within Fragment",
            "Accessory Fragment", "Accessory Order with Fragment", "QC", "Min group size nuc",
            "Max group size nuc", "Avg group size nuc"]
    cells = np.where(dense, "x", "")
    gtxt = ",".join(meta + iso) + "\n" + "".join(
        "g%03d,,wide,%s,%s\n" % (g, ",".join(["1"] * 11), ",".join(cells[g])) for g in range(G))
    ttxt = "," + ",".join("tr%d" % t for t in range(T)) + "\n" + "".join(
        iso[i] + "," + ",".join(lab[:, i]) + "\n" for i in range(N))
    (tmp_path / "g.csv").write_text(gtxt)
    (tmp_path / "t.csv").write_text(ttxt)
    with caplog.at_level(logging.INFO, logger=M.log.name):
        files = run_cli(["-g", str(tmp_path / "g.csv"), "-t", str(tmp_path / "t.csv"),
                         "--no_pairwise", "-e", str(P), "--seed", str(seed), "-p", "1.0"],
                        tmp_path / "out")
    assert not any("dense permutation kernels" in r.getMessage() for r in caplog.records)
    ids, strains, genes, names, traits = read_dense(gtxt, ttxt)
    assert strains == iso and len(ids) == G
    gb = orc.pack_rows(genes)
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    r = orc.permute_r(gb, tb, mb, N, P, seed)
    cnt_all = orc.counts_packed(gb, tb, mb)
    for t, trait in enumerate(names):
        rows = list(csv.reader(io.StringIO(files[trait + ".results.csv"])))
        col = {c: k for k, c in enumerate(rows[0])}
        cnt = cnt_all[:, t]
        _, p = orc.fisher_many(cnt)
        keep = [g for g in range(G) if cnt[g, 0] + cnt[g, 2] and cnt[g, 1] + cnt[g, 3]]
        assert 50 <= len(keep) < G and len(rows) - 1 == len(keep)     # skip rule at this width too
        seen = set()
        for d in rows[1:]:
            g = int(d[0][1:])
            seen.add(g)
            assert [int(d[col[c]]) for c in
                    ("Number_pos_present_in", "Number_neg_present_in",
                     "Number_pos_not_present_in", "Number_neg_not_present_in")] == \
                [cnt[g, 0], cnt[g, 2], cnt[g, 1], cnt[g, 3]]
            assert abs(float(d[col["Naive_p"]]) - p[g]) <= 1e-12 + 1e-11 * p[g]
            assert d[col["Empirical_p"]] == repr((float(r[g, t]) + 1.0) / (P + 1.0))
        assert seen == set(keep)


def test_small_population_result_files_are_the_references_bytes(exampledir, synth2dir, tmp_path):
    """Up to 170 valid isolates k_fisher returns SciPy's own double (spec S3), so every file the reference wrote for
    its example data (N = 100; -r: 46) and for the second data set (N = 52) comes out of our command line BYTE FOR
    BYTE -- p-values, their corrections, the order of tied rows, the pair counts: --no_pairwise (four flag sets),
    the pairwise stage (three; its two binomial p columns are SciPy's doubles too), --collapse, -r / -w /
    --include_input_columns, ';' and -m."""
    ex = _inputs(exampledir)
    runs = [(ex + ["--no_pairwise", "-p", "1.0"], "csv_no_pairwise"),
            (ex + ["--no_pairwise"], "csv_no_pairwise_default"),
            (ex + ["--no_pairwise", "--collapse", "-c", "I", "BH", "-p", "0.05", "0.01", "-m", "50"],
             "csv_no_pairwise_collapse_bh"),
            (ex + ["--no_pairwise", "-r", os.path.join(exampledir, "Restrict_to.csv")], "csv_no_pairwise_restrict"),
            (ex + ["-u"], "csv_pairwise_default"),
            (ex + ["-c", "I", "EPW", "-p", "0.05", "0.05"], "csv_pairwise_epw"),
            (ex + ["-c", "BH", "PW", "-p", "0.9", "0.05", "-m", "300"], "csv_pairwise_bh_pw")]
    s2 = ["-g", os.path.join(synth2dir, "gpa.csv"), "-t", os.path.join(synth2dir, "traits.csv")]
    restrict = os.path.join(synth2dir, "restrict.csv")
    with open(restrict, "w") as f:
        f.write(golden_text("synth2/restrict.csv.gz"))
    for name in ("gpa_semi.csv", "traits_semi.csv"):
        with open(os.path.join(synth2dir, name), "w", newline="") as f:
            f.write(golden_text("synth2/%s.gz" % name))
    runs += [(s2 + ["--no_pairwise", "-p", "1.0"], "synth2/no_pairwise"),
             (s2 + ["--no_pairwise", "--collapse", "-c", "I", "B", "-p", "0.5", "1.0"], "synth2/collapse"),
             (s2 + ["-u", "-c", "I", "EPW", "-p", "0.3", "1.0"], "synth2/pairwise"),
             (s2 + ["--no_pairwise", "-p", "1.0", "-r", restrict, "-w", "--include_input_columns", "4,6-7"],
              "synth2/restrict_w"),
             (["-g", os.path.join(synth2dir, "gpa_semi.csv"), "-t", os.path.join(synth2dir, "traits_semi.csv"),
               "--no_pairwise", "-p", "0.2", "--delimiter", ";", "-m", "40"], "synth2/semicolon")]
    differing, files_checked = [], 0
    for k, (argv, sub) in enumerate(runs):
        files = run_cli(argv, tmp_path / ("run%d" % k))
        assert files
        for fn, text in files.items():
            files_checked += 1
            want = golden_text("%s/%s.gz" % (sub, fn))
            # (the two binomial p columns of the pairwise files included: tree.binom_two_sided restates the arithmetic
            # behind the reference's ss.binom_test -- 0.30175781249999994 where the exact tail is 0.3017578125)
            if text != want:
                differing.append("%s/%s" % (sub, fn))
    assert not differing and files_checked >= 29, differing
