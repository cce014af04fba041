"""Population-structure stage on the GPU: Hamming counts,
bit gather, PhyloTree maxima, tree-statistic permutations and the default-mode
command line, against the oracle and files captured from the real reference."""
import csv
import io
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, golden_text, read_dense

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from scoary_amd.engine import AssociationEngine
    e = AssociationEngine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


@pytest.mark.parametrize("R,N", [(2, 1), (3, 31), (17, 64), (100, 5940), (257, 1000), (40, 20001)])
def test_hamming_counts_bit_exact(eng, R, N):
    rng = np.random.default_rng(R * 31 + N)
    X = (rng.random((R, N)) < rng.uniform(0.1, 0.9)).astype(np.uint8)
    got = eng.hamming(X)
    want = (X[:, None, :] != X[None, :, :]).sum(axis=2) if R * R * N < 5e7 else \
        np.array([[int((X[i] != X[j]).sum()) for j in range(R)] for i in range(R)])
    assert np.array_equal(got, want)


def test_upgma_random_cases_and_exampledata(eng):
    from scoary_amd import tree as T
    d = _json("upgma_cases.json")
    for c in d["cases"]:
        X = np.array(c["matrix"], dtype=np.uint8)          # strains x genes
        t = T.upgma(eng, X.T, c["names"])
        assert str(t) == c["newick"]
    ids, strains, genes, names, traits = read_dense(
        golden_text("exampledata/Gene_presence_absence.csv.gz"),
        golden_text("exampledata/Tetracycline_resistance.csv.gz"))
    t = T.upgma(eng, genes, strains)
    assert T.newick_text(t) == golden_text("exampledata/ExampleTree.nwk.gz").strip()


def test_device_upgma_equals_host_loop_with_ties(eng):
    """scoary_upgma (device merge loop) == the numpy quad-tree loop on tie-heavy inputs
    (duplicated strains: many distance-0 and equal-distance cells), sizes around the
    2x2-block boundaries; the degenerate all-distances-1 case is handed back to the host."""
    from scoary_amd import tree as T
    for seed in range(30):
        rng = np.random.default_rng(100 + seed)
        n = int(rng.choice([2, 3, 4, 5, 8, 9, 16, 17, 33, 64, 65, 130, 257]))
        X = (rng.random((n, int(rng.choice([3, 8, 40, 300])))) < 0.3).astype(np.uint8)
        if seed % 2:
            X[n // 2:] = X[:n - n // 2]
        tot = X.sum(axis=0)
        var = X[:, (tot > 0) & (tot < n)]
        if var.shape[1] == 0:
            continue
        names = ["s%d" % i for i in range(n)]
        Xi = var.astype(np.int64)
        cnt = Xi @ (1 - Xi).T + (1 - Xi) @ Xi.T
        want = T.upgma_from_counts(cnt, var.shape[1], names, native=False)
        assert T.upgma(eng, var.T, names) == want
        m = eng.upgma_merges(var)
        if m is not None:                                 # the device loop itself, not the fallback
            cluster = list(names)
            for i, j in m.tolist():
                cluster[i], cluster[j] = [cluster[i], cluster[j]], None
            assert cluster[i] == want
    comp = np.array([[1, 0, 1, 0], [0, 1, 0, 1]], dtype=np.uint8)       # distance exactly 1
    assert eng.upgma_merges(comp) is None
    assert T.upgma(eng, comp.T, ["a", "b"]) == T.upgma_from_counts(
        np.array([[0, 4], [4, 0]]), 4, ["a", "b"], native=False)


@pytest.mark.parametrize("R,N,K", [(1, 1, 1), (5, 40, 40), (70, 100, 97), (300, 2000, 1980), (3, 5000, 4999)])
def test_gather_bits_bit_exact(eng, R, N, K):
    import torch
    from scoary_amd.engine import pack_bits_rows
    rng = np.random.default_rng(R + N + K)
    X = (rng.random((R, N)) < 0.5).astype(np.uint8)
    idx = rng.permutation(N)[:K].astype(np.int32)
    rows = eng.vecrows(pack_bits_rows(X), N)
    got = eng.gather_bits(rows, torch.from_numpy(idx).cuda()).cpu().numpy().view(np.uint32)
    bits = np.unpackbits(got.view(np.uint8), axis=1, bitorder="little")
    assert np.array_equal(bits[:, :K], X[:, idx])
    assert not bits[:, K:].any()


def _dp_via_gpu(eng, tree, tips_names, states_by_name):
    import torch
    from scoary_amd import tree as T
    from scoary_amd.engine import pack_bits_rows
    index_of = {t: i for i, t in enumerate(tips_names)}
    prog = T.TreeProgram(tree, index_of)
    K = prog.ntips
    g = np.array([[1 if states_by_name[tips_names[i]][0] == "A" else 0 for i in prog.tips]], np.uint8)
    l = np.array([[1 if states_by_name[tips_names[i]][1] == "B" else 0 for i in prog.tips]], np.uint8)
    Wt = (K + 31) // 32
    gb = torch.from_numpy(np.ascontiguousarray(pack_bits_rows(g).view(np.uint32)[:, :Wt]).view(np.int32)).cuda()
    lb = torch.from_numpy(np.ascontiguousarray(pack_bits_rows(l).view(np.uint32)[:, :Wt]).view(np.int32)).cuda()
    out = eng.tree_pairs(torch.from_numpy(prog.ops).cuda(), prog.depth, gb, lb, K)
    return tuple(int(x) for x in out.cpu().numpy()[0, 0])


def test_phylotree_maxima_golden_cases(eng):
    for c in _json("phylotree_cases.json"):
        tips = list(c["gtc"].keys())
        got = _dp_via_gpu(eng, c["tree"], tips, c["gtc"])
        w = c["result"]
        assert got == (w["Total"], w["Pro"], w["Anti"]), c["tree"]


def test_tree_pairs_many_genes_vs_oracle(eng, orc):
    """G genes x L label rows on random deep / bushy trees."""
    import torch
    from scoary_amd import tree as T
    from scoary_amd.engine import pack_bits_rows
    rng = np.random.default_rng(5)

    def rand_tree(tips, cat):
        if len(tips) == 1:
            return tips[0]
        k = 1 if rng.random() < cat else int(rng.integers(1, len(tips)))
        return [rand_tree(tips[:k], cat), rand_tree(tips[k:], cat)]
    sys.setrecursionlimit(10000)
    for K, cat in [(2, 0), (9, 0.2), (64, 0.5), (333, 0.9), (700, 0.1), (1500, 0.97)]:
        names = ["t%d" % i for i in range(K)]
        tree = rand_tree(names, cat)
        prog = T.TreeProgram(tree, {t: i for i, t in enumerate(names)})
        assert prog.depth <= int(np.log2(K)) + 2
        G, L = 37, 11
        gm = (rng.random((G, K)) < rng.uniform(0.05, 0.95, (G, 1))).astype(np.uint8)
        lm = (rng.random((L, K)) < rng.uniform(0.2, 0.8, (L, 1))).astype(np.uint8)
        Wt = (K + 31) // 32                      # row stride the kernel expects

        def rows32(m01):
            w = pack_bits_rows(m01).view(np.uint32)[:, :Wt]
            return torch.from_numpy(np.ascontiguousarray(w).view(np.int32)).cuda()
        gb, lb = rows32(gm[:, prog.tips]), rows32(lm[:, prog.tips])
        out = eng.tree_pairs(torch.from_numpy(prog.ops).cuda(), prog.depth, gb, lb, K).cpu().numpy()
        ops, tips = orc.tree_program(tree, {t: i for i, t in enumerate(names)})
        for g in range(0, G, 5):
            for l in range(0, L, 3):
                st = np.array([(0 if gm[g, i] else 2) + (0 if lm[l, i] else 1) for i in tips],
                              dtype=np.uint8)
                assert tuple(out[g, l]) == orc.tree_dp(ops, st), (K, g, l)


def test_tree_permute_call_site_vs_reference_permute(eng, manifest):
    """methods.Permute(tree=...) == the reference's Permute() fed the same
    spec-S4 label permutations (golden, incl. its early abort)."""
    from scoary_amd import methods as m
    from scoary_amd import tree as T
    gold = _json("permute_tree_s4.json")
    tree = eval(_json("upgma_cases.json")["exampledata_tree"])
    ids, strains, genes, names, traits = read_dense(
        golden_text("exampledata/Gene_presence_absence.csv.gz"),
        golden_text("exampledata/Tetracycline_resistance.csv.gz"))
    for ti, trait in enumerate(names):
        ptree = T.prune_missing(tree, manifest["prune"][trait] + [None])
        for rec in gold[trait]:
            if rec["empirical_p"] is None:
                continue
            g = ids.index(rec["gene"])
            # the Permute() call site works on the trait's valid isolates only, so its
            # permutation counters see a different isolate numbering than the batched
            # CLI path: compare through the batched stage instead (same numbering as
            # the golden run), then the call-site signature separately below.
            stage = m._TreeStage(eng, ptree, strains, traits[ti], ti, rec["seed"])
            from scoary_amd.engine import pack_bits_rows
            rows = pack_bits_rows(genes[g:g + 1])
            obs = stage.observed(rows)
            w = rec["observed"]
            assert tuple(int(x) for x in obs[0]) == (w["Total"], w["Pro"], w["Anti"])
            ex = stage.permute(rows, obs, rec["P"])
            assert T.empirical_p_sequential(ex[0]) == rec["empirical_p"], rec
    # call-site signature: GTC dict in, float out, tree statistic when a tree is given
    gtc = {s: ("A" if genes[ids.index("TetRCG"), j] else "a") + ("B" if traits[0, j] == 1 else "b")
           for j, s in enumerate(strains)}
    emp = m.Permute(tree=tree, GTC=gtc, permutations=100, cutoffs={"I": 0.05}, seed=7,
                    trait_index=0)
    assert emp == [r for r in gold["Tetracycline_resistance"]
                   if r["gene"] == "TetRCG" and r["P"] == 100][0]["empirical_p"]


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
        if fn.endswith(".results.csv") or fn.endswith(".nwk"):
            with open(os.path.join(outdir, fn), newline="") as f:
                out[fn] = f.read()
    return out


@pytest.mark.parametrize("sub,extra", [
    ("csv_pairwise_default", ["-u"]),
    ("csv_pairwise_epw", ["-c", "I", "EPW", "-p", "0.05", "0.05"]),
    ("csv_pairwise_bh_pw", ["-c", "BH", "PW", "-p", "0.9", "0.05", "-m", "300"]),
])
def test_cli_default_pairwise_mode_vs_reference(exampledir, tmp_path, sub, extra):
    """The reference's Travis "Test1"-style runs (pairwise comparisons on)."""
    files = run_cli(["-g", os.path.join(exampledir, "Gene_presence_absence.csv"),
                     "-t", os.path.join(exampledir, "Tetracycline_resistance.csv")] + extra,
                    tmp_path)
    if "-u" in extra:
        assert files["Tree.nwk"].strip() == golden_text("exampledata/ExampleTree.nwk.gz").strip()
    for trait in ("Tetracycline_resistance", "Bogus_trait"):
        got = list(csv.reader(io.StringIO(files[trait + ".results.csv"])))
        want = list(csv.reader(io.StringIO(golden_text("%s/%s.results.csv.gz" % (sub, trait)))))
        assert got[0] == want[0]
        assert len(got) == len(want), (trait, len(got), len(want))
        # same ORDER of the rows that are exactly tied -- identical 2x2 tables, hence identical p in
        # either implementation (with --threads the reference's thread weave decides their order,
        # scoary/methods.py:1115-1122).  Rows of DIFFERENT tables whose p is mathematically equal
        # (a gene and its complement) sit one ulp apart in SciPy in an arbitrary direction; their
        # mutual order is rounding noise and is not compared.
        def by_table(rows):
            out = {}
            for r in rows[1:]:
                out.setdefault(tuple(r[3:7]), []).append(r[0])
            return out
        assert by_table(got) == by_table(want), trait
        gp = [float(r[10]) for r in got[1:]]
        if "EPW" not in extra and "PW" not in extra:
            assert all(gp[i] <= gp[i + 1] * (1 + 1e-9) for i in range(len(gp) - 1))
        wi = {r[0]: r for r in want[1:]}
        for r in got[1:]:
            w = wi[r[0]]
            assert r[:7] == w[:7]                                  # names + counts
            assert r[13:16] == w[13:16], r[0]                      # pair counts: exact
            for k in (7, 8, 9, 10, 11, 12, 16, 17):
                if r[k] == w[k]:
                    continue
                a, b = float(r[k]), float(w[k])
                assert abs(a - b) <= 1e-9 * abs(b) + 1e-12 * 6000, (r[0], got[0][k], r[k], w[k])
    # the reference's own pinned row (tests/test_scoary_output.py:12-14), pairwise part
    top = list(csv.reader(io.StringIO(files["Tetracycline_resistance.results.csv"])))[1]
    assert top[0] == "TetRCG" and [int(x) for x in top[13:16]] == [25, 25, 1]
    assert abs(float(top[16]) - 5.96046447754E-008) < 1e-9
    assert abs(float(top[17]) - 1.54972076416E-006) < 1e-7


def test_cli_pairwise_with_tree_permutations(exampledir, tmp_path):
    """--permute in default mode = tree-statistic permutations of the surviving
    genes (closes SURVEY D1): column present, values are the sequential
    estimator of the oracle's exceedance flags."""
    from oracle import oracle as orc
    P, seed = 120, 99
    files = run_cli(["-g", os.path.join(exampledir, "Gene_presence_absence.csv"),
                     "-t", os.path.join(exampledir, "Tetracycline_resistance.csv"),
                     "-e", str(P), "--seed", str(seed), "-p", "0.001"], tmp_path)
    ids, strains, genes, names, traits = read_dense(
        golden_text("exampledata/Gene_presence_absence.csv.gz"),
        golden_text("exampledata/Tetracycline_resistance.csv.gz"))
    tree = eval(_json("upgma_cases.json")["exampledata_tree"])
    index_of = {s: i for i, s in enumerate(strains)}
    gb = orc.pack_rows(genes)
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        manifest = json.load(f)
    checked = 0
    for ti, trait in enumerate(names):
        rows = list(csv.reader(io.StringIO(files[trait + ".results.csv"])))
        assert rows[0][18] == "Empirical_p"
        miss = manifest["prune"][trait]
        ptree = orc.prune_for_missing(tree, miss + [None]) if miss else tree
        ops, tips = orc.tree_program(ptree, index_of)
        tb = orc.pack_rows((traits[ti] == 1)[None].astype(np.uint8))[0]
        mb = orc.pack_rows((traits[ti] != 2)[None].astype(np.uint8))[0]
        for d in rows[1:8]:
            g = ids.index(d[0])
            obs, ex = orc.tree_permute(ops, tips, gb[g], tb, mb, len(strains), ti, P, seed)
            assert d[18] == repr(orc.empirical_p_with_abort(ex)), (trait, d[0])
            checked += 1
    assert checked >= 5


def test_tree_tip_limit_and_large_caterpillar(eng, orc):
    """pair counts are packed as total << 16 | count; with more than 32767 tips
    a reachable key could reach 2^30 and pass for "unreachable + reachable".  The largest
    accepted tree -- a 32767-tip caterpillar with alternating states, the shape that
    maximises the pair count (16383) -- equals the oracle, and 32768 tips are refused
    with SCOARY_ERR_SIZE instead of silently computing wrong maxima."""
    import ctypes
    import torch
    from scoary_amd.engine import pack_bits_rows
    K = 32767
    ops = np.array([0] + [-2 - i for i in range(1, K)], dtype=np.int32)   # push tip 0; merge-with-tip i
    g = (np.arange(K) % 2).astype(np.uint8)[None]                          # A a A a ...
    lab = ((np.arange(K) // 2) % 2).astype(np.uint8)[None]                 # b b B B ...
    state = np.where(g[0] == 1, np.where(lab[0] == 1, 0, 1), np.where(lab[0] == 1, 2, 3)).astype(np.uint8)
    orc_ops = np.zeros(2 * K - 1, dtype=np.int32)                          # oracle dialect: push / merge only
    orc_ops[1::2] = np.arange(1, K)
    orc_ops[2::2] = -1
    want = orc.tree_dp(orc_ops, state)
    assert want[0] > 8000                                                  # thousands of pairs: the packed range
    Wt = (K + 31) // 32
    gb = torch.from_numpy(np.ascontiguousarray(pack_bits_rows(g).view(np.uint32)[:, :Wt]).view(np.int32)).cuda()
    lb = torch.from_numpy(np.ascontiguousarray(pack_bits_rows(lab).view(np.uint32)[:, :Wt]).view(np.int32)).cuda()
    out = eng.tree_pairs(torch.from_numpy(ops).cuda(), 1, gb, lb, K)
    assert tuple(int(x) for x in out.cpu().numpy()[0, 0]) == want
    p = ctypes.c_void_p(gb.data_ptr())
    rc = eng.lib.scoary_tree_pairs(eng.h, p, 3, 1, p, p, 1, 1, 32768, p, None)
    assert rc == -3 and b"32767" in eng.lib.scoary_last_error(eng.h)


def test_thread_weave_tie_order_given_the_references_p_values(exampledir, tmp_path):
    """with --threads n the reference hands worker k the ranks k, k+n, ... and
    weaves the results back thread by thread (scoary/methods.py:1076-1078, 1115-1122), so the
    order of equal-p rows depends on n.  WHICH rank a gene has among mathematically tied rows
    is last-ulp noise of SciPy's pmf (a gene and its complement differ by one ulp in an
    arbitrary direction), so the comparison feeds the REFERENCE'S own p-values (golden
    --no_pairwise -p 1.0 CSVs) into StoreTraitResult: the whole row order must then equal the
    reference's `--threads 3` run, and differ from its single-threaded order."""
    from scoary_amd import methods as m, tree as T
    with open(os.path.join(exampledir, "Gene_presence_absence.csv"), "r", newline=None) as g, \
            open(os.path.join(exampledir, "Tetracycline_resistance.csv"), "r", newline=None) as t:
        gd = m.Csv_to_dic_Roary(g, ",", [], startcol=14)
        td, prune = m.Csv_to_dic(t, ",", None, gd["Strains"])
    res = m.Setup_results(gd["Roarydic"], td, False)
    upgma = T.upgma(m.get_engine(), gd["Zero_ones_matrix"].file_rows(), gd["Strains"])
    for trait in ("Tetracycline_resistance", "Bogus_trait"):
        ref = list(csv.reader(io.StringIO(golden_text("csv_no_pairwise/%s.results.csv.gz" % trait))))
        byname = {r[0]: r for r in ref[1:]}
        R = res["Results"][trait]
        assert set(R.genes) == set(byname)
        for col, k in (("p_v", 10), ("B_p", 11), ("BH_p", 12)):
            R.cols[col] = np.array([float(byname[gname][k]) for gname in R.genes])
    out = tmp_path / "w"
    os.makedirs(out)
    m.StoreResults(res["Results"], None, {"I": 0.05}, upgma, res["Gene_trait_combinations"], prune,
                   str(out) + "/", 0, 3, False, gd["Roarydic"], [], gd["Firstcolnames"])
    for trait in ("Tetracycline_resistance", "Bogus_trait"):
        with open(out / (trait + ".results.csv"), newline="") as f:
            got = [r[0] for r in csv.reader(f)][1:]
        want3 = [r[0] for r in csv.reader(io.StringIO(golden_text(
            "csv_pairwise_threads3/%s.results.csv.gz" % trait)))][1:]
        want1 = [r[0] for r in csv.reader(io.StringIO(golden_text(
            "csv_pairwise_default/%s.results.csv.gz" % trait)))][1:]
        assert got == want3, trait
        assert sorted(want1) == sorted(want3) and want1 != want3      # the weave does reorder ties
