"""Pin the oracle's population-structure restatements (UPGMA, PhyloTree maxima,
tree-statistic Permute, binomial test) against vectors captured from the real
reference (tests/golden/make_golden.py: tree_goldens)."""
import csv
import io
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden_text, read_dense
from oracle import oracle as orc


def _json(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def test_upgma_random_cases_incl_ties():
    d = _json("upgma_cases.json")
    assert len(d["cases"]) >= 35
    for c in d["cases"]:
        got = orc.upgma(np.array(c["matrix"]), c["names"])
        assert str(got) == c["newick"], (len(c["names"]), str(got), c["newick"])


@pytest.mark.timeout(600)
def test_upgma_exampledata_tree_is_the_shipped_tree(manifest):
    """ExampleTree.nwk (shipped with the reference) is the -u output of the
    reference on exampledata (manifest flag, checked at capture time)."""
    assert manifest["upgma_example_equals_shipped_tree"] and manifest["tree_nwk_equals_shipped"]
    ids, strains, genes, names, traits = read_dense(
        golden_text("exampledata/Gene_presence_absence.csv.gz"),
        golden_text("exampledata/Tetracycline_resistance.csv.gz"))
    tot = genes.sum(1)
    var = genes[(tot > 0) & (tot < genes.shape[1])]
    tree = orc.upgma(var.T, strains)
    assert orc.newick(tree) == golden_text("exampledata/ExampleTree.nwk.gz").strip()
    assert str(tree) == _json("upgma_cases.json")["exampledata_tree"]


def test_phylotree_maxima_random_trees():
    cases = _json("phylotree_cases.json")
    assert len(cases) >= 80
    for c in cases:
        tips = list(c["gtc"].keys())
        ops, tp = orc.tree_program(c["tree"], {t: i for i, t in enumerate(tips)})
        states = np.array([orc.TIP_STATE[c["gtc"][t]] for t in tips], dtype=np.uint8)
        got = orc.tree_dp(ops, states[tp])
        want = c["result"]
        assert got == (want["Total"], want["Pro"], want["Anti"]), c["tree"]


def test_binomial_test_values():
    # the two values the reference's own test pins (tests/test_scoary_output.py:13-14)
    assert orc.binom_two_sided_half(25, 25) == 5.960464477539063e-08
    assert orc.binom_two_sided_half(24, 25) == 1.5497207641601562e-06
    ss = pytest.importorskip("scipy.stats")
    for n in (1, 2, 7, 20, 33, 100, 401):
        for x in sorted({0, 1, n // 3, n // 2, (n + 1) // 2, n - 1, n}):
            want = ss.binomtest(x, n, 0.5).pvalue
            assert abs(orc.binom_two_sided_half(x, n) - want) <= 1e-12 * max(want, 1e-300)


def _example_tree():
    return eval(_json("upgma_cases.json")["exampledata_tree"])   # nested lists of names


def test_tree_permute_with_s4_labels_vs_reference_permute(manifest):
    """The reference's Permute(), fed (via a patched random.shuffle) exactly the
    label permutations spec S4 generates, including its early abort."""
    ids, strains, genes, names, traits = read_dense(
        golden_text("exampledata/Gene_presence_absence.csv.gz"),
        golden_text("exampledata/Tetracycline_resistance.csv.gz"))
    gold = _json("permute_tree_s4.json")
    tree = _example_tree()
    index_of = {s: i for i, s in enumerate(strains)}
    gb = orc.pack_rows(genes)
    N = len(strains)
    for ti, trait in enumerate(names):
        missing = manifest["prune"][trait]
        ptree = orc.prune_for_missing(tree, missing + [None]) if missing else tree
        ops, tips = orc.tree_program(ptree, index_of)
        assert len(tips) == N - len(missing)
        tb = orc.pack_rows((traits[ti] == 1)[None].astype(np.uint8))[0]
        mb = orc.pack_rows((traits[ti] != 2)[None].astype(np.uint8))[0]
        for rec in gold[trait]:
            g = ids.index(rec["gene"])
            obs, ex = orc.tree_permute(ops, tips, gb[g], tb, mb, N, ti, rec["P"], rec["seed"])
            w = rec["observed"]
            assert obs == (w["Total"], w["Pro"], w["Anti"]), rec["gene"]
            if rec["empirical_p"] is None:
                continue
            assert orc.empirical_p_with_abort(ex) == rec["empirical_p"], rec


def test_pairwise_columns_of_reference_csv():
    """Max_Pairwise_comparisons / supporting / opposing / best / worst p of the
    reference's default-mode CSV, recomputed by the oracle."""
    ids, strains, genes, names, traits = read_dense(
        golden_text("exampledata/Gene_presence_absence.csv.gz"),
        golden_text("exampledata/Tetracycline_resistance.csv.gz"))
    tree = _example_tree()
    index_of = {s: i for i, s in enumerate(strains)}
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        manifest = json.load(f)
    for ti, trait in enumerate(names):
        rows = list(csv.reader(io.StringIO(
            golden_text("csv_pairwise_default/%s.results.csv.gz" % trait))))
        h = rows[0]
        assert h[13:18] == ["Max_Pairwise_comparisons", "Max_supporting_pairs",
                            "Max_opposing_pairs", "Best_pairwise_comp_p", "Worst_pairwise_comp_p"]
        missing = manifest["prune"][trait]
        ptree = orc.prune_for_missing(tree, missing + [None]) if missing else tree
        ops, tips = orc.tree_program(ptree, index_of)
        assert len(rows) > 50
        for d in rows[1:]:
            g = ids.index(d[0])
            st = np.array([(0 if genes[g, i] else 2) + (0 if traits[ti, i] == 1 else 1)
                           for i in tips], dtype=np.uint8)
            tot, pro, anti = orc.tree_dp(ops, st)
            assert (tot, pro, anti) == (int(d[13]), int(d[14]), int(d[15])), d[0]
            best = orc.binom_two_sided_half(pro, tot)
            worst = orc.binom_two_sided_half(tot - anti, tot)
            if pro < anti:
                best, worst = worst, best
            assert abs(best - float(d[16])) <= 1e-12 * float(d[16])
            assert abs(worst - float(d[17])) <= 1e-12 * float(d[17])


def test_synth2_tree_and_pairwise_columns():
    """Second data set: the oracle's UPGMA tree is the reference's Tree.nwk, and the
    pairwise columns of its default-mode CSVs are reproduced (pruned trees for
    the traits with NA / absent isolates)."""
    ids, strains, genes, names, traits = read_dense(golden_text("synth2/gpa.csv.gz"),
                                                    golden_text("synth2/traits.csv.gz"))
    var = (genes.sum(1) > 0) & (genes.sum(1) < len(strains))
    tree = orc.upgma(genes[var].T, strains)
    assert orc.newick(tree) == golden_text("synth2/Tree.nwk.gz").strip()
    index_of = {s: i for i, s in enumerate(strains)}
    checked = 0
    for ti, trait in enumerate(names):
        rows = list(csv.reader(io.StringIO(
            golden_text("synth2/pairwise/%s.results.csv.gz" % trait))))
        missing = [s for i, s in enumerate(strains) if traits[ti, i] == 2]
        ptree = orc.prune_for_missing(tree, missing + [None]) if missing else tree
        ops, tips = orc.tree_program(ptree, index_of)
        for d in rows[1:]:
            g = ids.index(d[0])
            st = np.array([(0 if genes[g, i] else 2) + (0 if traits[ti, i] == 1 else 1)
                           for i in tips], dtype=np.uint8)
            assert orc.tree_dp(ops, st) == (int(d[13]), int(d[14]), int(d[15])), (trait, d[0])
            checked += 1
    assert checked > 150
