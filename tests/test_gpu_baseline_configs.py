"""GPU parity on the real BASELINE.json workloads.

Every test names the config it runs (`synth.make_config`): cfg2 in full, cfg3
at its full permutation count, cfg4 in full (200 000 rare variants x 5000
isolates, the regime of very short and empty minority lists).  The CPU oracle
checks a gene subsample bit for bit at the config's own P; the dense kernel
(an independent implementation of the same counts) checks EVERY gene at the
config's full P as well (round 3; cfg5 at its full shard size lives in
test_gpu_full_size.py).  Reference semantics: scoary/methods.py:804-814 (skip rule -- most
rare variants sit next to it), :1365 (estimator).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

P_TOL = 1e-12


@pytest.fixture(scope="module")
def eng():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    from scoary_amd.engine import AssociationEngine
    e = AssociationEngine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _bits(traits):
    from scoary_amd.engine import pack_bits_rows
    return (pack_bits_rows((traits == 1).astype(np.uint8)),
            pack_bits_rows((traits != 2).astype(np.uint8)))


def _check_subsample(orc, res, genes, tb, mb, N, P, seed, sub):
    counts = res["counts"].cpu().numpy()
    p = res["p"].cpu().numpy()
    r = res["r"].cpu().numpy().view(np.uint32)
    gb = orc.pack_rows(genes[sub])
    want_c = orc.counts_packed(gb, tb, mb).transpose(1, 0, 2)
    assert np.array_equal(counts[:, sub], want_c)
    _, want_p = orc.fisher_many(np.ascontiguousarray(want_c).reshape(-1, 4))
    assert np.max(np.abs(p[:, sub].ravel() - want_p)) < P_TOL
    want_r = orc.permute_r(gb, tb, mb, N, P, seed).T
    assert np.array_equal(r[:, sub], want_r)
    return counts, p, r


def test_cfg2_full_vs_oracle(eng, orc):
    """BASELINE configs[1] (cfg2) in full: 10 000 genes x 500 isolates x 1 trait,
    P = 1000 -- every one of the 1e7 tests against the oracle, list kernel and
    dense kernel."""
    from scoary_amd import synth
    genes, traits, P, seed = synth.make_config("cfg2")
    G, N = genes.shape
    assert (G, N, traits.shape[0], P) == (10_000, 500, 1, 1_000)
    tb, mb = _bits(traits)
    gm = eng.pack_dense(genes)
    trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
    dense = eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=False)
    eng.build_lists(gm)
    res = eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=True)
    assert np.array_equal(res["r"].cpu().numpy(), dense["r"].cpu().numpy())
    _check_subsample(orc, res, genes, tb, mb, N, P, seed, np.arange(G))


def test_cfg3_full_permutations_vs_oracle_subsample(eng, orc):
    """BASELINE configs[2] (cfg3, the headline) with its full P = 10 000 -- the
    20-tiles-per-trait launch bench.py times: every 97th gene (516 genes x 10
    traits x 10 000 permutations = 5.2e7 tests) against the oracle bit for bit,
    r <= P / r == P on skip-rule genes for all 500 000 pairs, and list kernel == dense
    kernel on all 500 000 pairs at the full P."""
    from scoary_amd import synth
    genes, traits, P, seed = synth.make_config("cfg3")
    G, N = genes.shape
    T = traits.shape[0]
    assert (G, N, T, P) == (50_000, 2_000, 10, 10_000)
    tb, mb = _bits(traits)
    gm = eng.pack_dense(genes)
    eng.build_lists(gm)
    trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
    res = eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=True)
    counts, p, r = _check_subsample(orc, res, genes, tb, mb, N, P, seed, np.arange(0, G, 97))
    # every one of the 500 000 (gene, trait) pairs at the FULL P against the dense
    # AND+popcount kernel, an independent implementation of the same counts (16 ms of GPU)
    dense = eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=False)
    assert np.array_equal(dense["r"].cpu().numpy().view(np.uint32), r)
    assert np.array_equal(dense["counts"].cpu().numpy(), counts)
    assert r.max() <= P
    skipped = (counts[:, :, 0] + counts[:, :, 2] == 0) | (counts[:, :, 1] + counts[:, :, 3] == 0)
    assert skipped.sum() > 0 and np.all(r[skipped] == P)
    # margins are permutation invariant: E[r]/P is the two-sided p itself (up to the
    # discreteness of the test), so the empirical p must track the Fisher p
    emp = (r + 1.0) / (P + 1.0)
    big = ~skipped & (p > 0.05)
    assert np.max(np.abs(emp[big] - p[big])) < 0.05


def test_cfg4_rare_variants_full(eng, orc):
    """BASELINE configs[3] (cfg4): 200 000 rare variants (minor-allele frequency
    ~ Beta(0.3, 3)) x 5000 isolates x 1 trait, P = 10 000, whole matrix on one
    GPU.  Most lists are very short or empty (wave groups with no list steps at
    all).  The oracle checks every 400th gene at the full P; the dense kernel
    checks ALL genes at the full P too."""
    from scoary_amd import synth
    genes, traits, P, seed = synth.make_config("cfg4")
    G, N = genes.shape
    assert (G, N, traits.shape[0], P) == (200_000, 5_000, 1, 10_000)
    tb, mb = _bits(traits)
    gm = eng.pack_dense(genes)
    trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
    eng.build_lists(gm)
    lens = genes.sum(1)
    assert (lens == 0).sum() > 1000 and np.median(lens) < 0.1 * N      # the regime this test is for
    res = eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=True)
    counts, p, r = _check_subsample(orc, res, genes, tb, mb, N, P, seed, np.arange(0, G, 400))
    assert r.max() <= P
    skipped = (counts[:, :, 0] + counts[:, :, 2] == 0) | (counts[:, :, 1] + counts[:, :, 3] == 0)
    assert skipped.sum() > 1000 and np.all(r[skipped] == P)
    # list kernel == dense kernel on EVERY gene at the config's full P (2e9 tests through
    # both), and once more at a P that is not a multiple of the tile width
    dense = eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=False)
    assert np.array_equal(dense["r"].cpu().numpy().view(np.uint32), r)
    assert np.array_equal(dense["counts"].cpu().numpy(), counts)
    P2 = 300
    small = eng.associate(gm, trv, mkv, permutations=P2, seed=seed, use_lists=True)
    dense = eng.associate(gm, trv, mkv, permutations=P2, seed=seed, use_lists=False)
    assert np.array_equal(small["r"].cpu().numpy(), dense["r"].cpu().numpy())
