"""GPU parity at the one BASELINE size that had only ever been benchmarked (VERDICT round 2,
item 1): a FULL per-GPU shard of configs[4] (cfg5, "HBM-capacity stress"): 125 000 genes x
10 000 isolates x 50 traits at the config's P = 100 000, through the code path bench.py times
for it -- the list-driven kernel with the real multi-batch plan of AssociationEngine.list_batch
(three batches of <= 43 520 permutations, > 2^31 16-bit per-tile counts, `accumulate` across the
batches, global permutation indices).

Checkers: (1) the CPU oracle on a stratified gene subsample (500 genes) x all 50 traits at the full P
(2.5e9 tests); (2) the dense AND+popcount kernels -- an independent implementation of the same
counts -- on EVERY (gene, trait) pair at the full P as well (6.25e11 tests, a few seconds of
GPU).  Reference semantics: scoary/methods.py:804-814 (skip rule), :1348-1365 (estimator).
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


def test_cfg5_full_shard_full_permutations(eng):
    import torch
    from oracle import oracle as orc
    from scoary_amd import synth
    from scoary_amd.engine import pack_bits_rows
    genes, traits, P, seed = synth.make_config("cfg5", G=125_000)
    G, N = genes.shape
    T = traits.shape[0]
    assert (G, N, T, P) == (125_000, 10_000, 50, 100_000)
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    gm = eng.pack_dense(genes)
    trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
    eng.build_lists(gm)

    # the plan bench.py --config cfg5 runs with: >= 3 batches, scratch under 4 GB -- and past
    # 2^31 bytes of 16-bit per-tile counts: a 32-bit byte offset (or a signed 32-bit element
    # index times two) would wrap inside this buffer
    ws = eng.workspace(gm, T, P, use_lists=True)
    tile_perms = 32 * eng.list_params(N)[0]
    assert ws.batch % 512 == 0 and -(-P // ws.batch) >= 3
    assert 2 * T * (G + 64) * (ws.batch // tile_perms) <= 4 << 30          # the 16-bit counts
    assert ws.scratch.numel() * 4 <= (4 << 30) + 8 * T * G                   # + slot-order regions
    assert 2 * T * (ws.batch // tile_perms) * G > 2 ** 31
    res = eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=True, workspace=ws)
    torch.cuda.synchronize()
    counts = res["counts"].cpu().numpy()
    p = res["p"].cpu().numpy()
    r = res["r"].cpu().numpy().view(np.uint32).copy()
    del ws

    # size-independent properties on all 6.25e6 pairs
    nval = (traits != 2).sum(1)
    assert np.array_equal(counts.sum(2), np.broadcast_to(nval[:, None], (T, G)))
    assert r.max() <= P
    skipped = (counts[:, :, 0] + counts[:, :, 2] == 0) | (counts[:, :, 1] + counts[:, :, 3] == 0)
    assert skipped.sum() > 1000 and np.all(r[skipped] == P)
    emp = (r + 1.0) / (P + 1.0)
    big = ~skipped & (p > 0.05)
    assert np.max(np.abs(emp[big] - p[big])) < 0.02          # E[r]/P is the two-sided p itself

    # (1) oracle, full P: genes stratified over the list-length order (the order the kernel
    # walks them in), both ends of it, and skip-rule genes
    ones = genes.sum(1, dtype=np.int64)
    by_len = np.argsort(-np.minimum(ones, N - ones), kind="stable")
    core = np.flatnonzero((ones == 0) | (ones == N))[:4]
    # (round 5: 500 genes instead of 124 -- the oracle draws its labels 32 permutations at a time now,
    # which left room for four times the genes in the same minute)
    sub = np.unique(np.concatenate([by_len[np.linspace(0, G - 1, 488).astype(np.int64)],
                                    by_len[:6], by_len[-6:], core]))
    assert 450 <= len(sub) <= 504
    gb = orc.pack_rows(genes[sub])
    want_c = orc.counts_packed(gb, tb, mb).transpose(1, 0, 2)
    assert np.array_equal(counts[:, sub], want_c)
    _, want_p = orc.fisher_many(np.ascontiguousarray(want_c).reshape(-1, 4))
    assert np.max(np.abs(p[:, sub].ravel() - want_p)) < P_TOL
    want_r = orc.permute_r(gb, tb, mb, N, P, seed).T
    assert np.array_equal(r[:, sub], want_r)

    # (2) the dense kernels on every pair, full P (chunked AND+popcount, label rows in HBM)
    dense = eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=False)
    assert np.array_equal(dense["r"].cpu().numpy().view(np.uint32), r)
    assert np.array_equal(dense["counts"].cpu().numpy(), counts)
