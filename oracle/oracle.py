"""CPU oracle for Scoary's association hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product package ``scoary_amd`` never does.

It wraps ``oracle.c`` (ctypes) and adds the two pieces of the path that are
host-side floating point in the reference and are restated here as plain
Python loops following the reference statement by statement:

  * ``bonferroni_bh``    scoary/methods.py:903-925   (step-up BH with ties)
  * ``setup_results``    scoary/methods.py:771-925   (the gene loop: skip rule,
                         sens/spec, memoised Fisher, B/BH) on dense 0/1 arrays

Parity status: PINNED by tests/test_oracle_golden.py against tests/golden/
(vectors captured from the real reference + SciPy 1.15.3).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")

GAMMA = 1.0 + 1e-14     # spec S3 tie rule (SciPy's), decided exactly in oracle.c (hg_leq)


def build(force=False):
    src = os.path.join(_HERE, "oracle.c")
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        L = ctypes.CDLL(_LIB)
        i64, u64, u32, vp = ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p
        L.orc_pack_rows.argtypes = [vp, i64, i64, vp]
        L.orc_counts_dense.argtypes = [vp, vp, i64, i64, vp]
        L.orc_counts_packed.argtypes = [vp, vp, vp, i64, i64, i64, vp]
        L.orc_fisher.argtypes = [i64, i64, i64, i64, vp, vp]
        L.orc_fisher_many.argtypes = [vp, i64, vp, vp]
        L.orc_fisher_scipy_many.argtypes = [vp, i64, vp]
        L.orc_fisher_scipy_many.restype = None
        L.orc_philox4x32_10.argtypes = [vp, vp, vp]
        L.orc_perm_labels.argtypes = [u64, u32, u32, vp, i64, i64, vp]
        L.orc_perm_block.argtypes = [u64, u32, u32, vp, i64, i64, vp]
        L.orc_perm_plan.argtypes = [i64, i64, vp, vp, vp]
        L.orc_permute_r.argtypes = [vp, vp, vp, i64, i64, i64, i64, u64, i64, vp]
        L.orc_num_threads.restype = ctypes.c_int
        L.orc_set_num_threads.argtypes = [ctypes.c_int]
        for f in (L.orc_pack_rows, L.orc_counts_dense, L.orc_counts_packed,
                  L.orc_fisher, L.orc_fisher_many, L.orc_philox4x32_10,
                  L.orc_perm_labels, L.orc_perm_block, L.orc_perm_plan, L.orc_permute_r,
                  L.orc_set_num_threads):
            f.restype = None
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def words(n):
    return (int(n) + 63) // 64


def pack_rows(dense):
    """dense: (G, N) 0/1 uint8 -> (G, W) uint64, bit i of word w = col 64w+i."""
    dense = np.ascontiguousarray(dense, dtype=np.uint8)
    G, N = dense.shape
    out = np.zeros((G, words(N)), dtype=np.uint64)
    lib().orc_pack_rows(_p(dense), G, N, _p(out))
    return out


def counts_dense(genes, trait):
    """genes (G, N) 0/1; trait (N,) with 0/1/2(missing) -> (G, 4) int32
    tpgp, tpgn, tngp, tngn -- the isolate-by-isolate loop of Perform_statistics."""
    genes = np.ascontiguousarray(genes, dtype=np.uint8)
    trait = np.ascontiguousarray(trait, dtype=np.uint8)
    G, N = genes.shape
    out = np.zeros((G, 4), dtype=np.int32)
    lib().orc_counts_dense(_p(genes), _p(trait), G, N, _p(out))
    return out


def counts_packed(gbits, tbits, mbits):
    gbits = np.ascontiguousarray(gbits, dtype=np.uint64)
    tbits = np.ascontiguousarray(tbits, dtype=np.uint64)
    mbits = np.ascontiguousarray(mbits, dtype=np.uint64)
    G, W = gbits.shape
    T = tbits.shape[0]
    out = np.zeros((G, T, 4), dtype=np.int32)
    lib().orc_counts_packed(_p(gbits), _p(tbits), _p(mbits), G, T, W, _p(out))
    return out


def fisher(a, b, c, d):
    p = ctypes.c_double()
    o = ctypes.c_double()
    lib().orc_fisher(int(a), int(b), int(c), int(d), ctypes.byref(p), ctypes.byref(o))
    return o.value, p.value


def fisher_scipy_many(counts):
    """scipy.stats.fisher_exact's own double for every [tpgp, tpgn, tngp, tngn] row (oracle.c orc_fisher_scipy: Boost's
    factorial-table pmf up to 170 isolates, its prime-factorised pmf up to 104 723; NaN beyond)."""
    counts = np.ascontiguousarray(counts, dtype=np.int32).reshape(-1, 4)
    p = np.empty(counts.shape[0])
    lib().orc_fisher_scipy_many(_p(counts), counts.shape[0], _p(p))
    return p


def fisher_many(counts):
    counts = np.ascontiguousarray(counts, dtype=np.int32).reshape(-1, 4)
    M = counts.shape[0]
    p = np.empty(M)
    o = np.empty(M)
    lib().orc_fisher_many(_p(counts), M, _p(p), _p(o))
    return o, p


def philox4x32_10(ctr, key):
    ctr = np.asarray(ctr, dtype=np.uint32)
    key = np.asarray(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib().orc_philox4x32_10(_p(ctr), _p(key), _p(out))
    return out


def perm_labels(seed, t, pi, mask_bits, npos, N):
    mask_bits = np.ascontiguousarray(mask_bits, dtype=np.uint64)
    out = np.zeros(words(N), dtype=np.uint64)
    lib().orc_perm_labels(int(seed), int(t), int(pi), _p(mask_bits), int(npos), int(N), _p(out))
    return out


def perm_block(seed, t, block, mask_bits, npos, N):
    """The 32 permutations 32*block .. 32*block+31 of trait t (spec S4): (32, W64) uint64."""
    mask_bits = np.ascontiguousarray(mask_bits, dtype=np.uint64)
    out = np.zeros((32, words(N)), dtype=np.uint64)
    lib().orc_perm_block(int(seed), int(t), int(block), _p(mask_bits), int(npos), int(N), _p(out))
    return out


def perm_plan(npos, nval):
    """(marks to place m, complement taken?, 8-bit round-0 probability q) of spec S4."""
    m = ctypes.c_int64()
    flip = ctypes.c_int()
    q = ctypes.c_uint32()
    lib().orc_perm_plan(int(npos), int(nval), ctypes.byref(m), ctypes.byref(flip), ctypes.byref(q))
    return int(m.value), bool(flip.value), int(q.value)


def permute_r(gbits, tbits, mbits, N, P, seed, perm_base=0):
    gbits = np.ascontiguousarray(gbits, dtype=np.uint64)
    tbits = np.ascontiguousarray(tbits, dtype=np.uint64)
    mbits = np.ascontiguousarray(mbits, dtype=np.uint64)
    G = gbits.shape[0]
    T = tbits.shape[0]
    r = np.zeros((G, T), dtype=np.uint32)
    lib().orc_permute_r(_p(gbits), _p(tbits), _p(mbits), G, T, int(N), int(P),
                        int(seed), int(perm_base), _p(r))
    return r


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


# ---------------------------------------------------------------------------
# Host-side floating point of the path, restated as the reference's loops.
# ---------------------------------------------------------------------------
def bonferroni_bh(pvals, number_of_tests):
    """methods.py:903-925.  pvals in gene (file) order -> (B_p, BH_p) lists.

    Stable ascending sort; tie[i] = exact fp equality with the next entry; the
    least significant entry keeps its p; walking towards the most significant,
    bh = last if tie else min(last, p*ntests/rank); both capped at 1.0."""
    n = len(pvals)
    order = sorted(range(n), key=lambda i: pvals[i])
    sp = [pvals[i] for i in order]
    tie = [sp[i - 1] == sp[i] for i in range(1, n)]
    bh = [0.0] * n
    last = sp[n - 1]
    bh[order[n - 1]] = last
    for ind in range(n - 2, -1, -1):
        if tie[ind]:
            val = last
        else:
            val = min(last, sp[ind] * number_of_tests / (ind + 1.0))
        bh[order[ind]] = val
        last = val
    B = [min(p * number_of_tests, 1.0) for p in pvals]
    BH = [min(x, 1.0) for x in bh]
    return B, BH


def setup_results(genes, trait):
    """The gene loop of Setup_results (methods.py:791-925) for ONE trait, no
    collapse.  genes (G, N) 0/1; trait (N,) 0/1/2.  Returns dict of arrays over
    the *testable* genes in file order + their indices."""
    cnt = counts_dense(genes, trait)
    G = cnt.shape[0]
    number_of_tests = G
    memo = {}
    idx, rows = [], []
    for g in range(G):
        tpgp, tpgn, tngp, tngn = (int(x) for x in cnt[g])
        num_pos, num_neg = tpgp + tpgn, tngp + tngn
        if tpgp + tngp == 0 or tpgn + tngn == 0:
            number_of_tests -= 1
            continue
        key = (tpgp, tpgn, tngp, tngn)
        if key not in memo:
            memo[key] = fisher(*key)
        odds, p = memo[key]
        sens = (float(tpgp) / num_pos * 100) if num_pos > 0 else 0.0
        spes = (float(tngn) / num_neg * 100) if num_neg > 0 else 0.0
        idx.append(g)
        rows.append((tpgp, tpgn, tngp, tngn, sens, spes, odds, p))
    pv = [r[7] for r in rows]
    B, BH = bonferroni_bh(pv, number_of_tests) if rows else ([], [])
    return {
        "index": np.array(idx, dtype=np.int64),
        "counts": np.array([r[:4] for r in rows], dtype=np.int32).reshape(-1, 4),
        "sens": np.array([r[4] for r in rows]),
        "spes": np.array([r[5] for r in rows]),
        "OR": np.array([r[6] for r in rows]),
        "p_v": np.array(pv),
        "B_p": np.array(B),
        "BH_p": np.array(BH),
        "number_of_tests": number_of_tests,
    }


# ---------------------------------------------------------------------------
# Population-structure stage: tree restatements
# ---------------------------------------------------------------------------
_lib_tree_ready = False


def _tree_lib():
    global _lib_tree_ready
    L = lib()
    if not _lib_tree_ready:
        i64, u64, u32, vp = ctypes.c_int64, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_void_p
        L.orc_tree_dp.argtypes = [vp, i64, vp, vp]
        L.orc_tree_dp.restype = ctypes.c_int
        L.orc_tree_permute.argtypes = [vp, i64, vp, i64, vp, vp, vp, i64, u32, i64, u64, vp, vp]
        L.orc_tree_permute.restype = ctypes.c_int
        _lib_tree_ready = True
    return L


def prune_for_missing(tree, prune):
    """PruneForMissing (methods.py:709-739): drop tips listed in ``prune`` (the
    list carries a trailing None so a fully pruned subtree, returned as None,
    is dropped by its parent too); unary nodes collapse into their child."""
    left, right = tree[0], tree[1]
    if isinstance(left, list):
        left = prune_for_missing(left, prune)
    if isinstance(right, list):
        right = prune_for_missing(right, prune)
    if left in prune and right in prune:
        return None
    if left in prune:
        return right
    if right in prune:
        return left
    return [left, right]


def tree_program(tree, index_of):
    """Nested-list tree -> (ops int32, tips int32): post-order stack program,
    ops[k] >= 0 = push tip number ops[k] (tips[ops[k]] = isolate index),
    -1 = merge the two top entries."""
    ops, tips = [], []

    def walk(node):
        if isinstance(node, (list, tuple)):
            walk(node[0])
            walk(node[1])
            ops.append(-1)
        else:
            ops.append(len(tips))
            tips.append(index_of[node])
    import sys as _sys
    old = _sys.getrecursionlimit()
    _sys.setrecursionlimit(max(old, 100000))
    try:
        walk(tree)
    finally:
        _sys.setrecursionlimit(old)
    return np.array(ops, dtype=np.int32), np.array(tips, dtype=np.int32)


TIP_STATE = {"AB": 0, "Ab": 1, "aB": 2, "ab": 3}


def tree_dp(ops, tipstate):
    """(max_contrasting_pairs, propairs, antipairs) -- classes.py:199-592."""
    ops = np.ascontiguousarray(ops, dtype=np.int32)
    tipstate = np.ascontiguousarray(tipstate, dtype=np.uint8)
    out = np.zeros(3, dtype=np.int32)
    rc = _tree_lib().orc_tree_dp(_p(ops), len(ops), _p(tipstate), _p(out))
    if rc != 0:
        raise ValueError("malformed tree program")
    return tuple(int(x) for x in out)


def tree_permute(ops, tips, gbits, tbits, mbits, N, t, P, seed):
    """Observed (Total, Pro, Anti) and the per-permutation exceedance flags of
    Permute (methods.py:1314-1369) for one gene, labels by spec S4."""
    ops = np.ascontiguousarray(ops, dtype=np.int32)
    tips = np.ascontiguousarray(tips, dtype=np.int32)
    obs = np.zeros(3, dtype=np.int32)
    ex = np.zeros(P, dtype=np.uint8)
    rc = _tree_lib().orc_tree_permute(
        _p(ops), len(ops), _p(tips), len(tips),
        _p(np.ascontiguousarray(gbits, dtype=np.uint64)),
        _p(np.ascontiguousarray(tbits, dtype=np.uint64)),
        _p(np.ascontiguousarray(mbits, dtype=np.uint64)), int(N), int(t), int(P), int(seed),
        _p(obs), _p(ex))
    if rc != 0:
        raise ValueError("malformed tree program")
    return tuple(int(x) for x in obs), ex


def empirical_p_with_abort(exceed):
    """The sequential estimator of methods.py:1348-1365: r counts exceedances;
    after each permutation i >= 30, if 1 - binom.cdf(r, i, 0.1) < 0.05 return
    (r+1)/(i+2); otherwise (r+1)/(P+1)."""
    import scipy.stats as ss
    r = 0
    P = len(exceed)
    for i in range(P):
        r += int(exceed[i])
        if i >= 30 and (1 - ss.binom.cdf(r, i, 0.1)) < 0.05:
            return (r + 1.0) / (i + 2.0)
    return (r + 1.0) / (P + 1.0)


def binom_two_sided_half(x, n):
    """scipy binom_test(x, n, 0.5) (methods.py:1267-1275), exact: the p = 0.5
    binomial is symmetric, so the two-sided p is 2*P(X <= min(x, n-x)), 1 when
    x == n/2, capped at 1."""
    from fractions import Fraction
    from math import comb
    x, n = int(x), int(n)
    if 2 * x == n:
        return 1.0
    k = min(x, n - x)
    tail = sum(comb(n, j) for j in range(k + 1))
    return float(min(Fraction(1), Fraction(2 * tail, 2 ** n)))


def _morton(i, j, levels):
    key = 0
    for b in range(levels - 1, -1, -1):
        key = (key << 2) | (((i >> b) & 1) << 1) | ((j >> b) & 1)
    return key


def upgma(zero_ones_strain_major, names):
    """UPGMA tree of methods.py:619-707 + classes.py:68-196, restated without
    the quad tree: Hamming distances (fraction of differing variable genes,
    d(i,i) = 1), repeated merge of the closest pair, size-weighted average
    (d_ik*size_i + d_jk*size_j)/new_size, dead clusters at distance 1 for the
    averaging and maxsize in the matrix.  The reference finds the minimum by
    descending a quad tree of (value, i, j) minima; among equal values that is
    the cell with the smallest bit-interleaved (i-major Morton) index."""
    import sys as _sys
    X = np.asarray(zero_ones_strain_major, dtype=np.uint8)
    n = X.shape[0]
    big = float(_sys.maxsize)
    D = [[big] * n for _ in range(n)]
    ngenes = X.shape[1]
    for i in range(n):
        D[i][i] = 1
        for j in range(i + 1, n):
            D[i][j] = D[j][i] = float(int((X[i] != X[j]).sum())) / ngenes
    levels = max(1, int(np.ceil(np.log2(max(n, 2)))) + 1)
    cluster = list(names)
    size = [1] * n
    alive = n
    new_cluster = None
    while alive > 1:
        best = None
        for i in range(n):
            for j in range(n):
                v = D[i][j]
                key = (v, _morton(i, j, levels))
                if best is None or key < best[0]:
                    best = (key, i, j)
        _, i, j = best
        new_cluster = [cluster[i], cluster[j]]
        new_size = size[i] + size[j]
        nd = []
        for k in range(n):
            if cluster[k] is None:
                nd.append(1)
            else:
                nd.append((D[i][k] * size[i] + D[j][k] * size[j]) / new_size)
        nd[i] = big
        for k in range(n):
            D[i][k] = D[k][i] = nd[k]
        for k in range(n):
            D[j][k] = D[k][j] = big
        cluster[i], cluster[j] = new_cluster, None
        size[i], size[j] = new_size, 0
        alive -= 1
    return new_cluster


def newick(tree):
    """StoreUPGMAtreeToFile's text form (methods.py:741-751)."""
    return str(tree).replace("[", "(").replace("]", ")") + ";"
