"""Population-structure stage, host side.

  upgma()              scoary/methods.py:619-707 + scoary/classes.py:68-196
                       (Hamming counts on the GPU, the O(N^2) merge loop on the
                       host with a vectorised min-quad-tree)
  prune_missing()      scoary/methods.py:709-739   PruneForMissing
  TreeProgram          the stack program scoary_tree_pairs / scoary_tree_permute
                       evaluate (replaces the recursive PhyloTree construction,
                       scoary/methods.py:1386-1402, scoary/classes.py:210-249)
  binom_two_sided()    ss.binom_test(x, n, 0.5), scoary/methods.py:1267-1275
  newick_text() / read_newick()   scoary/methods.py:741-752, scoary/nwkhandler.py

Trees are the reference's nested two-element lists of strain names.  All
traversals are iterative: UPGMA trees can be caterpillars thousands deep.
"""
import sys
from fractions import Fraction

import numpy as np

BIG = float(sys.maxsize)


# ---------------------------------------------------------------------------
# UPGMA
# ---------------------------------------------------------------------------
class _MinQuadTree:
    """Levels of 2x2 block minima over a square matrix (the reference's
    QuadTree).  Level 0 is the matrix padded to even size with BIG; level l+1
    holds the minima of level l's 2x2 blocks, padded to even size again."""

    def __init__(self, D):
        n = D.shape[0]
        m = n + n % 2
        cur = np.full((m, m), BIG, dtype=np.float64)
        cur[:n, :n] = D
        self.levels = [cur]
        while cur.shape[0] > 2:
            h = cur.shape[0] // 2
            nxt = cur.reshape(h, 2, h, 2).min(axis=(1, 3))
            if h % 2:
                pad = np.full((h + 1, h + 1), BIG, dtype=np.float64)
                pad[:h, :h] = nxt
                nxt = pad
            self.levels.append(nxt)
            cur = nxt

    def _set(self, idx, vec, row):
        v = np.asarray(vec, dtype=np.float64)
        for lv, cur in enumerate(self.levels):
            m = cur.shape[0]
            if v.shape[0] < m:
                v = np.concatenate([v, np.full(m - v.shape[0], BIG)])
            if row:
                cur[idx, :] = v
                pair = np.minimum(cur[idx & ~1, :], cur[idx | 1, :])
            else:
                cur[:, idx] = v
                pair = np.minimum(cur[:, idx & ~1], cur[:, idx | 1])
            v = pair.reshape(-1, 2).min(axis=1)
            idx //= 2

    def set_row(self, i, vec):
        self._set(i, vec, True)

    def set_col(self, j, vec):
        self._set(j, vec, False)

    def argmin(self):
        """Descend from the top: in each 2x2 block take the smallest
        (value, i, j) -- the reference's tie-break (classes.py:172-196)."""
        i = j = 0
        for cur in reversed(self.levels):
            i, j = 2 * i, 2 * j
            best = None
            for di in (0, 1):
                for dj in (0, 1):
                    key = (cur[i + di, j + dj], i + di, j + dj)
                    if best is None or key < best:
                        best = key
            i, j = best[1], best[2]
        return i, j


def upgma_from_counts(counts, ncols, names, native=None):
    """counts: (n, n) integer Hamming counts over ``ncols`` variable genes.
    The merge loop runs in the host library (scoary_upgma_merges, ~100x the numpy
    loop below) when it is built; ``native=False`` forces the numpy loop, which the
    tests keep as the cross-check."""
    n = len(names)
    D = counts.astype(np.float64) / float(ncols)      # pdist 'hamming' fractions
    np.fill_diagonal(D, 1.0)                          # methods.py:636-638
    if native is None:
        from . import io_native
        native = io_native.available()
    if native and n > 1:
        from . import io_native
        try:
            merges = io_native.upgma_merges(D).tolist()
        except RuntimeError:                          # degenerate input (a diagonal entry is the
            merges = None                             # minimum): the numpy loop mirrors the reference
        if merges is not None:
            cluster = list(names)
            for i, j in merges:
                cluster[i], cluster[j] = [cluster[i], cluster[j]], None
            return cluster[i]
    qt = _MinQuadTree(D)
    d0 = qt.levels[0]
    cluster = list(names)
    alive = np.ones(n, dtype=bool)
    size = np.ones(n, dtype=np.float64)
    new_cluster = cluster[0] if n == 1 else None
    for _ in range(n - 1):
        i, j = qt.argmin()
        new_cluster = [cluster[i], cluster[j]]
        new_size = size[i] + size[j]
        nd = (d0[i, :n] * size[i] + d0[j, :n] * size[j]) / new_size
        nd[~alive] = 1.0
        nd[i] = BIG
        qt.set_row(i, nd)
        qt.set_col(i, nd)
        dead = np.full(n, BIG)
        qt.set_row(j, dead)
        qt.set_col(j, dead)
        cluster[i], cluster[j] = new_cluster, None
        alive[j] = False
        size[i], size[j] = new_size, 0.0
    return new_cluster


def upgma(engine, dense_genes_by_strain, names):
    """dense (G, N) 0/1 presence matrix -> UPGMA tree over the N strains, from
    Hamming distances over the variable genes (methods.py:496-502, 619-707).
    Hamming counts and the merge loop run on the GPU (scoary_hamming,
    scoary_upgma); the host loop takes over in the degenerate case the device
    loop hands back."""
    dense = np.asarray(dense_genes_by_strain, dtype=np.uint8)
    tot = dense.sum(axis=1)
    var = dense[(tot > 0) & (tot < dense.shape[1])]
    if var.shape[0] == 0:
        raise ValueError("no variable genes: cannot build a tree")
    rows = np.ascontiguousarray(var.T)
    if len(names) > 1:
        merges = engine.upgma_merges(rows)            # device loop (scoary_upgma)
        if merges is not None:
            cluster = list(names)
            for i, j in merges.tolist():
                cluster[i], cluster[j] = [cluster[i], cluster[j]], None
            return cluster[i]
    return upgma_from_counts(engine.hamming(rows), var.shape[0], names)


# ---------------------------------------------------------------------------
# Tree utilities (iterative)
# ---------------------------------------------------------------------------
def _flatten(tree):
    """Nested lists -> (left[], right[], name[]) arrays; node 0 is the root.
    Tips have left = right = -1."""
    left, right, name = [], [], []
    stack = [(tree, -1, 0)]
    while stack:
        node, parent, side = stack.pop()
        me = len(left)
        left.append(-1)
        right.append(-1)
        name.append(None)
        if parent >= 0:
            (left if side == 0 else right)[parent] = me
        if isinstance(node, (list, tuple)):
            if len(node) != 2:
                raise ValueError("tree nodes must be binary")
            stack.append((node[1], me, 1))
            stack.append((node[0], me, 0))
        else:
            name[me] = node
    return left, right, name


def prune_missing(tree, prune):
    """PruneForMissing (methods.py:709-739): remove the tips in ``prune``;
    a node left with one child is replaced by that child, with none it
    disappears.  Returns None if nothing remains."""
    drop = set(x for x in prune if x is not None)
    if not drop:
        return tree
    left, right, name = _flatten(tree)
    n = len(left)
    result = [None] * n
    for v in range(n - 1, -1, -1):          # children have larger indices than parents
        if left[v] < 0:
            result[v] = None if name[v] in drop else name[v]
        else:
            a, b = result[left[v]], result[right[v]]
            if a is None and b is None:
                result[v] = None
            elif a is None:
                result[v] = b
            elif b is None:
                result[v] = a
            else:
                result[v] = [a, b]
    return result[0]


def tips_of(tree):
    return [x for x in _flatten(tree)[2] if x is not None]


class TreeProgram:
    """Stack program of a binary tree for scoary_tree_pairs / _permute:
    op >= 0 push tip op; -1 merge the two top entries; <= -2 merge the top entry
    with tip (-2 - op).  Tips are numbered in program order; ``tips[k]`` is the
    isolate index of tip k.  The larger subtree is always evaluated first, so
    the stack never gets deeper than log2(#tips) + 1."""

    def __init__(self, tree, index_of):
        left, right, name = _flatten(tree)
        n = len(left)
        if n < 3:
            raise ValueError("a tree needs at least two tips")
        unknown = [name[v] for v in range(n) if left[v] < 0 and name[v] not in index_of]
        if unknown:
            # (a degenerate UPGMA tree -- e.g. ONE variable gene -- carries an unnamed tip; the reference ends in
            # binom_test(0, 0) there: "n must be an integer not less than 1")
            raise ValueError("the tree has a tip that is no isolate of the gene table: %r" % (unknown[0],))
        size = [1] * n
        for v in range(n - 1, -1, -1):
            if left[v] >= 0:
                size[v] = size[left[v]] + size[right[v]]
        ops, tips = [], []
        depth = maxdepth = 0          # entries below the top of the stack
        started = False
        work = [("visit", 0)]
        while work:
            kind, v = work.pop()
            if kind == "merge":
                ops.append(-1)
                depth -= 1
            elif kind == "mergetip":
                ops.append(-2 - len(tips))
                tips.append(index_of[name[v]])
            elif left[v] < 0:
                ops.append(len(tips))
                tips.append(index_of[name[v]])
                if started:
                    depth += 1
                    maxdepth = max(maxdepth, depth)
                started = True
            else:
                a, b = left[v], right[v]
                if size[a] < size[b]:
                    a, b = b, a                         # heavy child first
                if left[b] < 0:
                    work.append(("mergetip", b))
                else:
                    work.append(("merge", v))
                    work.append(("visit", b))
                work.append(("visit", a))
        self.ops = np.array(ops, dtype=np.int32)
        self.tips = np.array(tips, dtype=np.int32)
        self.depth = max(1, maxdepth)
        self.ntips = len(tips)


# ---------------------------------------------------------------------------
# Newick in / out
# ---------------------------------------------------------------------------
def newick_text(tree):
    """str(nested lists) with [] -> () plus ';' (methods.py:741-751), written
    iteratively so deep trees do not hit the recursion limit."""
    out = []
    work = [(False, tree)]
    while work:
        literal, x = work.pop()
        if literal:
            out.append(x)
        elif isinstance(x, (list, tuple)):
            work.append((True, ")"))
            work.append((False, x[1]))
            work.append((True, ", "))
            work.append((False, x[0]))
            work.append((True, "("))
        else:
            out.append(repr(x))
    return "".join(out) + ";"


def read_newick(path):
    """Read a Newick file into nested two-element lists of tip names + the
    list of tip names (scoary/nwkhandler.py:10-40).  Branch lengths, support
    values and internal names are ignored; quotes around names are stripped; a
    node with more than two children is resolved in ete3's order
    (_resolve_polytomy)."""
    with open(path) as f:
        text = f.read().strip()
    text = text[:text.index(";")] if ";" in text else text
    pos = 0
    stack = [[]]
    members = []
    n = len(text)
    while pos < n:
        ch = text[pos]
        if ch == "(":
            stack.append([])
            pos += 1
        elif ch == ")":
            kids = stack.pop()
            pos += 1
            # skip internal label / support / branch length
            while pos < n and text[pos] not in ",()":
                pos += 1
            if not kids:
                raise SystemExit("Corrupted or non-existing custom tree file? empty group")
            stack[-1].append(_resolve_polytomy(kids))
        elif ch == ",":
            pos += 1
        elif ch.isspace():
            pos += 1
        else:
            if ch in "'\"":
                end = text.index(ch, pos + 1)
                label = text[pos + 1:end]
                pos = end + 1
                while pos < n and text[pos] not in ",()":
                    pos += 1
            else:
                end = pos
                while end < n and text[end] not in ",():":
                    end += 1
                label = text[pos:end].strip().lstrip("'\"").rstrip("'\"")
                pos = end
                while pos < n and text[pos] not in ",()":
                    pos += 1
            stack[-1].append(label)
            members.append(label)
    return _resolve_polytomy(stack[0]), members


def _resolve_polytomy(kids):
    """Children [k0, k1, ..., kn] of one node -> nested pairs the way the reference's
    ``myTree.resolve_polytomy(recursive=True)`` (ete3, scoary/nwkhandler.py:19) arranges
    them: the first child stays at the top, every further child becomes the sibling of a new
    first child, the last two children are paired --
        [a, b, c]    -> [[b, c], a]
        [a, b, c, d] -> [[[c, d], b], a]
    (ete3 is not installed here: restated from ete3's TreeNode.resolve_polytomy, which adds
    n-2 nested first children and then hands the original children out top-down.)"""
    if len(kids) == 1:
        return kids[0]
    node = [kids[-2], kids[-1]]
    for k in reversed(kids[:-2]):
        node = [node, k]
    return node


# ---------------------------------------------------------------------------
# Binomial test and the sequential permutation estimator
# ---------------------------------------------------------------------------
def _boost_binom_tail_half(k, n, upper):
    """SciPy's binom.cdf(k, n, 0.5) (upper False) / binom.sf(k, n, 0.5) (True) as SciPy >= 1.7 computes them: Boost.Math's
    cdf(binomial) = ibetac(k + 1, n - k, p), its complement = ibeta(k + 1, n - k, p), restated operation by operation
    for integer arguments at x = y = 1/2 (boost/math/special_functions/beta.hpp, ibeta_imp): the b == 1 special case, the
    swap that puts the smaller argument second, and -- while that argument is below 40 -- the finite sum
    binomial_ccdf(n, k, x, y): pow(x, n), then term *= ((i + 1) * y) / ((n - i) * x) from i = n - 1 down to k + 1.  Plain
    fp64, the roundings of that sum are the "noise" of SciPy's value (0.30175781249999994 where the tail is 9888 / 2^15).
    Returns None where Boost takes another branch (second argument >= 40: a continued fraction; n > 1021: pow
    underflows).  Checked equal to scipy.stats.binom.cdf / .sf bit for bit on every (k, n) up to n = 300 it covers
    (tests/test_host_logic.py)."""
    if k < 0:
        return 1.0 if upper else 0.0
    if k >= n:
        return 0.0 if upper else 1.0
    a, b = k + 1, n - k
    invert = not upper
    if a == 1:
        a, b = b, a
        invert = not invert
    if b == 1:
        if a == 1:
            return 0.5
        # y < 0.5 is false at y = 1/2: invert ? -powm1(x, a) : pow(x, a); powm1 falls through to pow(x, a) - 1 here
        v = -(0.5 ** a - 1.0) if invert else 0.5 ** a
        return min(max(v, 0.0), 1.0)
    lam = (a - (a + b) * 0.5) if a < b else ((a + b) * 0.5 - b)
    if lam < 0:
        a, b = b, a
        invert = not invert
    if b >= 40 or a + b - 1 > 1021:
        return None
    # binomial_ccdf(nn, kk) with nn = a + b - 1 = n, kk = a - 1: the sum runs from i = n - 1 down to kk + 1, b - 1 terms,
    # and its terms do not depend on kk -- the partial sums of one pass per n serve every k (same operations, same order)
    sums = _CCDF_SUMS.get(n)
    if sums is None:
        if len(_CCDF_SUMS) > 4096:
            _CCDF_SUMS.clear()
        result = 0.5 ** n
        term = result
        sums = [result]
        i = n - 1
        while i > 0 and len(sums) < 40:
            term *= ((i + 1) * 0.5) / ((n - i) * 0.5)
            result += term
            sums.append(result)
            i -= 1
        _CCDF_SUMS[n] = sums
    result = sums[b - 1]
    v = 1.0 - result if invert else result
    return min(max(v, 0.0), 1.0)             # rv_discrete.cdf / .sf clip to [0, 1]


_CCDF_SUMS = {}


def binom_two_sided(x, n, ask_scipy=True):
    """ss.binom_test(x, n, 0.5) (scoary/methods.py:1267-1275; scipy.stats.binomtest(...).pvalue since SciPy 1.12).
    At p = 1/2 the pmf is symmetric and SciPy's search for the far-side bound always lands on the mirror image of x
    (a ratio of neighbouring terms is never within its 1e-7 of 1), so the value is min(1, cdf(k) + sf(n - k - 1)) with
    k = min(x, n - x), 1 when x == n / 2.  Where SciPy's own arithmetic is restated (_boost_binom_tail_half: every n up
    to 78, the tails beyond) the result is SciPy's double -- the bytes the reference prints; for the few central (x, n)
    with 79 <= n <= 85 (tables of at most 170 isolates, whose result files are held to the reference's bytes) SciPy is
    asked (``ask_scipy``: the caller says whether this run is one of those -- importing scipy.stats costs 0.4 s);
    otherwise the exact dyadic tail 2 P(X <= k), rounded once (SciPy's value is within 4e-14 of it up to
    n = 400, 1e-13 in the far tail at n = 1500)."""
    x, n = int(x), int(n)
    if 2 * x == n:
        return 1.0
    k = min(x, n - x)
    lo, hi = _boost_binom_tail_half(k, n, False), _boost_binom_tail_half(n - k - 1, n, True)
    if lo is not None and hi is not None:
        return min(1.0, lo + hi)
    if n <= 85 and ask_scipy:
        try:
            import scipy.stats as ss
            if hasattr(ss, "binomtest"):
                return float(ss.binomtest(x, n, 0.5).pvalue)
            return float(ss.binom_test(x, n, 0.5))
        except ImportError:
            pass
    # prefix sums of C(n, 0..k), by the recurrence C(n, j+1) = C(n, j) (n-j) / (j+1), kept per n
    # (thousands of genes share a few hundred values of n)
    pre = _BINOM_PREFIX.get(n)
    if pre is None:
        if len(_BINOM_PREFIX) > 4096:
            _BINOM_PREFIX.clear()
        pre = _BINOM_PREFIX[n] = ([1], [1])             # C(n, j), sum_{i<=j} C(n, i)
    cs, sums = pre
    while len(cs) <= k:
        j = len(cs) - 1
        cs.append(cs[j] * (n - j) // (j + 1))
        sums.append(sums[j] + cs[j + 1])
    return float(min(Fraction(1), Fraction(2 * sums[k], 2 ** n)))


_BINOM_PREFIX = {}


def binom_two_sided_many(xs, ns):
    """binom_two_sided for arrays of (x, n): SciPy's double for every pair -- restated where Boost sums a finite
    series (_boost_binom_tail_half), and for the central pairs of n >= 79 (Boost's continued fraction, not restated)
    asked from SciPy itself in ONE vectorised binom.cdf / binom.sf call, the arithmetic behind binom_test /
    binomtest: min(1, cdf(k) + sf(n - k - 1)).  Without SciPy those pairs get the exact dyadic tail."""
    xs = np.asarray(xs, dtype=np.int64)
    ns = np.asarray(ns, dtype=np.int64)
    out = np.empty(xs.shape[0], dtype=np.float64)
    memo, ask = {}, []
    for i, (x, n) in enumerate(zip(xs.tolist(), ns.tolist())):
        v = memo.get((x, n))
        if v is None:
            if 2 * x == n:
                v = 1.0
            else:
                k = min(x, n - x)
                lo, hi = _boost_binom_tail_half(k, n, False), _boost_binom_tail_half(n - k - 1, n, True)
                v = min(1.0, lo + hi) if (lo is not None and hi is not None) else ("ask", k)
            memo[(x, n)] = v
        if isinstance(v, tuple):
            ask.append(i)
        else:
            out[i] = v
    if ask:
        ask = np.array(ask)
        n_a = ns[ask]
        k_a = np.minimum(xs[ask], n_a - xs[ask])
        try:
            from scipy.stats import binom
            out[ask] = np.minimum(1.0, binom.cdf(k_a, n_a, 0.5) + binom.sf(n_a - k_a - 1, n_a, 0.5))
        except ImportError:
            out[ask] = [binom_two_sided(int(x), int(n), ask_scipy=False) for x, n in zip(xs[ask], n_a)]
    return out


_ABORT_CACHE = {}


def _abort_thresholds_scipy(i):
    """Smallest r with 1 - binom.cdf(r, i, 0.1) < 0.05 for every i of the int64 array ``i``, with SciPy's binom.cdf
    exactly as the reference evaluates it (candidates bracketed by the 0.95 quantile)."""
    import scipy.stats as ss
    big = np.iinfo(np.int64).max
    r0 = ss.binom.ppf(0.95, i, 0.1).astype(np.int64)
    best = np.full(i.shape, big, dtype=np.int64)
    for d in (3, 2, 1, 0, -1, -2, -3):          # descending: smallest r wins last
        r = np.maximum(r0 + d, 0)
        ok = (1 - ss.binom.cdf(r, i, 0.1)) < 0.05
        best = np.where(ok, r, best)
    # guard the bracketing assumption: below `best` the test must fail
    below = np.maximum(best - 1, 0)
    bad = (best > 0) & ((1 - ss.binom.cdf(below, i, 0.1)) < 0.05)
    if bad.any() or (best == big).any():
        for k in np.nonzero(bad | (best == big))[0]:
            rr = np.arange(0, i[k] + 1)
            hit = np.nonzero((1 - ss.binom.cdf(rr, i[k], 0.1)) < 0.05)[0]
            best[k] = hit[0] if hit.size else big
    return best


def _lgamma(x):
    """log Gamma(x) for float64 arrays x >= 1, to ~1e-14 (absolute below 1, relative above): shifted to x + 16, then Stirling's series."""
    x = np.asarray(x, dtype=np.float64)
    shift = np.zeros_like(x)
    y = x.copy()
    for _ in range(16):
        shift += np.log(y)
        y += 1.0
    inv = 1.0 / y
    inv2 = inv * inv
    series = inv * (1.0 / 12 - inv2 * (1.0 / 360 - inv2 * (1.0 / 1260 - inv2 * (1.0 / 1680))))
    return (y - 0.5) * np.log(y) - y + 0.9189385332046727 + series - shift


def _abort_thresholds_direct(i):
    """The same thresholds without SciPy (importing scipy.stats costs 0.4 s of a 2.7 s run): the upper tail
    P(X > r), X ~ Binomial(i, 0.1), summed term by term from a window around the 0.95 quantile (pmf at the window's
    top from log-gamma, then the ratio recurrence both ways; ~1e-13 relative).  Returns (thresholds, undecided):
    ``undecided`` marks the i whose deciding tail is within 1e-9 of 0.05 or whose window does not bracket the
    answer -- the caller asks SciPy for those, so the last bit of ITS cdf decides as it does in the reference."""
    i = np.asarray(i, dtype=np.int64)
    fi = i.astype(np.float64)
    r_lo = np.maximum(np.floor(0.1 * fi + 1.645 * 0.3 * np.sqrt(fi)).astype(np.int64) - 4, 0)
    span = 9                                              # candidates r_lo .. r_lo + span - 1
    js = np.minimum(r_lo + span, i)                       # first term of the far tail
    fj = js.astype(np.float64)
    logp = (_lgamma(fi + 1.0) - _lgamma(fj + 1.0) - _lgamma(fi - fj + 1.0)
            + fj * np.log(0.1) + (fi - fj) * np.log(0.9))
    top = np.exp(logp)                                    # pmf(js)
    # far tail: sum_{j >= js} pmf(j) by pmf(j + 1) = pmf(j) (i - j) / (9 (j + 1)); ~10 sigma of terms is everything
    nterms = int(10 * 0.3 * np.sqrt(float(fi.max())) + 40)
    term, tail, j = top.copy(), top.copy(), fj.copy()
    for _ in range(nterms):
        term = term * np.maximum(fi - j, 0.0) / (9.0 * (j + 1.0))
        tail += term
        j += 1.0
    tail = np.where(r_lo + span > i, 0.0, tail)           # the window reaches past i: nothing beyond it
    # sf(r) for r = js - 1 down to r_lo: sf(r) = sf(r + 1) + pmf(r + 1), pmf downwards by the inverse ratio
    sf = np.empty((span, i.shape[0]))
    cur_sf, pm, jj = tail - top, top.copy(), fj.copy()    # cur_sf = sf(js) = sum_{j > js}
    cur_sf = np.where(r_lo + span > i, 0.0, cur_sf)
    # walk r from r_lo + span - 1 down to r_lo; where js was clipped to i the rows above i hold sf = 0
    for d in range(span - 1, -1, -1):
        r = r_lo + d
        active = r < js                                   # rows at or above js (clipped case) keep sf(js) = 0 ...
        add = np.where(active, pm, 0.0)                   # pmf(r + 1) with r + 1 == jj
        cur_sf = cur_sf + add
        sf[d] = np.where(r >= i, 0.0, cur_sf)
        step = active & (jj > 0)
        pm = np.where(step, pm * (9.0 * jj) / np.maximum(fi - jj + 1.0, 1.0), pm)
        jj = np.where(step, jj - 1.0, jj)
    ok = sf < 0.05
    first = np.argmax(ok, axis=0)                         # smallest d with sf < 0.05
    thr = r_lo + first
    col = np.arange(i.shape[0])
    undecided = ~ok.any(axis=0) | ((first == 0) & (r_lo > 0))          # not bracketed from below / above
    near = np.abs(sf - 0.05) < 1e-9
    undecided |= near[first, col] | near[np.maximum(first - 1, 0), col]
    return thr, undecided


def _abort_thresholds(P):
    """For i in [30, P): smallest r with 1 - binom.cdf(r, i, 0.1) < 0.05 (methods.py:1360-1361)."""
    if P not in _ABORT_CACHE:
        thr = np.full(P, np.iinfo(np.int64).max, dtype=np.int64)
        if P > 30:
            i = np.arange(30, P)
            if P > 50000:                                 # hours of permutations: SciPy's half second is nothing
                best = _abort_thresholds_scipy(i)
            else:
                best, undecided = _abort_thresholds_direct(i)
                if undecided.any():
                    best[undecided] = _abort_thresholds_scipy(i[undecided])
            thr[30:] = best
        _ABORT_CACHE[P] = thr
    return _ABORT_CACHE[P]


def empirical_p_sequential(exceed):
    """The reference's estimator (methods.py:1348-1365) over the exceedance
    flags of permutations 0..P-1: r accumulates; from i >= 30 on, the first i
    with 1 - binom.cdf(r, i, 0.1) < 0.05 returns (r+1)/(i+2); else (r+1)/(P+1)."""
    ex = np.asarray(exceed, dtype=np.int64)
    P = ex.shape[0]
    r = np.cumsum(ex)
    thr = _abort_thresholds(P)
    hit = np.nonzero(r >= thr)[0]
    if hit.size:
        i = int(hit[0])
        return (float(r[i]) + 1.0) / (i + 2.0)
    return (float(r[-1]) + 1.0) / (P + 1.0)


def empirical_p_sequential_many(exceed, chunk_elems=1 << 22):
    """empirical_p_sequential for every row of a (genes, P) array of exceedance flags, a few thousand rows at a time."""
    ex = np.asarray(exceed)
    K, P = ex.shape
    out = np.empty(K, dtype=np.float64)
    thr = _abort_thresholds(P)
    step = max(1, chunk_elems // max(P, 1))
    for k0 in range(0, K, step):
        r = np.cumsum(ex[k0:k0 + step], axis=1, dtype=np.int64)
        hit = r >= thr[None, :]
        any_hit = hit.any(axis=1)
        i = np.argmax(hit, axis=1)
        rows = np.arange(r.shape[0])
        out[k0:k0 + step] = np.where(any_hit, (r[rows, i].astype(np.float64) + 1.0) / (i + 2.0),
                                     (r[:, -1].astype(np.float64) + 1.0) / (P + 1.0))
    return out
