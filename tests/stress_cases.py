"""Randomised cross-checks of the list path's three device stages and of K1, one case per call (fixed
seeds): what tools/stress_{tiles,lists,listbuild}.py and tools/bigcheck.py used to run by
hand.  tests/test_gpu_stress.py runs them under `pytest -m gpu`; the tools are thin loops
over the same functions for longer soaks on a GPU box.

Every function returns (ok, description).
"""
import numpy as np

TILE_N = [1, 5, 63, 64, 65, 127, 128, 129, 500, 1000, 2559, 2560, 4000, 5120, 9000, 10240, 15000,
          20480, 31000, 40959, 40960, 70001, 131070]
LIST_N = [1, 2, 31, 32, 33, 64, 100, 511, 1000, 2047, 2048, 2559, 2560, 3333, 5119, 5120, 7777,
          10239, 10240, 13001, 20479, 20480, 33333, 40959, 40960, 61111, 100003]
BUILD_N = [1, 2, 15, 16, 17, 31, 33, 64, 100, 511, 1000, 2047, 2559, 2560, 3333, 5119, 5120, 7777,
           10239, 10240, 13001, 20479]


def _bits(traits):
    from scoary_amd.engine import pack_bits_rows
    return (pack_bits_rows((traits == 1).astype(np.uint8)),
            pack_bits_rows((traits != 2).astype(np.uint8)))


def _traits(rng, T, N, p_missing):
    traits = (rng.random((T, N)) < rng.uniform(0.05, 0.95)).astype(np.uint8)
    for t in range(T):
        if rng.random() < p_missing:
            traits[t, rng.random(N) < rng.uniform(0.0, 0.3)] = 2
    return traits


def tiles_case(eng, case, seed=11):
    """Label tiles (all three generator kernels) == transposed row-major labels of
    k_perm_generate: spec S4 draws, tile layout, zero row N, zero padding columns."""
    rng = np.random.default_rng([seed, case])
    N = int(rng.choice(TILE_N))
    T = int(rng.integers(1, 5))
    P = int(rng.choice([1, 31, 64, 65, 500, 513, 1200]))
    if case % 10 == 9:
        T, P = 3, 22000                                   # >= 1024 wavefronts: one-wavefront kernel
        N = int(rng.choice([64, 100, 333]))
    base = 32 * int(rng.integers(0, 32))               # tiles start at a multiple of 32 permutations
    traits = _traits(rng, T, N, 0.6)
    tb, mb = _bits(traits)
    masks, trv = eng.vecrows(mb, N), eng.vecrows(tb, N)
    _, margins = eng.counts(eng.pack_dense(np.ones((1, N), dtype=np.uint8)), trv, masks)
    rows = eng.perm_generate(masks, margins, N, P, base, 5 + case).cpu().numpy().view(np.uint32)
    tiles = eng.perm_generate_tiles(masks, margins, N, P, base, 5 + case).cpu().numpy().view(np.uint32)
    tw_, stride, _g, _c, _p = eng.list_params(N)
    RS, tperm = stride // 4, tw_ * 32
    ntiles = -(-P // tperm)
    tw = int(eng.lib.scoary_list_tile_words(N))
    S = int(eng.lib.scoary_list_segments(N))
    if S > 1:         # N > 20479: segments of 20352 two-dword rows, 40708 dwords apart, each with its zero row
        seg = tiles.reshape(T, ntiles, S, 40708)
        n_s = [min(20352, N - k * 20352) for k in range(S)]
        zero = all(not seg[:, :, k, 2 * n_s[k]:2 * n_s[k] + 2].any() for k in range(S))
        body = np.concatenate([seg[:, :, k, :2 * n_s[k]] for k in range(S)], axis=2)     # (T, ntiles, 2 N)
        tiles = np.concatenate([body, np.zeros((T, ntiles, 2), dtype=np.uint32) if zero
                                else np.ones((T, ntiles, 2), dtype=np.uint32)], axis=2).reshape(T, ntiles, N + 1, 2)
    else:
        tiles = tiles.reshape(T, ntiles, tw)[:, :, :(N + 1) * RS].reshape(T, ntiles, N + 1, RS)
    bits = np.unpackbits(rows.view(np.uint8).reshape(T, P, -1), axis=2, bitorder="little")[:, :, :N]
    ok = True
    for t in range(T):
        for tile in range(ntiles):
            tb_ = np.unpackbits(np.ascontiguousarray(tiles[t, tile]).view(np.uint8), axis=1,
                                bitorder="little")
            lo, hi = tile * tperm, min(P, tile * tperm + tperm)
            ok &= (not tb_[N].any() and np.array_equal(tb_[:N, :hi - lo], bits[t, lo:hi].T)
                   and not tb_[:N, hi - lo:].any())
    return bool(ok), "tiles N=%d T=%d P=%d base=%d" % (N, T, P, base)


def _gene_freq(rng, dens, G):
    return {"uniform": rng.uniform(0, 1, (G, 1)), "sparse": rng.uniform(0, 0.03, (G, 1)),
            "dense": rng.uniform(0.97, 1, (G, 1)), "half": np.full((G, 1), 0.5),
            "ties": np.full((G, 1), 0.2)}[dens]


def lists_case(eng, case, seed=23):
    """List-driven permutation kernel == dense kernel (itself oracle-checked in
    test_gpu_parity.py): identical r for every (gene, trait)."""
    import torch
    rng = np.random.default_rng([seed, case])
    N = int(rng.choice(LIST_N))
    G = int(rng.choice([1, 3, 15, 16, 17, 63, 64, 65, 300, 1000, 4097]))
    T = int(rng.integers(1, 4))
    P = int(rng.choice([1, 100, 128, 129, 512, 513, 700]))
    dens = str(rng.choice(["uniform", "sparse", "dense", "half"]))
    if N > 40959:                                    # long segmented lists: keep the host generation short
        G, P = min(G, 300), min(P, 129)
    genes = (rng.random((G, N)) < _gene_freq(rng, dens, G)).astype(np.uint8)
    traits = _traits(rng, T, N, 0.5)
    tb, mb = _bits(traits)
    gm = eng.pack_dense(genes)
    trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
    d = eng.associate(gm, trv, mkv, permutations=P, seed=case, use_lists=False)["r"]
    eng.build_lists(gm)
    l = eng.associate(gm, trv, mkv, permutations=P, seed=case, use_lists=True)["r"]
    return bool(torch.equal(d, l)), "lists G=%d N=%d T=%d P=%d %s" % (G, N, T, P, dens)


def listbuild_case(eng, case, seed=91):
    """Device list builder (scoary_lists_plan / _fill) == host builder (scoary_lists_build,
    the checker): every array and every index entry."""
    from scoary_amd import io_native
    from scoary_amd.engine import pack_bits_rows
    rng = np.random.default_rng([seed, case])
    N = int(rng.choice(BUILD_N))
    G = int(rng.choice([1, 3, 15, 16, 17, 31, 33, 63, 64, 65, 300, 1000, 2047, 2049, 4097, 9000]))
    dens = str(rng.choice(["uniform", "sparse", "dense", "half", "ties"]))
    genes = (rng.random((G, N)) < _gene_freq(rng, dens, G)).astype(np.uint8)
    if dens == "ties" and G > 4:                     # many equal lengths: the sort must be stable
        genes[G // 2:] = genes[:G - G // 2]
    gm = eng.pack_dense(genes)
    L = eng.build_lists(gm)
    lanes, stride, gpw, classes, piece = eng.list_params(N)
    H = io_native.build_lists(pack_bits_rows(genes), N, stride, gpw, classes, piece)
    ok = (L.entries == H["entries"]
          and np.array_equal(L.order.cpu().numpy(), H["order"])
          and np.array_equal(L.flipped.cpu().numpy(), H["flipped"])
          and np.array_equal(L.start.cpu().numpy(), H["start"])
          and np.array_equal(L.ngroups.cpu().numpy(), H["ngroups"])
          and np.array_equal(L.idx.cpu().numpy().view(np.uint32)[:L.entries], H["idx"][:L.entries]))
    return bool(ok), "listbuild G=%d N=%d %s" % (G, N, dens)


COUNT_N = [1, 31, 32, 33, 64, 127, 128, 129, 500, 511, 513, 1000, 2000, 2047, 2049, 5000, 10000, 20001, 50000]
COUNT_T = [1, 2, 3, 4, 5, 8, 9, 12, 13, 16, 17, 25, 28, 31, 32, 33, 50, 64, 65, 70, 97]


def counts_case(eng, case, seed=91):
    """K1 (round 4): k_counts through the trait plan -- every kernel instance (1, 2, 4, 8 ... 32
    traits per pass), one to four passes, one to sixteen wavefronts per 64 genes, shared / partly
    shared / all-distinct / all-missing validity rows -- against numpy's own AND + popcount on
    the dense arrays (tpgp, tpgn, tngp, tngn of scoary/methods.py:950-965), the plan's margins
    and mask classes against numpy, and the one-call scoary_counts against the planned form."""
    import ctypes
    import torch
    rng = np.random.default_rng([seed, case])
    N = int(rng.choice(COUNT_N))
    T = int(rng.choice(COUNT_T))
    G = int(rng.choice([1, 63, 64, 65, 300, 1000, 4097])) if N < 20000 else int(rng.choice([1, 65, 300]))
    genes = (rng.random((G, N)) < rng.uniform(0.0, 1.0, (G, 1))).astype(np.uint8)
    traits = (rng.random((T, N)) < rng.uniform(0.05, 0.95, (T, 1))).astype(np.uint8)
    kind = str(rng.choice(["none", "two", "pool", "distinct", "all-missing"]))
    if kind == "two":
        for t in rng.choice(T, size=min(2, T), replace=False):
            traits[t, rng.random(N) < 0.02] = 2
    elif kind == "pool":
        pool = [rng.random(N) < rng.uniform(0.0, 0.2) for _ in range(3)]
        for t in range(T):
            k = int(rng.integers(0, 4))
            if k < 3:
                traits[t, pool[k]] = 2
    elif kind == "distinct":
        for t in range(T):
            traits[t, rng.random(N) < 0.1] = 2
    elif kind == "all-missing":
        traits[int(rng.integers(0, T))] = 2
    tb, mb = _bits(traits)
    gm = eng.pack_dense(genes)
    trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
    plan = eng.trait_plan(trv, mkv, N)
    counts, _ = eng.counts(gm, trv, mkv, plan=plan)
    got = counts.cpu().numpy().astype(np.int64)
    g64 = genes.astype(np.int64)
    pos, val = (traits == 1).astype(np.int64), (traits != 2).astype(np.int64)
    a = pos @ g64.T                                                      # [T, G]
    gmv = val @ g64.T
    npos, nval = pos.sum(1)[:, None], val.sum(1)[:, None]
    want = np.stack([a, npos - a, gmv - a, nval - npos - gmv + a], axis=2)
    valid = traits != 2
    first, tpp = {}, int(eng.lib.scoary_counts_traits_per_pass(T))   # classes: per pass of tpp traits
    want_cls = np.array([first.setdefault((t // tpp, valid[t].tobytes()), t) for t in range(T)])
    m = plan.margins.cpu().numpy()
    ok = (np.array_equal(got, want) and np.array_equal(plan.mask_class.cpu().numpy(), want_cls)
          and np.array_equal(m[:, 0], npos[:, 0]) and np.array_equal(m[:, 1], nval[:, 0]))
    c2 = torch.full((T, G, 4), -7, dtype=torch.int32, device=eng.device)
    m2 = torch.zeros((T, 2), dtype=torch.int32, device=eng.device)
    vp = ctypes.c_void_p
    rc = eng.lib.scoary_counts(eng.h, vp(gm.tiled.data_ptr()), vp(trv.data_ptr()), vp(mkv.data_ptr()), G, T, N,
                               vp(c2.data_ptr()), vp(m2.data_ptr()), eng._stream())
    ok = ok and rc == 0 and bool(torch.equal(c2, counts)) and bool(torch.equal(m2, plan.margins))
    return bool(ok), "counts G=%d N=%d T=%d masks=%s" % (G, N, T, kind)


SEG_N = [20480, 20481, 31000, 40704, 40705, 61111, 100003, 131070]


def seglists_case(eng, case, seed=57):
    """Segmented lists (N > 20479: one sub-list per gene and 20 352-isolate segment, no host
    builder to compare with): order / flip of the whole row, and every sub-list holds exactly the
    gene's minority positions of its segment as 16-bit row indices (row - segment start), in
    grid-compaction order (class-aligned while every residue class has positions left), padded
    with the segment's zero row to the wave group's common length."""
    rng = np.random.default_rng([seed, case])
    N = int(rng.choice(SEG_N))
    G = int(rng.choice([1, 3, 63, 64, 65, 130, 300]))
    dens = str(rng.choice(["uniform", "sparse", "dense", "half", "ties"]))
    genes = (rng.random((G, N)) < _gene_freq(rng, dens, G)).astype(np.uint8)
    if dens == "ties" and G > 4:
        genes[G // 2:] = genes[:G - G // 2]
    if G > 2:
        genes[1, :20352] = genes[1, 0]                # a sub-list that is empty or full in segment 0
    gm = eng.pack_dense(genes)
    L = eng.build_lists(gm)
    S, SEG = int(eng.lib.scoary_list_segments(N)), 20352
    idx = L.idx.cpu().numpy().view(np.uint16)                      # 16-bit entries: row index in the segment
    start = L.start.cpu().numpy().astype(np.int64).reshape(S, G) * 64
    nhalf = L.ngroups.cpu().numpy().astype(np.int64).reshape(S, G)
    order, flipped = L.order.cpu().numpy(), L.flipped.cpu().numpy()
    ones = genes.sum(1, dtype=np.int64)
    lens = np.minimum(ones, N - ones)
    ok = (np.array_equal(flipped.astype(bool), 2 * ones > N) and np.array_equal(np.sort(order), np.arange(G))
          and bool(np.all(np.diff(lens[order]) <= 0))
          and np.array_equal(order, np.argsort(-lens, kind="stable")))       # stable length sort
    for k in range(G):
        q, j = divmod(k, 64)
        g = order[k]
        minority = genes[g] == (0 if flipped[g] else 1)
        for s in range(S):
            n_s = min(SEG, N - s * SEG)
            n = np.arange(nhalf[s, q * 64] * 16)
            at = start[s, q * 64] + ((n // 8) * 64 + j) * 8 + n % 8
            vals = idx[at]
            want = np.flatnonzero(minority[s * SEG:s * SEG + n_s])
            full = int(np.bincount(want % 32, minlength=32).min()) * 32     # grid rows no class has left yet
            ok = ok and len(want) <= len(vals) and np.array_equal(np.sort(vals[:len(want)]), want) \
                and bool(np.all(vals[len(want):] == n_s)) and (len(at) == 0 or at.max() < 2 * L.entries) \
                and nhalf[s, k] == nhalf[s, q * 64] \
                and np.array_equal(vals[:full] % 32, (k + np.arange(full)) % 32)
    return bool(ok), "seglists G=%d N=%d %s" % (G, N, dens)


BIG_SHAPES = [(200000, 5000, 1, 512, "rare"), (30000, 10000, 50, 128, "uniform"),
              (100000, 2559, 3, 1024, "uniform")]


def big_case(eng, shape):
    """Large shapes (one per tile width 8 / 4 / 16): list kernel == dense kernel on every
    (gene, trait) pair."""
    import torch
    from scoary_amd import synth
    G, N, T, P, kind = shape
    rng = np.random.default_rng(G + N)
    genes = synth.make_genes(G, N, rng, kind=kind, core_frac=0.02)
    traits = synth.make_traits(T, N, rng, missing_traits=(0,))
    tb, mb = _bits(traits)
    gm = eng.pack_dense(genes)
    trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
    d = eng.associate(gm, trv, mkv, permutations=P, seed=5, use_lists=False)["r"]
    eng.build_lists(gm)
    l = eng.associate(gm, trv, mkv, permutations=P, seed=5, use_lists=True)["r"]
    return bool(torch.equal(d, l)), "big G=%d N=%d T=%d P=%d %s" % shape
