"""Randomised cross-check on the GPU box: label tiles (both generator kernels) against
the row-major labels of k_perm_generate, over many (N, T, P, missing-value) shapes.

    python tools/stress_tiles.py [cases]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from scoary_amd.engine import AssociationEngine, pack_bits_rows  # noqa: E402

eng = AssociationEngine(0)
rng = np.random.default_rng(11)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for case in range(cases):
    N = int(rng.choice([1, 5, 63, 64, 65, 127, 128, 129, 500, 1000, 2559, 2560, 4000, 5120, 9000, 10240, 15000, 20480, 31000, 40959]))
    T = int(rng.integers(1, 5))
    P = int(rng.choice([1, 31, 64, 65, 500, 513, 1200]))
    if case % 10 == 9:
        T, P = 3, 22000                                   # >= 1024 wavefronts: one-wavefront kernel
        N = int(rng.choice([64, 100, 333]))
    base = int(rng.integers(0, 1000))
    traits = (rng.random((T, N)) < rng.uniform(0.05, 0.95)).astype(np.uint8)
    for t in range(T):
        if rng.random() < 0.6:
            traits[t, rng.random(N) < rng.uniform(0.0, 0.3)] = 2
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    masks, trv = eng.vecrows(mb, N), eng.vecrows(tb, N)
    _, margins = eng.counts(eng.pack_dense(np.ones((1, N), dtype=np.uint8)), trv, masks)
    rows = eng.perm_generate(masks, margins, N, P, base, 5 + case).cpu().numpy().view(np.uint32)
    tiles = eng.perm_generate_tiles(masks, margins, N, P, base, 5 + case).cpu().numpy().view(np.uint32)
    tw_, stride, _g, _c, _p = eng.list_params(N)
    RS, tperm = stride // 4, tw_ * 32
    ntiles = -(-P // tperm)
    tw = int(eng.lib.scoary_list_tile_words(N))
    tiles = tiles.reshape(T, ntiles, tw)[:, :, :(N + 1) * RS].reshape(T, ntiles, N + 1, RS)
    bits = np.unpackbits(rows.view(np.uint8).reshape(T, P, -1), axis=2, bitorder="little")[:, :, :N]
    ok = True
    for t in range(T):
        for tile in range(ntiles):
            tb_ = np.unpackbits(np.ascontiguousarray(tiles[t, tile]).view(np.uint8), axis=1,
                                bitorder="little")
            lo, hi = tile * tperm, min(P, tile * tperm + tperm)
            ok &= not tb_[N].any() and np.array_equal(tb_[:N, :hi - lo], bits[t, lo:hi].T) \
                and not tb_[:N, hi - lo:].any()
    bad += not ok
    print(case, N, T, P, "ok" if ok else "MISMATCH")
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
