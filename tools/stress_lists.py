"""Randomised cross-check on the GPU box: list-driven vs dense permutation kernels
(and a subsample against nothing else -- the dense kernel is oracle-checked in
tests/) over many (G, N, T, P, density, missing-value) shapes.

    python tools/stress_lists.py [cases]
"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from scoary_amd.engine import AssociationEngine, pack_bits_rows  # noqa: E402

eng = AssociationEngine(0)
rng = np.random.default_rng(23)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for case in range(cases):
    N = int(rng.choice([1, 2, 31, 32, 33, 64, 100, 511, 1000, 2047, 2048, 2559, 2560, 3333, 5119, 5120,
                        7777, 10239, 10240, 13001, 20479, 20480, 33333, 40959]))
    G = int(rng.choice([1, 3, 15, 16, 17, 63, 64, 65, 300, 1000, 4097]))
    T = int(rng.integers(1, 4))
    P = int(rng.choice([1, 100, 128, 129, 512, 513, 700]))
    dens = rng.choice(["uniform", "sparse", "dense", "half"])
    f = {"uniform": rng.uniform(0, 1, (G, 1)), "sparse": rng.uniform(0, 0.03, (G, 1)),
         "dense": rng.uniform(0.97, 1, (G, 1)), "half": np.full((G, 1), 0.5)}[dens]
    genes = (rng.random((G, N)) < f).astype(np.uint8)
    traits = (rng.random((T, N)) < rng.uniform(0.05, 0.95)).astype(np.uint8)
    for t in range(T):
        if rng.random() < 0.5:
            traits[t, rng.random(N) < rng.uniform(0.0, 0.3)] = 2
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    gm = eng.pack_dense(genes)
    trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
    d = eng.associate(gm, trv, mkv, permutations=P, seed=case, use_lists=False)["r"]
    eng.build_lists(gm)
    l = eng.associate(gm, trv, mkv, permutations=P, seed=case, use_lists=True)["r"]
    ok = bool(torch.equal(d, l))
    bad += not ok
    print(case, G, N, T, P, dens, "ok" if ok else "MISMATCH")
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
