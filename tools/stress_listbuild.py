"""Randomised cross-check on the GPU box: the device list builder (scoary_lists_plan / _fill)
against the host builder (scoary_lists_build, the checker) over many (G, N, density) shapes --
every array and every index entry must be identical.

    python tools/stress_listbuild.py [cases]
"""
import sys

import numpy as np

sys.path.insert(0, ".")
from scoary_amd import io_native  # noqa: E402
from scoary_amd.engine import AssociationEngine, pack_bits_rows  # noqa: E402

eng = AssociationEngine(0)
rng = np.random.default_rng(91)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 80
bad = 0
for case in range(cases):
    N = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 64, 100, 511, 1000, 2047, 2559, 2560, 3333, 5119, 5120,
                        7777, 10239, 10240, 13001, 20479, 20480, 33333, 40959]))
    G = int(rng.choice([1, 3, 15, 16, 17, 31, 33, 63, 64, 65, 300, 1000, 2047, 2049, 4097, 9000]))
    dens = rng.choice(["uniform", "sparse", "dense", "half", "ties"])
    f = {"uniform": rng.uniform(0, 1, (G, 1)), "sparse": rng.uniform(0, 0.03, (G, 1)),
         "dense": rng.uniform(0.97, 1, (G, 1)), "half": np.full((G, 1), 0.5),
         "ties": np.full((G, 1), 0.2)}[dens]
    genes = (rng.random((G, N)) < f).astype(np.uint8)
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
    bad += not ok
    print(case, G, N, dens, "ok" if ok else "MISMATCH")
print("mismatching cases:", bad)
sys.exit(1 if bad else 0)
