"""Large-shape cross-check on the GPU box: the list-driven and the dense permutation
kernels must give identical exceedance counts for every (gene, trait) pair.

    python tools/bigcheck.py
"""
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from scoary_amd import synth
from scoary_amd.engine import AssociationEngine, pack_bits_rows
eng = AssociationEngine(0)
for (G, N, T, P, kind) in [(200000, 5000, 1, 512, "rare"), (30000, 10000, 50, 128, "uniform"), (100000, 2559, 3, 1024, "uniform")]:
    rng = np.random.default_rng(G + N)
    genes = synth.make_genes(G, N, rng, kind=kind, core_frac=0.02)
    traits = synth.make_traits(T, N, rng, missing_traits=(0,))
    tb = pack_bits_rows((traits == 1).astype(np.uint8)); mb = pack_bits_rows((traits != 2).astype(np.uint8))
    gm = eng.pack_dense(genes)
    trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
    t = time.time(); d = eng.associate(gm, trv, mkv, permutations=P, seed=5, use_lists=False)["r"]; torch.cuda.synchronize(); td = time.time() - t
    eng.build_lists(gm)
    t = time.time(); l = eng.associate(gm, trv, mkv, permutations=P, seed=5, use_lists=True)["r"]; torch.cuda.synchronize(); tl = time.time() - t
    print(G, N, T, P, kind, "equal:", bool(torch.equal(d, l)), "dense %.3f s lists %.3f s" % (td, tl), "r sum", int(l.sum()))
