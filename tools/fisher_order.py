#!/usr/bin/env python3
"""k_fisher in table order (scoary_fisher) against list-slot order (scoary_fisher_lists),
in isolation, on the BASELINE shapes:  python tools/fisher_order.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scoary_amd import synth                                    # noqa: E402
from scoary_amd.engine import AssociationEngine, pack_bits_rows  # noqa: E402

eng = AssociationEngine(0)
for name, G, N, T, kind in (("cfg2", 10000, 500, 1, "uniform"), ("cfg3", 50000, 2000, 10, "uniform"),
                            ("cfg4", 200000, 5000, 1, "rare"), ("20k x 1000 x 1", 20000, 1000, 1, "uniform")):
    rng = np.random.default_rng(1)
    genes = synth.make_genes(G, N, rng, kind=kind)
    traits = synth.make_traits(T, N, rng)
    gm = eng.tile_rows(pack_bits_rows(genes), N)
    eng.build_lists(gm)
    trv = eng.vecrows(pack_bits_rows((traits == 1).astype(np.uint8)), N)
    mkv = eng.vecrows(pack_bits_rows((traits != 2).astype(np.uint8)), N)
    counts, margins = eng.counts(gm, trv, mkv)
    out = []
    for lists in (None, gm.lists):
        for _ in range(5):
            eng.fisher(counts, lists=lists)
        torch.cuda.synchronize()
        eng.set_timing(True)
        for _ in range(30):
            eng.fisher(counts, lists=lists)
        torch.cuda.synchronize()
        out.append(eng.kernel_ms("k_fisher") * 1e3)
        eng.set_timing(False)
    print("%-16s table order %8.1f us   slot order %8.1f us" % (name, out[0], out[1]), flush=True)
