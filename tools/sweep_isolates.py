"""List-driven vs dense permutation kernels across the isolate count (every tile width of the
list path: 16 / 8 / 4 / 2 / 1 dwords, and the segmented tiles of N > 40959), uniform and
rare-variant genes; G is chosen so that the matrix has ~1e8 cells.  Sustained timing of the
permutation kernel alone and of the whole step (counts + Fisher + labels + exceedance).

    python tools/sweep_isolates.py [--permutations 1024]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scoary_amd import synth  # noqa: E402
from scoary_amd.engine import AssociationEngine, pack_bits_rows  # noqa: E402


def timed(eng, gm, trv, mkv, P, use_lists, steps=6, warmup=2):
    ws = eng.workspace(gm, trv.shape[0], P, use_lists=use_lists)
    for _ in range(warmup):
        eng.associate(gm, trv, mkv, permutations=P, seed=3, use_lists=use_lists, workspace=ws)
    torch.cuda.synchronize()
    eng.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.associate(gm, trv, mkv, permutations=P, seed=3, use_lists=use_lists, workspace=ws)
    torch.cuda.synchronize()
    step_ms = (time.perf_counter() - t0) / steps * 1e3
    k_ms = eng.kernel_ms(eng.list_kernel_name(gm.N) if use_lists else "k_permute")
    eng.set_timing(False)
    return step_ms, k_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--permutations", type=int, default=1024)
    ap.add_argument("--traits", type=int, default=2)
    args = ap.parse_args()
    eng = AssociationEngine(0)
    P, T = args.permutations, args.traits
    print("# N, gene kind, G, tile row dwords (segments) | whole step ms: lists / dense | permutation kernel ms: "
          "lists / dense | tests/s lists | speed-up of the step")
    for N in (500, 2000, 2559, 5000, 10000, 20000, 40959, 50000, 100000, 122496):
        G = max(2048, int(1e8 // N) // 64 * 64)
        for kind in ("uniform", "rare"):
            rng = np.random.default_rng(N)
            genes = synth.make_genes(G, N, rng, kind=kind)
            traits = synth.make_traits(T, N, rng)
            tb = pack_bits_rows((traits == 1).astype(np.uint8))
            mb = pack_bits_rows((traits != 2).astype(np.uint8))
            gm = eng.pack_dense(genes)
            trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
            d_step, d_k = timed(eng, gm, trv, mkv, P, False)
            eng.build_lists(gm)
            l_step, l_k = timed(eng, gm, trv, mkv, P, True)
            tw = eng.list_params(N)[0]
            seg = int(eng.lib.scoary_list_segments(N))
            print("N=%-6d %-7s G=%-6d TW=%-2d%s | step %8.3f / %8.3f ms | kernel %8.3f / %8.3f ms | %.3e tests/s | x%.2f"
                  % (N, kind, G, tw, " (%d seg)" % seg if seg > 1 else "        ", l_step, d_step, l_k, d_k,
                     G * T * P / (l_step * 1e-3), d_step / l_step), flush=True)
            del gm


if __name__ == "__main__":
    main()
