#!/usr/bin/env python3
"""EVIDENCE ONLY, not product (north_star excludes MFMA): the matrix-core formulation of the
permutation exceedance counts END TO END (tools/mfma_count_e2e.hip) against the product's kernels, on the
BASELINE shapes with P = 10240 (a multiple of the 128-permutation block tile):

    python tools/mfma_count_e2e.py [--config cfg3|cfg4] [--genes G] [--reps 5]

For each config: the same tiled gene matrix, the same label rows (scoary_perm_generate) and the same regions
(scoary_fisher) go through (1) the MFMA kernel, (2) the dense AND + popcount kernel, (3) the list-driven kernel;
r must be bit-identical for all three; the time of each is the mean of --reps launches (hipEvents).  Nothing here
is imported by the product; the shared object is built on the fly with hipcc if missing.
"""
import argparse
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SRC = os.path.join(ROOT, "tools", "mfma_count_e2e.hip")
LIB = os.path.join(ROOT, "tools", "mfma_count_e2e.so")


def build():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", SRC, "-o", LIB])
    lib = ctypes.CDLL(LIB)
    i64, vp = ctypes.c_int64, ctypes.c_void_p
    lib.mfma_exceed.restype = ctypes.c_int
    lib.mfma_exceed.argtypes = [vp, i64, i64, vp, i64, i64, vp, i64, vp, ctypes.c_int, vp]
    return lib


def timed(torch, fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3", choices=["cfg3", "cfg4"])
    ap.add_argument("--genes", type=int, default=None)
    ap.add_argument("--permutations", type=int, default=10240)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import torch
    from scoary_amd import synth
    from scoary_amd.engine import AssociationEngine, pack_bits_rows
    lib = build()
    eng = AssociationEngine(0)
    genes, traits, _P, seed = synth.make_config(a.config, G=a.genes)
    P = a.permutations
    assert P % 128 == 0
    G, N = genes.shape
    T = traits.shape[0]
    gm = eng.tile_rows(pack_bits_rows(genes), N)
    trv = eng.vecrows(pack_bits_rows((traits == 1).astype(np.uint8)), N)
    mkv = eng.vecrows(pack_bits_rows((traits != 2).astype(np.uint8)), N)
    plan = eng.trait_plan(trv, mkv, N)
    counts, margins = eng.counts(gm, trv, mkv, plan=plan)
    p, odds, crit = eng.fisher(counts)
    perms = eng.perm_generate(mkv, margins, N, P, 0, seed)               # [T, P, Wp] label rows
    Qp, Gp = eng.quads(N), eng.padded_genes(G)
    vp = ctypes.c_void_p
    r_mfma = torch.full((T, G), 0x7fffffff, dtype=torch.int32, device=eng.device)

    def run_mfma(variant=0, out=None):
        rc = lib.mfma_exceed(vp(gm.tiled.data_ptr()), Gp, Qp, vp(perms.data_ptr()), T, P, vp(crit.data_ptr()), G,
                             vp((out if out is not None else r_mfma).data_ptr()), variant, eng._stream())
        assert rc == 0, rc
    r_dense = torch.zeros((T, G), dtype=torch.int32, device=eng.device)

    def run_dense():
        r_dense.zero_()
        eng.permute(gm, perms, crit, r_dense, P=P)
    eng.build_lists(gm)
    tiles = eng.perm_generate_tiles(mkv, margins, N, P, 0, seed)
    scratch = eng.permute_lists_scratch(G, T, N, P)
    r_lists = torch.zeros((T, G), dtype=torch.int32, device=eng.device)

    def run_lists():
        eng.permute_lists(gm, tiles, crit, margins, P, r_lists, scratch=scratch, accumulate=False)
    t_mfma = timed(torch, run_mfma, a.reps)
    t_res, same_res = None, None
    if Qp == 16:                                           # gene rows resident in LDS (N <= 2048)
        r_res = torch.full((T, G), 0x7fffffff, dtype=torch.int32, device=eng.device)
        t_res = timed(torch, lambda: run_mfma(1, r_res), a.reps)
        same_res = bool(torch.equal(r_res, r_mfma))
    r_v3 = torch.full((T, G), 0x7fffffff, dtype=torch.int32, device=eng.device)
    t_v3 = timed(torch, lambda: run_mfma(2, r_v3), a.reps)
    same_v3 = bool(torch.equal(r_v3, r_mfma))
    t_dense = timed(torch, run_dense, a.reps)
    t_lists = timed(torch, run_lists, a.reps)
    same_d = bool(torch.equal(r_mfma, r_dense))
    same_l = bool(torch.equal(r_mfma, r_lists))
    tests = G * T * P
    macs = float(Gp) * T * P * Qp * 128
    ones = genes.sum(1, dtype=np.int64)
    print("# %s shape: %d genes x %d isolates x %d traits x %d permutations = %.3e tests; minority fraction of the genes %.3f"
          % (a.config, G, N, T, P, tests, float(np.minimum(ones, N - ones).mean()) / N))
    print("MFMA end to end (1-bit operands from HBM, fp4 expansion in registers, region test on the accumulators): "
          "%8.3f ms  %.3e tests/s  %.3e dense MAC/s   r == dense kernel: %s, == list kernel: %s"
          % (t_mfma, tests / (t_mfma * 1e-3), macs / (t_mfma * 1e-3), same_d, same_l))
    if t_res is not None:
        print("MFMA end to end, gene rows expanded once and resident in LDS (N <= 2048):                           "
              "%8.3f ms  %.3e tests/s  %.3e dense MAC/s   r == streaming MFMA: %s   MFMA / lists = %.2f"
              % (t_res, tests / (t_res * 1e-3), macs / (t_res * 1e-3), same_res, t_res / t_lists))
        same_d = same_d and same_res
    print("MFMA end to end, v_perm expansion + per-lane exceedance counters (no ballots per tile):             "
          "%8.3f ms  %.3e tests/s  %.3e dense MAC/s   r == streaming MFMA: %s   MFMA / lists = %.2f"
          % (t_v3, tests / (t_v3 * 1e-3), macs / (t_v3 * 1e-3), same_v3, t_v3 / t_lists))
    same_d = same_d and same_v3
    print("dense AND + popcount kernel (k_permute_reg / chunked):                                               "
          "%8.3f ms  %.3e tests/s" % (t_dense, tests / (t_dense * 1e-3)))
    print("list-driven kernel (k_permute_lists + reduce, regions converted):                                    "
          "%8.3f ms  %.3e tests/s   MFMA / lists = %.2f" % (t_lists, tests / (t_lists * 1e-3), t_mfma / t_lists))
    if not (same_d and same_l):
        bad = (r_mfma != r_dense).nonzero()
        print("MISMATCH: first differing (trait, gene):", bad[:5].tolist(),
              r_mfma[r_mfma != r_dense][:5].tolist(), r_dense[r_mfma != r_dense][:5].tolist())
        raise SystemExit(1)


if __name__ == "__main__":
    main()
