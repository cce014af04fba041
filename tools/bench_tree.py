#!/usr/bin/env python3
"""Throughput of the population-structure kernels: Hamming counts,
PhyloTree maxima (k_tree_dp) and tree-statistic permutations, on synthetic
data, with the C oracle timed beside them.  Prints one JSON line.

    python tools/bench_tree.py [--isolates 2000] [--genes 500] [--permutations 2000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--isolates", type=int, default=2000)
    ap.add_argument("--genes", type=int, default=500)
    ap.add_argument("--all-genes", type=int, default=20000)
    ap.add_argument("--permutations", type=int, default=2000)
    ap.add_argument("--kernels-only", action="store_true",
                    help="k_hamming and k_tree_dp only (counter passes: the device UPGMA loop launches thousands of "
                         "tiny kernels, each one serialised by rocprofv3 --pmc)")
    args = ap.parse_args()
    import torch
    from oracle import oracle as orc
    from scoary_amd import methods as m
    from scoary_amd import tree as T
    from scoary_amd.engine import AssociationEngine, pack_bits_rows

    rng = np.random.default_rng(1)
    N, G, P = args.isolates, args.genes, args.permutations
    dense = (rng.random((args.all_genes, N)) < rng.uniform(0.05, 0.95, (args.all_genes, 1))).astype(np.uint8)
    strains = ["iso%d" % i for i in range(N)]
    eng = AssociationEngine(0)
    out = {"isolates": N, "genes_for_tree": args.all_genes, "survivor_genes": G, "permutations": P}

    t0 = time.perf_counter()
    counts = eng.hamming(np.ascontiguousarray(dense.T))
    torch.cuda.synchronize()
    out["hamming_s_incl_transfer"] = time.perf_counter() - t0
    eng.set_timing(True)
    eng.hamming(np.ascontiguousarray(dense.T))
    out["k_hamming_ms"] = eng.kernel_ms("k_hamming")
    eng.set_timing(False)
    out["k_hamming_pair_words_per_s"] = N * N * (args.all_genes / 32.0) / (out["k_hamming_ms"] * 1e-3)
    t0 = time.perf_counter()
    tree = T.upgma_from_counts(counts, args.all_genes, strains)
    out["upgma_host_library_s"] = time.perf_counter() - t0
    if not args.kernels_only:
        rows01 = np.ascontiguousarray(dense.T)
        eng.upgma_merges(rows01[:64])                       # warm-up
        t0 = time.perf_counter()
        tree_dev = T.upgma(eng, dense, strains)             # Hamming + merge loop on the device
        out["upgma_device_s_incl_hamming_and_transfer"] = time.perf_counter() - t0
        out["upgma_device_equals_host"] = tree_dev == tree

    trait = (rng.random(N) < 0.4).astype(np.uint8)
    stage = m._TreeStage(eng, tree, strains, trait, 0, 5)
    out["stack_depth"] = stage.prog.depth
    rows = pack_bits_rows(dense[:G])
    obs = stage.observed(rows)
    torch.cuda.synchronize()
    eng.set_timing(True)
    t0 = time.perf_counter()
    ex = stage.permute(rows, obs, P)
    torch.cuda.synchronize()
    out["tree_permute_wall_s"] = time.perf_counter() - t0
    out["k_tree_dp_ms"] = eng.kernel_ms("k_tree_dp")
    eng.set_timing(False)
    evals = G * P
    out["tree_evaluations"] = evals
    out["tree_evaluations_per_s_kernel"] = evals / (out["k_tree_dp_ms"] * 1e-3)
    out["tree_merges_per_s_kernel"] = evals * (N - 1) / (out["k_tree_dp_ms"] * 1e-3)
    # CPU oracle on a sample (one thread)
    index_of = {s: i for i, s in enumerate(strains)}
    ops, tips = orc.tree_program(tree, index_of)
    gb = orc.pack_rows(dense[:2])
    tb = orc.pack_rows((trait == 1)[None])[0]
    mb = orc.pack_rows(np.ones((1, N), dtype=np.uint8))[0]
    t0 = time.perf_counter()
    Ps = 200
    o, exo = orc.tree_permute(ops, tips, gb[0], tb, mb, N, 0, Ps, 5)
    dt = time.perf_counter() - t0
    out["oracle_tree_evaluations_per_s_1core"] = Ps / dt
    out["oracle_agrees"] = bool(np.array_equal(exo, ex[0, :Ps]) and tuple(obs[0]) == o)
    # roofline objects (round 6).  Both kernels are integer-VALU work, as K1 / K3 are:
    #   k_hamming: the K1 family -- 2 lane-ops (XOR, popcount-accumulate) per 32-bit word and isolate PAIR;
    #     ceiling = the measured rate of that op pair (tools/valu_peak.hip: 4.1e13 lane-ops/s), and the
    #     algorithmic bytes (bit matrix once + N^2 counts) against 8 TB/s;
    #   k_tree_dp: lane = one (gene, permutation) tree evaluation, tips - 1 node merges of ~80 VALU ops
    #     (two packed integer keys per state: adds and v_max); ceiling = the nominal SIMD peak, 78.6 T lane-ops/s.
    # SQ_INSTS_VALU x 64 from the committed PMC pass (tools/round6_run.sh tree) replaces the op estimates.
    W32 = (args.all_genes + 31) // 32
    ham_ops = 2.0 * N * N * W32
    ham_bytes = N * W32 * 4.0 + 4.0 * N * N
    out["roofline_k_hamming"] = {
        "bound": "valu (XOR + popcount-accumulate)", "lane_ops": ham_ops,
        "achieved_lane_ops_per_s": ham_ops / (out["k_hamming_ms"] * 1e-3), "peak_lane_ops_per_s": 4.1e13,
        "frac": ham_ops / (out["k_hamming_ms"] * 1e-3) / 4.1e13,
        "frac_of_nominal_simd_peak": ham_ops / (out["k_hamming_ms"] * 1e-3) / (256 * 4 * 32 * 2.4e9),
        "bytes": ham_bytes, "gbs": ham_bytes / (out["k_hamming_ms"] * 1e-3) / 1e9,
        "hbm_frac": ham_bytes / (out["k_hamming_ms"] * 1e-3) / 1e9 / 8000.0}
    merges = float(evals) * (N - 1)
    out["roofline_k_tree_dp"] = {
        "bound": "valu (node merges: integer add / max on packed keys)", "merges": merges,
        "merges_per_s": merges / (out["k_tree_dp_ms"] * 1e-3),
        "ops_per_merge_estimate": 80, "achieved_lane_ops_per_s_estimate": 80 * merges / (out["k_tree_dp_ms"] * 1e-3),
        "peak_lane_ops_per_s": 256 * 4 * 32 * 2.4e9,
        "frac_estimate": 80 * merges / (out["k_tree_dp_ms"] * 1e-3) / (256 * 4 * 32 * 2.4e9)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
