"""Would walking the LABEL's minority side pay?  

The list kernel walks every gene's minority list (length Lg) against label tiles.  For a
(gene, trait) pair whose trait has the shorter minority side (Lt = min(npos, nval - npos) < Lg)
the transposed kernel -- permutation lists of length Lt against bit tiles of 512 GENES --
would add fewer rows.  Genes are sorted by Lg, so per trait the pairs that would move are a
PREFIX [0, k_t) of the list order: the split costs no per-pair dispatch.

This script measures the walk of such a split on the box with the kernel that exists, because
the transposed kernel's inner loop IS k_permute_lists with the roles swapped (lists =
permutations, tile columns = genes; only its epilogue differs):

  full(t)   k_permute_lists on all G genes, trait t alone                      (today)
  rest(t)   k_permute_lists on the genes with Lg <= Lt, trait t alone          (what stays)
  moved(t)  k_permute_lists on P lists of exactly Lt entries x k_t tile columns: the
            transposed walk, (a) with the class-rotated entry order of spec S6 (needs a list
            generator that buckets by residue class), (b) entries in a random order (what the
            sequential sampler of spec S4 emits without that)

  sum_t rest + moved   vs   sum_t full   is an UPPER bound on the gain: the transposed kernel
additionally has to generate the permutation lists, test per-GENE regions from bit-plane
constants and reduce over permutations in a second kernel.

    python tools/ab_label_side.py [--config cfg3]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
from scoary_amd import synth  # noqa: E402
from scoary_amd.engine import AssociationEngine, pack_bits_rows  # noqa: E402
from ab_conflicts import shuffle_entries  # noqa: E402


def timed_walk(eng, gm, trv, mkv, P, steps=12, warmup=4):
    """Mean k_permute_lists time of back-to-back associate() steps."""
    T = trv.shape[0]
    ws = eng.workspace(gm, T, P, use_lists=True)
    for _ in range(warmup):
        eng.associate(gm, trv, mkv, permutations=P, seed=3, use_lists=True, workspace=ws)
    torch.cuda.synchronize()
    eng.set_timing(True)
    for _ in range(steps):
        eng.associate(gm, trv, mkv, permutations=P, seed=3, use_lists=True, workspace=ws)
    torch.cuda.synchronize()
    ms = eng.kernel_ms("k_permute_lists")
    eng.set_timing(False)
    return ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    args = ap.parse_args()
    eng = AssociationEngine(0)
    genes, traits, P, seed = synth.make_config(args.config)
    G, N = genes.shape
    T = traits.shape[0]
    rng = np.random.default_rng(5)
    ones = genes.sum(1, dtype=np.int64)
    Lg = np.minimum(ones, N - ones)
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    gm_all = eng.pack_dense(genes)
    eng.build_lists(gm_all)
    all_t = timed_walk(eng, gm_all, eng.vecrows(tb, N), eng.vecrows(mb, N), P)
    print("%s: k_permute_lists, all %d traits in one launch: %.3f ms" % (args.config, T, all_t), flush=True)
    tot = {"full": 0.0, "rest": 0.0, "moved_rotated": 0.0, "moved_unrotated": 0.0}
    entries_now = entries_split = 0
    for t in range(T):
        npos, nval = int((traits[t] == 1).sum()), int((traits[t] != 2).sum())
        Lt = min(npos, nval - npos)
        keep = Lg <= Lt                                   # pairs that stay with the gene-list kernel
        k_t = int((~keep).sum())
        trv, mkv = eng.vecrows(tb[t:t + 1], N), eng.vecrows(mb[t:t + 1], N)
        full = timed_walk(eng, gm_all, trv, mkv, P)
        gm_rest = eng.pack_dense(genes[keep])
        eng.build_lists(gm_rest)
        rest = timed_walk(eng, gm_rest, trv, mkv, P)
        # the transposed walk: P lists of exactly Lt entries, k_t tile columns ("permutations")
        lists = np.zeros((P, N), dtype=np.uint8)
        for i in range(P):
            lists[i, rng.choice(N, size=Lt, replace=False)] = 1
        gm_b = eng.pack_dense(lists)
        eng.build_lists(gm_b)
        cols = max(k_t, 1)
        moved_r = timed_walk(eng, gm_b, trv, mkv, cols) if k_t else 0.0
        shuffle_entries(eng, gm_b, rng)
        moved_u = timed_walk(eng, gm_b, trv, mkv, cols) if k_t else 0.0
        tot["full"] += full
        tot["rest"] += rest
        tot["moved_rotated"] += moved_r
        tot["moved_unrotated"] += moved_u
        entries_now += int(Lg.sum())
        entries_split += int(Lg[keep].sum()) + k_t * Lt
        print("trait %d: Lt=%d, %d of %d genes have the longer list | full %.3f  rest %.3f  moved "
              "%.3f (rotated) / %.3f (sampler order) ms" % (t, Lt, k_t, G, full, rest, moved_r, moved_u),
              flush=True)
        del gm_rest, gm_b
    print("row additions: %.4g now, %.4g split = %.1f %% fewer"
          % (entries_now, entries_split, 100.0 * (1 - entries_split / entries_now)))
    print("sum over traits: full %.3f ms | split, class-rotated permutation lists %.3f ms (%+.1f %%) | "
          "split, lists in sampler order %.3f ms (%+.1f %%)"
          % (tot["full"], tot["rest"] + tot["moved_rotated"],
             100.0 * ((tot["rest"] + tot["moved_rotated"]) / tot["full"] - 1),
             tot["rest"] + tot["moved_unrotated"],
             100.0 * ((tot["rest"] + tot["moved_unrotated"]) / tot["full"] - 1)))
    print("(upper bound on the gain: list generation, per-gene region constants and the reduction over "
          "permutations of a transposed kernel are not in these numbers)")


if __name__ == "__main__":
    main()
