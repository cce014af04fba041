"""What do LDS bank conflicts cost k_permute_lists?  

Same box, same list lengths, two gene sets per shape:
  random   -- every gene's minority positions drawn uniformly (what the BASELINE synthetic
              configs have): the residue classes of a list (position mod C, spec S6) hold
              unequal numbers of positions, the surplus of the fuller classes goes into the
              holes the emptier ones leave ("hole fillers") and those entries sit on the bank
              slot of another gene of their LDS service group;
  balanced -- the same minority count per gene (a multiple of C), but exactly count / C
              positions in every residue class: no hole fillers, every entry aligned, no
              conflict in any ds_read_b128.
  shuffled -- the random genes again, but the entries of every list put in a random order
              AFTER the device builder (the class rotation of spec S6 undone): what a list
              whose order cannot be arranged would cost -- e.g. the permutation lists of a
              "walk the label's minority side" kernel, which come out
              of the sequential sampler in isolate order.
random - balanced is everything the remaining conflicts cost; shuffled - random is what the
class rotation is worth.  Run under rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE to
see the counters move (tools/ab_conflicts.sh).

    python tools/ab_conflicts.py [--shapes cfg3 cfg4 cfg5] [--variant random|balanced|both]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from scoary_amd import synth  # noqa: E402
from scoary_amd.engine import AssociationEngine, pack_bits_rows  # noqa: E402

SHAPES = {   # G, N, T, P, gene kind: the BASELINE shapes, cfg4 / cfg5 with fewer genes (the per-gene host
             # loop that balances the classes is slow; all variants get the same genes, lengths and geometry)
    "cfg3": (50_000, 2_000, 10, 10_000, "uniform"),
    "cfg4": (25_000, 5_000, 1, 10_000, "rare"),
    "cfg5": (7_500, 10_000, 50, 12_800, "uniform"),
    # round 6: the segmented kernel (k_permute_seglists, N > 20479; 16-bit entries, ds_read_b64 of two-dword rows):
    # `balanced` is balanced inside every isolate segment (each segment has its own sub-list)
    "wide": (6_000, 50_000, 2, 1_024, "uniform"),
}
SEG_ROWS = 20352                                               # kSegRows (scoary_common.hpp)


def gene_sets(G, N, C, kind, rng):
    """(random, balanced): two (G, N) uint8 matrices with identical per-gene minority counts."""
    base = synth.make_genes(G, N, rng, kind=kind)
    ones = base.sum(1, dtype=np.int64)
    flip = 2 * ones > N
    L = np.where(flip, N - ones, ones) // C * C               # minority count, multiple of C
    L = np.minimum(L, (N // C) * C // 2 // C * C)
    rnd = np.zeros((G, N), dtype=np.uint8)
    bal = np.zeros((G, N), dtype=np.uint8)
    segs = [(a, min(N, a + SEG_ROWS)) for a in range(0, N, SEG_ROWS)] if N > 20479 else [(0, N)]
    per_class = [[np.arange(a + (c - a) % C, b, C) for c in range(C)] for a, b in segs]
    for g in range(G):
        if L[g] == 0:
            continue
        rnd[g, rng.choice(N, size=L[g], replace=False)] = 1
        for (a, b), pcs in zip(segs, per_class):
            k = int(L[g] * (b - a) / N) // C                   # the same count in every class of the segment
            for c in range(C):
                bal[g, rng.choice(pcs[c], size=min(k, len(pcs[c])), replace=False)] = 1
    inv = flip[:, None].astype(np.uint8)                       # keep which value is the minority
    return rnd ^ inv, bal ^ inv


def shuffle_entries(eng, gm, rng):
    """Undo spec S6's entry order: every list's real entries (not its zero-row padding) in a
    random order, in place in the device index array.  Layout (scoary_lists_fill): entry n of
    slot j of wave group q sits at base(q) + ((n // piece) * gpw + j) * piece + n % piece."""
    L = gm.lists
    _tw, stride, gpw, _C, piece = eng.list_params(gm.N)
    idx = L.idx.cpu().numpy().view(np.uint32).copy()
    start = L.start.cpu().numpy().astype(np.int64) * 32          # entries (kListStartUnit)
    nhalf = L.ngroups.cpu().numpy().astype(np.int64)
    zero_row = np.uint32(gm.N * stride)
    for k in range(gm.G):
        q, j = divmod(k, gpw)
        n = np.arange(nhalf[q * gpw] * 16)
        at = start[q * gpw] + ((n // piece) * gpw + j) * piece + n % piece
        vals = idx[at]
        real = vals != zero_row
        m = int(real.sum())
        assert real[:m].all()                                      # padding sits at the tail
        idx[at[:m]] = rng.permutation(vals[:m])
    L.idx.copy_(torch.from_numpy(idx.view(np.int32)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=["cfg3", "cfg4", "cfg5"])
    ap.add_argument("--variant", default="all", choices=["random", "balanced", "shuffled", "all"])
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--genes-scale", type=float, default=1.0)
    args = ap.parse_args()
    eng = AssociationEngine(0)
    for name in args.shapes:
        G, N, T, P, kind = SHAPES[name]
        G = int(G * args.genes_scale)
        rng = np.random.default_rng(7)
        tw, _stride, _gpw, C, _piece = eng.list_params(N)
        rnd, bal = gene_sets(G, N, C, kind, rng)
        traits = synth.make_traits(T, N, rng)
        tb = pack_bits_rows((traits == 1).astype(np.uint8))
        mb = pack_bits_rows((traits != 2).astype(np.uint8))
        trv, mkv = eng.vecrows(tb, N), eng.vecrows(mb, N)
        out = {}
        ref_r = None
        k3 = eng.list_kernel_name(N)
        for variant, genes in (("random", rnd), ("balanced", bal), ("shuffled", rnd)):
            if args.variant not in ("all", variant) or (variant == "shuffled" and k3 != "k_permute_lists"):
                continue
            gm = eng.pack_dense(genes)
            eng.build_lists(gm)
            if variant == "shuffled":
                shuffle_entries(eng, gm, rng)
            ws = eng.workspace(gm, T, P, use_lists=True)
            # sustained: the steps run back to back like bench.py's (a synchronize between steps
            # lets the clock drop and the kernel restart below the power cap)
            for _ in range(args.warmup):
                eng.associate(gm, trv, mkv, permutations=P, seed=3, use_lists=True, workspace=ws)
            torch.cuda.synchronize()
            eng.set_timing(True)
            for _ in range(args.steps):
                eng.associate(gm, trv, mkv, permutations=P, seed=3, use_lists=True, workspace=ws)
            torch.cuda.synchronize()
            ms = [eng.kernel_ms(k3)]                             # mean over the timed launches
            eng.set_timing(False)
            if variant == "random":
                ref_r = ws.r.clone()
            elif variant == "shuffled" and ref_r is not None:      # the order never changes the counts
                assert torch.equal(ref_r, ws.r), "shuffled lists changed r"
            out[variant] = sorted(ms)[len(ms) // 2]
            print("%-5s %-9s G=%d N=%d T=%d P=%d C=%d entries=%d  %s %.3f ms (mean of %d back-to-back steps)"
                  % (name, variant, G, N, T, P, C, gm.lists.entries, k3, out[variant], args.steps), flush=True)
            del ws, gm
        if "random" in out and "balanced" in out:
            print("%-5s remaining conflicts (hole fillers) cost %.2f %% of the kernel time" %
                  (name, 100.0 * (out["random"] - out["balanced"]) / out["random"]), flush=True)
        if "random" in out and "shuffled" in out:
            print("%-5s without the class rotation the kernel takes %.2f %% longer" %
                  (name, 100.0 * (out["shuffled"] - out["random"]) / out["random"]), flush=True)


if __name__ == "__main__":
    main()
