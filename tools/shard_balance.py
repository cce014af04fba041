#!/usr/bin/env python3
"""What each rank of a gene-sharded run would do, measured one shard after the other on ONE GPU
(round 6): for every rank of a `world`-way split of a config -- contiguous equal-count
blocks (rounds 1-5) and the reference's stride domains (scoary/methods.py:1076-1078; dist.GenePartition)
-- the shard is set up exactly as bench.py sets it up and its step is timed alone, so the per-rank
times are free of the time-sharing of a shared-GPU rehearsal.  The slowest rank is the step of the
8-GPU run; max / mean is what the partition costs.

    python tools/shard_balance.py [--config cfg3] [--gene-order sorted] [--world 8] [--steps 20]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="cfg3")
    ap.add_argument("--gene-order", default="sorted", choices=["config", "sorted"])
    ap.add_argument("--gene-kind", default=None)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--exampledata", default=None, help="a Roary table instead of a config (-g); traits synthetic")
    args = ap.parse_args()
    import torch
    import bench
    from scoary_amd import dist as sdist
    from scoary_amd import synth
    from scoary_amd.engine import AssociationEngine, pack_bits_rows
    eng = AssociationEngine(0)
    if args.exampledata:
        from scoary_amd import methods as m
        path = args.exampledata
        if path.endswith(".gz"):                                  # the committed fixture (tests/golden/exampledata)
            import gzip
            import tempfile
            tmp = tempfile.NamedTemporaryFile("w", suffix=".csv", delete=False, newline="")
            with gzip.open(path, "rt", newline="") as f:
                tmp.write(f.read())
            tmp.close()
            path = tmp.name
        with open(path, "r", newline=None) as f:
            table = m.Csv_to_dic_Roary(f, ",", [], startcol=14)["Roarydic"]
        N = len(table.strains)
        base = np.unpackbits(table.rows64.view(np.uint8), axis=1, bitorder="little")[:, :N].copy()
        rng = np.random.default_rng(1)
        traits, P, seed = synth.make_traits(2, N, rng), 10_000, 7
        what = "%s (%d genes x %d isolates, file order), 2 synthetic traits" % (os.path.basename(args.exampledata),
                                                                               base.shape[0], N)
    else:
        base, traits, P, seed = synth.make_config(args.config, gene_kind=args.gene_kind)
        base = bench.order_genes(base, args.gene_order)
        what = "%s, gene order %s" % (args.config, args.gene_order)
    G, N = base.shape
    T = traits.shape[0]
    trv = eng.vecrows(pack_bits_rows((traits == 1).astype(np.uint8)), N)
    mkv = eng.vecrows(pack_bits_rows((traits != 2).astype(np.uint8)), N)
    plan = eng.trait_plan(trv, mkv, N)
    print("# %s: %d genes x %d isolates x %d traits, P = %d, split %d ways; every shard timed alone on one GPU"
          % (what, G, N, T, P, args.world))
    out = {}
    for kind in ("contiguous", "stride"):
        part = sdist.GenePartition(G, args.world, kind)
        rows = []
        for rk in range(args.world):
            genes = np.ascontiguousarray(base[part.index(rk)])
            ones = genes.sum(axis=1, dtype=np.int64)
            entries = int(np.minimum(ones, N - ones).sum())
            gm = eng.tile_rows(pack_bits_rows(genes), N)
            eng.build_lists(gm)
            ws = eng.workspace(gm, T, P, use_lists=True)
            use_graph = eng.auto_graph_eligible(gm, T, P)
            graph = eng.capture(gm, trv, mkv, P, seed, ws, use_lists=True, plan=plan)[0] if use_graph else None

            def step():
                if graph is not None:
                    graph.launch()
                else:
                    eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=True, workspace=ws, plan=plan,
                                  graph=False)
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.steps):
                step()
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1) / args.steps
            eng.set_timing(True)
            for _ in range(3):
                eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=True, workspace=ws, plan=plan, graph=False)
            torch.cuda.synchronize()
            k3 = eng.kernel_ms(eng.list_kernel_name(N))
            eng.set_timing(False)
            if graph is not None:
                graph.close()
            rows.append({"rank": rk, "genes": int(genes.shape[0]), "list_entries": entries, "step_ms": ms,
                         "k_permute_lists_ms": k3, "hip_graph": bool(use_graph)})
            del gm, ws
        steps = np.array([r["step_ms"] for r in rows])
        k3s = np.array([r["k_permute_lists_ms"] for r in rows])
        ent = np.array([r["list_entries"] for r in rows], dtype=np.float64)
        out[kind] = {"per_rank": rows, "step_ms_max": float(steps.max()), "step_ms_mean": float(steps.mean()),
                     "step_max_over_mean": float(steps.max() / steps.mean()),
                     "k3_max_over_min": float(k3s.max() / max(k3s.min(), 1e-9)),
                     "k3_max_over_mean": float(k3s.max() / k3s.mean()),
                     "entries_max_over_mean": float(ent.max() / ent.mean()),
                     "tests_per_s_at_slowest_rank": float(G) * T * P / (steps.max() * 1e-3)}
        print("\n%s partition" % kind)
        print("rank  genes   list entries   step ms   k_permute_lists ms")
        for r in rows:
            print("%4d %6d %14d %9.4f %12.4f" % (r["rank"], r["genes"], r["list_entries"], r["step_ms"],
                                                 r["k_permute_lists_ms"]))
        o = out[kind]
        print("slowest rank %.4f ms, mean %.4f ms: max / mean = %.3f (list entries %.3f); k_permute_lists max / min = %.3f"
              % (o["step_ms_max"], o["step_ms_mean"], o["step_max_over_mean"], o["entries_max_over_mean"],
                 o["k3_max_over_min"]))
        print("=> %d GPUs at the slowest rank's step: %.3e tests/s" % (args.world, o["tests_per_s_at_slowest_rank"]))
    print("\nJSON " + json.dumps(out))


if __name__ == "__main__":
    main()
