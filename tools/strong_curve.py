#!/usr/bin/env python3
"""The strong-scaling curve that can be measured on ONE GPU (round 6): for N = 1, 2, 4, 8 the config's genes are
split into the stride domains of an N-rank run (dist.GenePartition), every rank's shard is set up as bench.py
sets it up and its step is timed ALONE (hipGraph replay where bench.py would replay), and the N-GPU step is
the slowest shard's.  What this leaves out is the exchange: +0.02-0.04 ms per step on a launch-bound shard
(profiles/r06_bench_cfg4_rank_of_8_sharded.json against ..._single_process.json), hidden under the next step's
kernels on the larger ones.  No multi-GPU box reaches the build: this is the prediction the first SCALE run
of the driver can be held against (bench.py --gpus N prints the same quantities as `scaling_strong`).

    python tools/strong_curve.py [--configs cfg3 cfg4] [--worlds 1 2 4 8] [--gene-order config|sorted]
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
    ap.add_argument("--configs", nargs="+", default=["cfg3", "cfg4"])
    ap.add_argument("--worlds", nargs="+", type=int, default=[1, 2, 4, 8])
    ap.add_argument("--gene-order", default="config", choices=["config", "sorted"])
    ap.add_argument("--steps", type=int, default=20)
    args = ap.parse_args()
    import torch
    import bench
    from scoary_amd import dist as sdist
    from scoary_amd import synth
    from scoary_amd.engine import AssociationEngine, pack_bits_rows
    eng = AssociationEngine(0)
    out = {}
    for cfg in args.configs:
        base, traits, P, seed = synth.make_config(cfg)
        base = bench.order_genes(base, args.gene_order)
        G, N = base.shape
        T = traits.shape[0]
        trv = eng.vecrows(pack_bits_rows((traits == 1).astype(np.uint8)), N)
        mkv = eng.vecrows(pack_bits_rows((traits != 2).astype(np.uint8)), N)
        plan = eng.trait_plan(trv, mkv, N)
        print("# %s (gene order %s): %d genes x %d isolates x %d traits, P = %d; stride shards, each timed alone"
              % (cfg, args.gene_order, G, N, T, P))
        print("%5s %12s %12s %10s %14s %9s %6s" % ("GPUs", "slowest ms", "fastest ms", "max/mean", "tests/s", "speed-up", "graph"))
        rows, t1 = [], None
        for world in args.worlds:
            part = sdist.GenePartition(G, world, "stride")
            ms = []
            graphed = False
            for rk in range(world):
                genes = np.ascontiguousarray(base[part.index(rk)])
                gm = eng.tile_rows(pack_bits_rows(genes), N)
                eng.build_lists(gm)
                ws = eng.workspace(gm, T, P, use_lists=True)
                use_graph = eng.auto_graph_eligible(gm, T, P)
                graphed = graphed or use_graph
                graph = eng.capture(gm, trv, mkv, P, seed, ws, use_lists=True, plan=plan)[0] if use_graph else None

                def step():
                    if graph is not None:
                        graph.launch()
                    else:
                        eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=True, workspace=ws,
                                      plan=plan, graph=False)
                for _ in range(5):
                    step()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.steps):
                    step()
                e1.record()
                e1.synchronize()
                ms.append(e0.elapsed_time(e1) / args.steps)
                if graph is not None:
                    graph.close()
                del gm, ws
            ms = np.array(ms)
            rate = float(G) * T * P / (ms.max() * 1e-3)
            if t1 is None:
                t1 = ms.max()
            rows.append({"gpus": world, "slowest_ms": float(ms.max()), "fastest_ms": float(ms.min()),
                         "max_over_mean": float(ms.max() / ms.mean()), "tests_per_s": rate,
                         "speedup": float(t1 / ms.max()), "hip_graph": bool(graphed)})
            print("%5d %12.4f %12.4f %10.3f %14.4e %9.2f %6s" % (world, ms.max(), ms.min(), ms.max() / ms.mean(), rate,
                                                                 t1 / ms.max(), graphed))
        out[cfg] = rows
    print("JSON " + json.dumps(out))


if __name__ == "__main__":
    main()
