#!/usr/bin/env python3
"""End-to-end run of the command line on a synthetic Roary-style table
(write CSVs -> python -m scoary_amd ...), with wall-clock per stage.  A sanity
and scalability check of the host side, not a benchmark of the kernels.

    python tools/e2e_synth.py [--genes 20000] [--isolates 1000] [--traits 5]
                              [--permute 1000] [--pairwise]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=20000)
    ap.add_argument("--isolates", type=int, default=1000)
    ap.add_argument("--traits", type=int, default=5)
    ap.add_argument("--permute", type=int, default=1000)
    ap.add_argument("--pairwise", action="store_true")
    ap.add_argument("--profile", action="store_true", help="run the CLI under cProfile")
    a = ap.parse_args()
    rng = np.random.default_rng(3)
    d = tempfile.mkdtemp()
    G, N, T = a.genes, a.isolates, a.traits
    t0 = time.time()
    dense = rng.random((G, N)) < rng.beta(0.4, 0.4, (G, 1))        # U-shaped gene frequencies
    causal = dense[7] ^ (rng.random(N) < 0.05)
    iso = ["iso_%d" % i for i in range(N)]
    meta = ["Gene", "Non-unique Gene name", "Annotation", "No. isolates", "No. sequences",
            "Avg sequences per isolate", "Genome Fragment", "Order within Fragment",
            "Accessory Fragment", "Accessory Order with Fragment", "QC", "Min group size nuc",
            "Max group size nuc", "Avg group size nuc"]
    gpa = os.path.join(d, "gpa.csv")
    with open(gpa, "w") as f:
        f.write(",".join(meta + iso) + "\n")
        cells = np.where(dense, "x", "")
        for g in range(G):
            f.write("gene_%d,,synthetic," % g + ",".join(["1"] * 11) + "," + ",".join(cells[g]) + "\n")
    tr = os.path.join(d, "traits.csv")
    with open(tr, "w") as f:
        f.write("," + ",".join("trait_%d" % t for t in range(T)) + "\n")
        lab = rng.random((T, N)) < 0.4
        lab[0] = causal
        for i in range(N):
            f.write(iso[i] + "," + ",".join("1" if lab[t, i] else "0" for t in range(T)) + "\n")
    print("wrote inputs (%.0f MB) in %.1f s" % (os.path.getsize(gpa) / 1e6, time.time() - t0))
    out = os.path.join(d, "out")
    prof = os.path.join(d, "cli.prof")
    cmd = [sys.executable] + (["-m", "cProfile", "-o", prof] if a.profile else []) + [
        "-m", "scoary_amd", "-g", gpa, "-t", tr, "-o", out, "--no-time",
           "-e", str(a.permute), "-p", "0.05"]
    if not a.pairwise:
        cmd.append("--no_pairwise")
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True)
    dt = time.time() - t0
    print("exit", r.returncode, "wall %.1f s" % dt)
    print("\n".join(r.stdout.splitlines()[-12:]))
    if r.returncode:
        print(r.stderr[-2000:])
    if a.profile and os.path.exists(prof):
        import pstats
        pstats.Stats(prof).sort_stats("cumulative").print_stats(35)
    for fn in sorted(os.listdir(out)):
        p = os.path.join(out, fn)
        print(fn, os.path.getsize(p), "bytes")
    with open(os.path.join(out, "trait_0.results.csv")) as f:
        print(f.readline().strip()[:200])
        print(f.readline().strip()[:300])


if __name__ == "__main__":
    main()
