#!/usr/bin/env python3
"""The command line at the scale of one cfg5 shard (round 6): a Roary-style
table of 125 000 genes x 10 000 isolates (2.5 GB of text: "1" / "0" cells, cfg5's gene-frequency
spectrum) and 50 traits (two with 1 % missing values), then

    python -m scoary_amd -g table -t traits --no_pairwise --permute 1000 -p 1.0
    python -m scoary_amd -g table -t traits --no_pairwise --permute 1000            (default cut-off 0.05)

with the stage clock of each run ("Stage detail" log line), the bytes written and rows per file.
What is checked: no stage's cost grows with G x T Python objects -- the reader is the native one,
the result rows are (table, index array), the per-trait statistics run on worker threads and the
result files are written natively (scoary_results_write).

    python tools/e2e_cfg5_shard.py [--genes 125000 --isolates 10000 --traits 50 --permute 1000]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_table(path, G, N, rng, block=2048):
    from scoary_amd import synth
    meta = ["Gene", "Non-unique Gene name", "Annotation", "No. isolates", "No. sequences",
            "Avg sequences per isolate", "Genome Fragment", "Order within Fragment", "Accessory Fragment",
            "Accessory Order with Fragment", "QC", "Min group size nuc", "Max group size nuc", "Avg group size nuc"]
    ones = np.zeros(G, dtype=np.int64)
    with open(path, "wb") as f:
        f.write((",".join(meta + ["iso_%d" % i for i in range(N)]) + "\n").encode())
        row = np.empty((block, 2 * N), dtype=np.uint8)
        row[:, 1::2] = ord(",")
        row[:, -1] = ord("\n")
        for g0 in range(0, G, block):
            g1 = min(G, g0 + block)
            dense = synth.make_genes(g1 - g0, N, rng, kind="uniform", core_frac=0.05, block=block)
            ones[g0:g1] = dense.sum(axis=1)
            row[:g1 - g0, 0::2] = np.where(dense, ord("1"), ord("0"))
            for i in range(g1 - g0):
                f.write(("gene_%d,,synthetic,%d,1,1,1,1,1,1,1,1,1,1," % (g0 + i, ones[g0 + i])).encode())
                f.write(row[i].tobytes())
    return ones


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=125_000)
    ap.add_argument("--isolates", type=int, default=10_000)
    ap.add_argument("--traits", type=int, default=50)
    ap.add_argument("--permute", type=int, default=1000)
    ap.add_argument("--keep", action="store_true")
    a = ap.parse_args()
    from scoary_amd import synth
    rng = np.random.default_rng(20260904)
    d = tempfile.mkdtemp(prefix="scoary_cfg5_")
    G, N, T = a.genes, a.isolates, a.traits
    t0 = time.time()
    gpa = os.path.join(d, "gpa.csv")
    write_table(gpa, G, N, rng)
    traits = synth.make_traits(T, N, rng, missing_traits=(8, 9))
    tr = os.path.join(d, "traits.csv")
    with open(tr, "w") as f:
        f.write("," + ",".join("trait_%d" % t for t in range(T)) + "\n")
        sym = np.array(["0", "1", "NA"])
        for i in range(N):
            f.write("iso_%d," % i + ",".join(sym[traits[:, i]]) + "\n")
    print("wrote %d genes x %d isolates (%.2f GB) and %d traits in %.1f s"
          % (G, N, os.path.getsize(gpa) / 1e9, T, time.time() - t0))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for label, cut in (("-p 1.0", ["-p", "1.0"]), ("default cut-off (-p 0.05)", [])):
        out = os.path.join(d, "out_" + ("all" if cut else "default")) + os.sep
        os.makedirs(out)
        cmd = [sys.executable, "-m", "scoary_amd", "-g", gpa, "-t", tr, "-o", out, "--no-time", "--no_pairwise",
               "-e", str(a.permute)] + cut
        t0 = time.time()
        p = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=ROOT)
        wall = time.time() - t0
        files = [f for f in os.listdir(out) if f.endswith(".results.csv")]
        nbytes = sum(os.path.getsize(os.path.join(out, f)) for f in files)
        rows = 0
        for fn in files:
            with open(os.path.join(out, fn), "rb") as fh:
                rows += sum(chunk.count(b"\n") for chunk in iter(lambda: fh.read(1 << 24), b"")) - 1
        print("\n== %s: exit %d, wall %.2f s; %d result files, %d rows, %.1f MB"
              % (label, p.returncode, wall, len(files), rows, nbytes / 1e6))
        if p.returncode != 0:
            print(p.stdout[-3000:], p.stderr[-3000:])
        for line in (p.stdout + p.stderr).splitlines():
            if line.startswith(("Stage seconds", "Stage detail", "Checked a total")):
                print(line)
        first = sorted(files)[0] if files else None
        if first:
            with open(os.path.join(out, first)) as fh:
                print(first, "| " + fh.readline().strip()[:160])
                print(first, "| " + fh.readline().strip()[:200])
    if not a.keep:
        import shutil
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
