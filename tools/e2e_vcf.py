#!/usr/bin/env python3
"""BASELINE config 4 as it is literally defined -- "VCF-derived 200k variants x 5000 isolates x 1
trait, --permute 10000" -- end to end through the command line, with seconds and bytes per second
of every stage:

    synthesize a haploid VCF (rare variants, a share of multi-allelic sites, a few missing calls)
    python -m scoary_amd.vcf2scoary  VCF -> presence/absence table      (scoary/vcf2scoary.py:50-218)
    python -m scoary_amd -g table -t traits -s 11 --no_pairwise --permute P
                                                                         (scoary/methods.py:437-460: the
                                                                          non-Roary branch, identifiers
                                                                          CHROM_|_POS_|_ID)

    python tools/e2e_vcf.py [--sites 200000] [--samples 5000] [--permute 10000] [--keep DIR]

A check of the host side around the kernels, not a kernel benchmark.
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def write_vcf(path, sites, samples, rng, multi_frac=0.03, missing_frac=0.0005, block=2000):
    """Haploid VCF 4.1: GT only, minor-allele frequency ~ Beta(0.3, 3) (the `rare` spectrum of
    synth.make_config("cfg4")), multi_frac of the sites with two ALT alleles, a few "." calls.
    Returns the number of table rows the converter has to produce (one per ALT allele)."""
    names = ["S%04d" % i for i in range(samples)]
    rows = 0
    with open(path, "wb") as f:
        f.write(b"##fileformat=VCFv4.1\n"
                b"##FORMAT=<ID=GT,Number=1,Type=String,Description=\"Genotype\">\n"
                b"##INFO=<ID=TYPE,Number=A,Type=String,Description=\"The type of allele.\">\n")
        f.write(("#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n").encode())
        pos = 0
        for s0 in range(0, sites, block):
            nb = min(block, sites - s0)
            maf = rng.beta(0.3, 3.0, size=(nb, 1)).astype(np.float32)
            gt = (rng.random((nb, samples), dtype=np.float32) < maf).astype(np.uint8)
            multi = rng.random(nb) < multi_frac
            second = (rng.random((nb, samples), dtype=np.float32) < 0.3) & (gt == 1) & multi[:, None]
            gt[second] = 2
            cells = np.empty((nb, samples, 2), dtype=np.uint8)
            cells[:, :, 0] = ord("\t")
            cells[:, :, 1] = gt + ord("0")
            if missing_frac > 0:
                cells[:, :, 1][rng.random((nb, samples), dtype=np.float32) < missing_frac] = ord(".")
            for k in range(nb):
                pos += int(rng.integers(1, 40))
                alt = b"C,A" if multi[k] else b"C"
                f.write(b"NC_000962\t%d\t%d\tT\t%s\t9999\t0\tTYPE=snp\tGT" % (pos, s0 + k, alt))
                f.write(cells[k].tobytes())
                f.write(b"\n")
                rows += 2 if multi[k] else 1
    return rows, names


def run(cmd, **kw):
    t0 = time.time()
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, **kw)
    return r, time.time() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sites", type=int, default=200_000)
    ap.add_argument("--samples", type=int, default=5_000)
    ap.add_argument("--permute", type=int, default=10_000)
    ap.add_argument("--keep", default=None, help="directory to work in and keep (default: a temp dir, removed)")
    ap.add_argument("--profile", action="store_true", help="run the command line under cProfile")
    a = ap.parse_args()
    d = a.keep or tempfile.mkdtemp(prefix="scoary_vcf_")
    os.makedirs(d, exist_ok=True)
    rng = np.random.default_rng(20260903)
    vcf, table, traits, out = (os.path.join(d, n) for n in ("cfg4.vcf", "mutations.csv", "traits.csv", "out"))
    t0 = time.time()
    rows, names = write_vcf(vcf, a.sites, a.samples, rng)
    lab = rng.random(a.samples) < 0.3
    with open(traits, "w") as f:
        f.write(",resistance\n")
        for s, v in zip(names, lab):
            f.write("%s,%d\n" % (s, v))
    print("synthesized %d sites x %d samples: %.0f MB VCF, %d table rows expected, %.1f s"
          % (a.sites, a.samples, os.path.getsize(vcf) / 1e6, rows, time.time() - t0))

    r, dt = run([sys.executable, "-m", "scoary_amd.vcf2scoary", "--force", "--out", table, vcf])
    if r.returncode:
        print(r.stdout[-2000:], r.stderr[-2000:])
        raise SystemExit("vcf2scoary failed")
    vb, tb = os.path.getsize(vcf), os.path.getsize(table)
    print("stage convert   %7.2f s   VCF %.0f MB in at %.0f MB/s, table %.0f MB out at %.0f MB/s"
          % (dt, vb / 1e6, vb / 1e6 / dt, tb / 1e6, tb / 1e6 / dt))

    prof = os.path.join(d, "cli.prof")
    cmd = [sys.executable] + (["-m", "cProfile", "-o", prof] if a.profile else []) + [
        "-m", "scoary_amd", "-g", table, "-t", traits, "-s", "11", "-o", out, "--no-time", "--no_pairwise",
        "-e", str(a.permute), "-p", "0.05"]
    r, dt = run(cmd)
    print("command line exit %d, wall %.2f s (interpreter start, torch import and the first HIP module load included)"
          % (r.returncode, dt))
    if r.returncode:
        print(r.stdout[-3000:], r.stderr[-3000:])
        raise SystemExit("command line failed")
    for line in r.stdout.splitlines():
        if line.startswith(("Stage seconds", "Stage detail", "stage ")) or "genes for associations" in line:
            print(line)
    res = os.path.join(out, "resistance.results.csv")
    with open(res) as f:
        lines = f.readlines()
    print("result rows written: %d (of %d variants); %d bytes" % (len(lines) - 1, rows, os.path.getsize(res)))
    print(lines[0].strip()[:160])
    print(lines[1].strip()[:260])
    if a.profile and os.path.exists(prof):
        import pstats
        pstats.Stats(prof).sort_stats("cumulative").print_stats(30)
    if not a.keep:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
