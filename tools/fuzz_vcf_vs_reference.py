#!/usr/bin/env python3
"""vcf2scoary against the REAL reference's script (build container only: imports /root/reference/scoary/vcf2scoary.py),
random and damaged VCFs under random flags.  No GPU.     python tools/fuzz_vcf_vs_reference.py [cases] [seed]

Held per case: the same way out (completes / the same exit message or status / an exception on both sides) and, when a
table was written, the same bytes.  Where the reference raises something else than SystemExit there is nothing to be
equal to: tallied, with what this build does."""
import contextlib
import io
import os
import shutil
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference")
import scoary.vcf2scoary as ref  # noqa: E402
from scoary_amd import vcf2scoary as ours  # noqa: E402


def make_vcf(rng):
    S = int(rng.integers(0, 7))
    V = int(rng.integers(0, 14))
    nl = "\r\n" if rng.random() < 0.25 else "\n"
    meta = []
    r = rng.random()
    if r < 0.8:
        meta.append("##fileformat=VCFv4.%d" % rng.integers(0, 4))
    elif r < 0.9:
        meta.append("##fileformat=VCFv3.3")
    meta.append('##FORMAT=<ID=GT,Number=%s,Type=String,Description="Genotype">' % ("1" if rng.random() < 0.9 else "2"))
    if rng.random() < 0.04:
        meta[-1] = '##FORMAT=<ID=DP,Number=1,Type=Integer,Description="Depth">'        # no GT entry
    if rng.random() < 0.7:
        meta.append('##INFO=<ID=TYPE,Number=A,Type=String,Description="type, of ""allele"", quoted">')
    if rng.random() < 0.4:
        meta.append("##contig=<ID=chr1,length=1000>")
    if rng.random() < 0.3:
        meta.append("##source=freebayes v1.3")
    if rng.random() < 0.1:
        meta.append("##odd line without equals")
    rng.shuffle(meta)
    lines = list(meta)
    if rng.random() < 0.95:
        lines.append("\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"][:9 if rng.random() < 0.95 else 8]
                               + ["s%d" % i if rng.random() < 0.9 else 'sample "%d"' % i for i in range(S)]))
    for v in range(V):
        nalt = int(rng.integers(1, 4))
        alt = ",".join("ACGT"[a] for a in range(nalt))
        if rng.random() < 0.08:
            alt = "A,,T"[:2 * nalt - 1]
        g = rng.integers(0, nalt + 1, S).astype(object)
        for i in range(S):
            u = rng.random()
            if u < 0.1:
                g[i] = "."
            elif u < 0.13:
                g[i] = "0/1"
            elif u < 0.15:
                g[i] = ""
            else:
                g[i] = str(g[i])
        sub = rng.random() < 0.5
        cells = [x + ":12:0.5" if sub else x for x in g]
        info = str(rng.choice(["DP=5;TYPE=snp", "TYPE=ins;AF=0.5", "TYPE=del", "TYPE=snp;DP=1", "TYPE=mnp,snp", "TYPE=complex", "TYPE=complex", "DP=1" if rng.random() < 0.1 else "TYPE=snp"]))
        row = ["chr%d" % rng.integers(1, 3), str(100 + v), ".", "G", alt, "50", "PASS", info, "GT:DP:AF" if sub else "GT"] + cells
        if rng.random() < 0.01:
            row = row[:int(rng.integers(1, len(row) + 1))]                            # a short record
        if rng.random() < 0.05 and len(row) > 2:
            row[2] = '"quoted id"'
        lines.append("\t".join(row))
    if rng.random() < 0.1:
        lines.insert(int(rng.integers(0, len(lines) + 1)), "")
    text = nl.join(lines) + (nl if rng.random() < 0.9 else "")
    return text


def run(mod, argv, cwd):
    """-> (kind, detail): ("ok", None) | ("exit", message or status) | ("raise", exception name)"""
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = ["vcf2scoary"] + argv
    os.chdir(cwd)
    sink = io.StringIO()
    try:
        with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
            try:
                if mod is ours:
                    mod.main(argv)
                else:
                    mod.main()
            except SystemExit as e:
                if e.code in (0, None):
                    return "ok", None
                return "exit", str(e.code)
            except BaseException as e:
                return "raise", type(e).__name__
        return "ok", None
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    same = 0
    bad, crash_tally = [], {}
    for k in range(cases):
        text = make_vcf(rng)
        flags = []
        u = rng.random()
        if u < 0.35:
            flags += ["--types", str(rng.choice(["snp", "ins,del", "snp,mnp,complex", "nothing", ""]))]
            # (not "ALL" spelled out: the reference's `args.types is not "ALL"` then filters EVERYTHING out -- DESIGN section 7)
        exists = rng.random() < 0.15
        if exists and rng.random() < 0.5:
            flags += ["--force"]
        missing_input = rng.random() < 0.03
        outs = []
        for mod in (ref, ours):
            tmp = tempfile.mkdtemp(prefix="vcffz_")
            try:
                src = os.path.join(tmp, "in.vcf")
                if not missing_input:
                    with open(src, "w", newline="") as f:
                        f.write(text)
                out = os.path.join(tmp, "out.csv")
                if exists:
                    with open(out, "w") as f:
                        f.write("old\n")
                default_out = rng.random() < 0.0                       # (the default path is cwd-relative: same code)
                argv = flags + ([] if default_out else ["--out", out]) + [src]
                kind, detail = run(mod, argv, tmp)
                data = None
                if os.path.exists(out):
                    with open(out, "rb") as f:
                        data = f.read()
                outs.append((kind, detail, data))
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        (rk, rd, rdata), (ok_, od, odata) = outs
        if rk == "raise":
            key = (rd, ok_ + (": " + str(od) if od else ""))
            crash_tally[key] = crash_tally.get(key, 0) + 1
            continue
        # messages carry the temporary path of the input: compare them with the directory cut out
        norm = lambda s: None if s is None else s.replace("\\", "/").split("/vcffz_")[0] + s[s.rfind("/"):] if "/vcffz_" in s else s
        if (rk, norm(rd)) != (ok_, norm(od)) or rdata != odata:
            bad.append((k, flags, (rk, rd), (ok_, od), None if rdata == odata else (rdata or b"")[:300], (odata or b"")[:300], text[:500]))
        else:
            same += 1
    print("%d cases (seed %d): %d end as the reference's script ends (same exit, same bytes), %d differ, %d where the "
          "reference itself raises" % (cases, seed, same, len(bad), sum(crash_tally.values())))
    for (r, o), n in sorted(crash_tally.items(), key=lambda kv: -kv[1]):
        print("  reference raises %-22s x %4d -> ours %s" % (r, n, o))
    for b in bad[:10]:
        print("  case %d flags %s: reference %s, ours %s" % b[:4])
        if b[4] is not None:
            print("     reference bytes %r\n     our bytes       %r" % (b[4], b[5]))
        print("     vcf: %r" % b[6])
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
