#!/usr/bin/env python3
"""First differing lines between our pairwise-stage CSVs on the reference's example data and the reference's files."""
import gzip, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
tmp = tempfile.mkdtemp()
for fn in ("Gene_presence_absence.csv", "Tetracycline_resistance.csv"):
    open(os.path.join(tmp, fn), "w", newline="").write(gzip.open(os.path.join(G, "exampledata", fn + ".gz"), "rt", newline="").read())
from scoary_amd import methods as m
for sub, extra in (("csv_pairwise_default", ["-u"]), ("csv_pairwise_bh_pw", ["-c", "BH", "PW", "-p", "0.9", "0.05", "-m", "300"])):
    od = os.path.join(tmp, sub)
    sys.argv = ["scoary", "-g", os.path.join(tmp, "Gene_presence_absence.csv"), "-t", os.path.join(tmp, "Tetracycline_resistance.csv")] + extra + ["-o", od, "--no-time"]
    try:
        m.main()
    except SystemExit:
        pass
    for tr in ("Tetracycline_resistance", "Bogus_trait"):
        got = open(os.path.join(od, tr + ".results.csv"), newline="").read().split("\n")
        want = gzip.open(os.path.join(G, sub, tr + ".results.csv.gz"), "rt", newline="").read().split("\n")
        n = 0
        for i, (a, b) in enumerate(zip(got, want)):
            if a != b:
                n += 1
                if n <= 4:
                    print("DIFF", sub, tr, "line", i, file=sys.stderr)
                    print("  ours:", a[:60], "...", a[-150:], file=sys.stderr)
                    print("  ref :", b[:60], "...", b[-150:], file=sys.stderr)
        print("DIFF", sub, tr, "differing lines:", n, "of", len(want), len(got), file=sys.stderr)
