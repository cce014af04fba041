#!/usr/bin/env python3
"""All cases of tests/golden/fuzz_corpus.json.gz through the command line on this GPU, every difference from the
reference's record listed (the pytest module stops at the first under -x).   python tools/fuzz_report.py [out.txt]"""
import io
import os
import sys
import tempfile
import traceback
import contextlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as tf  # noqa: E402


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
    bad = 0
    for case in tf.CORPUS["cases"]:
        tmp = tempfile.mkdtemp(prefix="fuzzrun_")
        sink = io.StringIO()
        try:
            with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
                got = tf.run_case(case, tmp)
            diffs = tf.compare_case(case, got)
            if diffs and got["status"] == "ok":
                for fn, text in got["files"].items():
                    if any(d.startswith(fn) for d in diffs):
                        diffs.append("---- our %s ----\n%s" % (fn, text))
        except BaseException:
            diffs = ["our command line raised: " + traceback.format_exc().strip().splitlines()[-1]
                     + " @ " + " | ".join(l.strip() for l in traceback.format_exc().strip().splitlines()[-7:-1])]
        if diffs:
            bad += 1
            print("case %3d  N=%-3d G=%-3d T=%d roary=%-5s ref=%s  argv: %s" % (
                case["id"], case["N"], case["G"], case["T"], case["roary"], case["ref"]["status"],
                " ".join(case["argv"])), file=out)
            for d in diffs:
                print("      " + d[:6000], file=out)
    print("%d of %d cases differ from the reference" % (bad, len(tf.CORPUS["cases"])), file=out)
    # inputs on which the REFERENCE raises something other than SystemExit (kept by FUZZ_KEEP_CRASHES=1 corpora): there is
    # no behaviour to match; listed is what this command line does with them (it must end, one way or the other)
    tally = {}
    for case in tf.CORPUS.get("crash_cases", []):
        tmp = tempfile.mkdtemp(prefix="fuzzrun_")
        sink = io.StringIO()
        try:
            with contextlib.redirect_stdout(sink), contextlib.redirect_stderr(sink):
                got = tf.run_case(case, tmp)
            what = "completes" if got["status"] == "ok" else "exits: " + str(got["message"])[:70]
        except BaseException as e:
            what = "raises " + type(e).__name__ + ": " + str(e)[:60]
            where = " | ".join(l.strip() for l in traceback.format_exc().strip().splitlines()[-9:-1] if "File" in l)
            print("crash case %d (%s): %s @ %s" % (case["id"], " ".join(case["argv"]), what, where), file=out)
        key = (case["ref"]["message"][:60], what)
        tally[key] = tally.get(key, 0) + 1
    for (ref_msg, what), n in sorted(tally.items(), key=lambda kv: -kv[1]):
        print("reference crashes with %-62s x %3d -> ours %s" % (ref_msg, n, what), file=out)
    out.flush()


if __name__ == "__main__":
    main()
