#!/usr/bin/env python3
"""Kernel timeline of a few consecutive steps from a rocprofv3 --kernel-trace run (rocpd sqlite): every
kernel between the `--first`-th and the `--last`-th launch of the anchor kernel, with its start relative to
the window, its duration, the gap since the previous kernel ENDED (any stream) and its stream / queue.
Round 6: where the 0.03 ms per step go that a gene-sharded rank's graph replay has over the single process.

    python tools/step_timeline.py <dir with *_results.db> [--anchor k_permute_lists --first 12 --last 14]
"""
import argparse
import glob
import sqlite3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--anchor", default="k_permute_lists")
    ap.add_argument("--first", type=int, default=12)
    ap.add_argument("--last", type=int, default=14)
    a = ap.parse_args()
    for db in sorted(glob.glob(a.dir + "/**/*.db", recursive=True)):
        con = sqlite3.connect(db)
        cols = [r[1] for r in con.execute("pragma table_info(kernels)")]
        want = [c for c in ("name", "start", "end", "duration", "stream_id", "queue_id", "stream", "queue") if c in cols]
        rows = con.execute("select %s from kernels order by start" % ", ".join(want)).fetchall()
        if not rows:
            continue
        idx = [i for i, r in enumerate(rows) if a.anchor in r[0]]
        if len(idx) <= a.last:
            print(db, "only", len(idx), "launches of", a.anchor)
            continue
        lo, hi = idx[a.first], idx[a.last]
        t0 = rows[lo][1]
        print("# %s: kernels from launch %d to launch %d of %s (us)" % (db.split("/")[-1], a.first, a.last, a.anchor))
        print("%10s %10s %9s  %-10s %s" % ("start", "duration", "gap", "stream", "kernel"))
        prev_end = None
        for r in rows[max(lo - 6, 0):hi + 1]:
            d = dict(zip(want, r))
            end = d.get("end", d["start"] + d["duration"])
            gap = "" if prev_end is None else "%9.2f" % ((d["start"] - prev_end) / 1e3)
            name = d["name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60]
            sid = d.get("stream_id", d.get("stream", d.get("queue_id", d.get("queue", ""))))
            print("%10.2f %10.2f %9s  %-10s %s" % ((d["start"] - t0) / 1e3, (end - d["start"]) / 1e3, gap, sid, name))
            prev_end = end if prev_end is None else max(prev_end, end)
        steps = [(rows[idx[i + 1]][1] - rows[idx[i]][1]) / 1e3 for i in range(a.first, a.last)]
        print("# anchor-to-anchor period, us:", " ".join("%.1f" % s for s in steps))


if __name__ == "__main__":
    main()
