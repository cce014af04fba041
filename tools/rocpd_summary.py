#!/usr/bin/env python3
"""Turn the rocprofv3 (ROCm 7.2 rocpd sqlite) outputs of tools/profile.sh into
the text / JSON summaries committed under profiles/.

    python tools/rocpd_summary.py gpurun_out/prof_r01 profiles/r01

writes  <prefix>_kernel_stats.txt   (the --kernel-trace --stats table)
        <prefix>_pmc.json           (per-kernel average PMC values per launch)
HBM traffic convention (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads 1/2 of the bytes of a wide
coalesced stream, so traffic_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import json
import os
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0]


def main():
    src, prefix = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(prefix) or ".", exist_ok=True)
    stats_db = os.path.join(src, "stats", "stats_results.db")
    if os.path.exists(stats_db):
        c = sqlite3.connect(stats_db)
        rows = c.execute("select name, total_calls, total_duration, average, percentage "
                         "from top_kernels").fetchall()
        with open(prefix + "_kernel_stats.txt", "w") as f:
            wl = sys.argv[3] if len(sys.argv) > 3 else "cfg3"
            cmd = "python tools/bench_tree.py" if "tree" in wl else (
                "python bench.py --no-cpu-baseline%s   (defaults: 5 warm-up + 20 timed steps)"
                % ("" if wl == "cfg3" else " --config " + wl))
            f.write("# rocprofv3 --kernel-trace --stats -- %s   (durations in us)\n" % cmd)
            f.write("%-34s %6s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
            for name, calls, tot, avg, pct in rows:
                f.write("%-34s %6d %14.3f %12.3f %8.3f\n" % (short(name)[:34], calls, tot, avg, pct))
            # per-launch durations of the dominant kernel, in launch order (the first
            # launches of a process run on unsettled clocks)
            if rows:
                top = rows[0][0]
                durs = [d / 1e3 for (d,) in c.execute(
                    "select duration from kernels where name = ? order by start", (top,))]
                shown = durs[:40]
                f.write("# %s per launch, us%s: %s\n" % (short(top)[:34],
                                                        "" if len(durs) <= 40 else " (first 40 of %d)" % len(durs),
                                                        " ".join("%.1f" % d for d in shown)))
        c.close()
    pmc = {}
    for sub, db in (("pmc_fetch", "fetch_results.db"), ("pmc_write", "write_results.db"),
                    ("pmc_sq", "sq_results.db"), ("pmc_lds", "lds_results.db")):
        path = os.path.join(src, sub, db)
        if not os.path.exists(path):
            continue
        c = sqlite3.connect(path)
        for name, counter, n, avg, dur in c.execute(
                "select kernel_name, counter_name, count(*), avg(value), avg(duration) "
                "from counters_collection group by kernel_name, counter_name"):
            k = pmc.setdefault(short(name), {})
            k[counter] = avg
            k.setdefault("_launches", {})[counter] = n
            k.setdefault("_avg_duration_ns", {})[counter] = dur
        c.close()
    for k, v in pmc.items():
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            v["hbm_traffic_bytes_per_launch"] = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            v["hbm_traffic_bytes_per_launch_raw"] = (v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024
            v["WRITE_SIZE_bytes"] = v["WRITE_SIZE"] * 1024
            v["FETCH_SIZE_bytes_corrected"] = 2 * v["FETCH_SIZE"] * 1024
    sha = None
    sha_path = os.path.join(src, "kernel_source_sha256.txt")
    if os.path.exists(sha_path):
        sha = open(sha_path).read().strip()
    pmc["_meta"] = {"workload": sys.argv[3] if len(sys.argv) > 3 else "cfg3",
                    # sha256 over scoary_amd/csrc/{bench.KERNEL_SOURCES} at profile time
                    "kernel_source_sha256": sha,
                    "source": "tools/profile.sh -> rocprofv3 --pmc (separate passes)",
                    "traffic_formula": "(2*FETCH_SIZE + WRITE_SIZE)*1024 bytes per launch"}
    with open(prefix + "_pmc.json", "w") as f:
        json.dump(pmc, f, indent=1, sort_keys=True)
    print("wrote", prefix + "_kernel_stats.txt", prefix + "_pmc.json")


if __name__ == "__main__":
    main()
