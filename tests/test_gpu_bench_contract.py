"""bench.py's output contract on a small shape: one JSON line with the fields the round driver
and the judge read (metric / value / unit / n_gpus / steps / warmup / ms_per_step /
higher_is_better / scaling / vs_baseline / dtype / data / config.workload, roofline{bound,
achieved, peak, unit, frac, traffic}, cpu_baseline{value, unit, cores, kind, sample}) and
internally consistent numbers; eager and hipGraph replay."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--genes", "3000",
                          "--permutations", "1024", "--steps", "3", "--warmup", "1"] + extra,
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # exactly ONE JSON line
    return json.loads(lines[0])


@pytest.mark.parametrize("extra", [["--cpu-seconds", "0.5"], ["--graph", "--no-cpu-baseline"],
                                   ["--kernel", "dense", "--no-cpu-baseline"]])
def test_bench_json_contract(extra):
    d = _run(extra)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "kernel_ms", "setup_ms"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["scaling"] == "weak"
    assert "workload" in d["config"] and "model" not in d["config"]
    G, T, P = d["config"]["genes_per_gpu"], d["config"]["traits"], d["config"]["permutations"]
    assert (G, P) == (3000, 1024)
    assert abs(d["value"] - G * T * P / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    assert d["value_incl_setup"] < d["value"] and d["ms_per_step_median"] > 0
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms"):
        assert k in r, k
    assert r["bound"] == "valu" and r["peak"] > 70
    # overridden shape: no committed PMC profile applies -> counters refused, not invented
    assert r["frac"] is None and r["achieved"] is None and "overridden" in r["counters_refused"]
    k3 = "k_permute" if "dense" in extra else "k_permute_lists"
    # the useful part of the issue (full-adder lane-ops from the list plan) needs no counters
    if "dense" in extra:
        assert r["useful_lane_ops"] is None and r["useful_frac"] is None
    else:
        assert r["useful_lane_ops"] > 0 and 0 < r["useful_frac_unpadded"] <= r["useful_frac"] < 1
        assert r["minority_entries_per_gene"] <= r["padded_entries_per_gene"] <= 1000 + 16
        # 2 * 31/32 lane-ops per padded entry and 32 permutations
        want = r["padded_entries_per_gene"] * 2 * 31 / 32 / 32
        assert abs(r["adder_ops_per_test"] - want) < 0.02 * want
        assert r["overhead_ops_per_test"] is None and r["measured_op_peak"] == 56.1
    assert r["kernel"] == k3 and d["kernel_ms"][k3] > 0
    assert sum(d["kernel_ms"].values()) < 3 * d["ms_per_step"]
    assert d["config"]["hip_graph"] == ("--graph" in extra)
    if "--no-cpu-baseline" not in extra:
        for name in ("cpu_baseline", "cpu_baseline_port"):
            c = d[name]
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in c, (name, k)
        assert d["cpu_baseline"]["kind"] == "scipy-restatement" and d["cpu_baseline"]["matches_gpu"] is True
        assert d["cpu_baseline_port"]["kind"] == "port"
        assert d["cpu_baseline"]["value"] < d["cpu_baseline_port"]["value"] < d["value"]
