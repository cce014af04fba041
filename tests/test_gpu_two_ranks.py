"""Two real ranks with real kernels in the driver-run evidence.

No multi-GPU box is available to the build or to `pytest -m gpu`, so the two ranks share the
one device (`--share-gpu`) and exchange over gloo; everything else is the multi-GPU path as
it runs on 8 GPUs: bench.py's own launcher (re-exec under torch.distributed.run), one
process per rank, gene shards (weak: every rank its own shard; strong: shard_bounds of the
config's genes), labels regenerated from the seed on every rank, scoary_pack_records, the
asynchronous gather of 10-word records overlapped with the next step, and rank 0's checks.
`--verify-gather`: rank 0 recomputes every rank's shard alone and compares the records it
received bit for bit.  Reference analogue: the stride domains and the result weave of
scoary/methods.py:1076-1097, :1115-1122.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(extra, ranks=2):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--share-gpu",
                          "--backend", "gloo", "--config", "cfg2", "--steps", "4", "--warmup", "1",
                          "--verify-gather"] + extra,
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                  # rank 0 prints, and only rank 0
    return json.loads(lines[0])


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks_share_the_gpu(scaling):
    d = _bench(["--scaling", scaling])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == scaling
    assert d["gather_matches_single_rank"] is True
    assert "gather" in d["config"]["exchange"] and d["config"]["parallelism"] == "gene-shard x2"
    G_total = 20_000 if scaling == "weak" else 10_000
    assert d["config"]["genes_total"] == G_total
    assert d["config"]["genes_per_gpu"] == (10_000 if scaling == "weak" else 5_000)
    assert abs(d["value"] - G_total * 1 * 1000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    # one line per rank: its kernels, the bytes it sends and what the exchange cost it
    pr = d["per_rank"]
    assert [r["rank"] for r in pr] == [0, 1]
    for r in pr:
        assert r["genes"] == d["config"]["genes_per_gpu"]
        assert r["exchange_bytes"] == 1 * r["genes"] * 40       # T x genes x 10 int32 words
        assert r["kernel_ms"]["k_permute_lists"] > 0 and r["kernel_ms"]["k_counts"] > 0
        assert r["exchange_exposed_ms"] is not None and 0 <= r["exchange_exposed_ms"] < 5000
        assert r["ms_per_step"] <= d["ms_per_step"] * (1 + 1e-9)      # value uses the MAX over ranks


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_bench_three_ranks_uneven_strong_shards():
    """Three ranks, strong scaling: 10 000 genes do not divide by three (3334 / 3333 / 3333), so
    the gather pads the shorter shards to the longest and rank 0 trims them again; every
    record still equals the single-rank run's."""
    d = _bench(["--scaling", "strong"], ranks=3)
    assert d["n_gpus"] == 3 and d["rccl_ranks"] == 3 and d["gather_matches_single_rank"] is True
    assert d["config"]["genes_total"] == 10_000
    assert [r["genes"] for r in d["per_rank"]] == [3334, 3333, 3333]
    assert [r["exchange_bytes"] for r in d["per_rank"]] == [3334 * 40, 3333 * 40, 3333 * 40]


@pytest.mark.parametrize("extra", [["--no_pairwise", "-e", "200", "--seed", "7"], ["--collapse", "-c", "I", "-p", "0.05"]],
                         ids=["fisher-permutations", "pairwise-collapse"])
def test_command_line_two_ranks_share_the_gpu(exampledir, tmp_path, extra):
    """`python -m scoary_amd` under torch.distributed.run with two ranks on the one GPU
    (SCOARY_SHARE_GPU=1, SCOARY_DIST_BACKEND=gloo): every rank parses one byte range of the
    gene table, takes one contiguous gene shard through the kernels, the per-gene records are
    all-gathered, rank 0 writes -- and the result files are byte-identical to a single-process
    run's (Tree.nwk included in default mode).  Reference analogue: scoary/methods.py:1076-1122."""
    inputs = ["-g", os.path.join(exampledir, "Gene_presence_absence.csv"),
              "-t", os.path.join(exampledir, "Tetracycline_resistance.csv")]
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    one, two = tmp_path / "one", tmp_path / "two"
    cmd = [sys.executable, "-m", "scoary_amd"] + inputs + extra + ["--no-time", "-u"]
    out = subprocess.run(cmd + ["-o", str(one)], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    env2 = dict(env, SCOARY_SHARE_GPU="1", SCOARY_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
                          "-m", "scoary_amd"] + inputs + extra + ["--no-time", "-u", "-o", str(two)],
                         capture_output=True, text=True, timeout=900, env=env2, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "byte ranges" in out.stdout                       # the ranks split the reading
    names = sorted(f for f in os.listdir(one) if not f.startswith("scoary"))
    assert any(f.endswith(".results.csv") for f in names)
    assert names == sorted(f for f in os.listdir(two) if not f.startswith("scoary"))
    for f in names:
        assert (one / f).read_bytes() == (two / f).read_bytes(), f
    assert os.path.exists(two / "scoary.rank1.log")          # rank 1 ran, logged, wrote no results
