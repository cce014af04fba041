"""Two real ranks with real kernels in the driver-run evidence.

No multi-GPU box is available to the build or to `pytest -m gpu`, so the two ranks share the
one device (`--share-gpu`) and exchange over gloo; everything else is the multi-GPU path as
it runs on 8 GPUs: bench.py's own launcher (re-exec under torch.distributed.run), one
process per rank, gene shards (weak: every rank its own shard; strong: the reference's stride
domains of the config's genes, dist.GenePartition), labels regenerated from the seed on every
rank, scoary_pack_records, the asynchronous gather of 10-word records overlapped with the next
step, and rank 0's checks.  `--verify-gather`: rank 0 recomputes every rank's shard alone and
compares the records it received bit for bit, then weaves the blocks back into gene order and
compares them with ONE run over the whole matrix.  Reference analogue: the stride domains and
the result weave of scoary/methods.py:1076-1097, :1115-1122.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(extra, ranks=2, config="cfg2", steps=4):
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--share-gpu",
                          "--backend", "gloo", "--config", config, "--steps", str(steps), "--warmup", "1",
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
    """Three ranks, strong scaling: 10 000 genes do not divide by three (stride domains of 3334 /
    3333 / 3333), so the gather pads the shorter shards to the longest and the weave drops the pad
    slots again; every record still equals the single-rank run's."""
    d = _bench(["--scaling", "strong"], ranks=3)
    assert d["config"]["gene_partition"].startswith("stride")
    assert d["n_gpus"] == 3 and d["rccl_ranks"] == 3 and d["gather_matches_single_rank"] is True
    assert d["config"]["genes_total"] == 10_000
    assert [r["genes"] for r in d["per_rank"]] == [3334, 3333, 3333]
    assert [r["exchange_bytes"] for r in d["per_rank"]] == [3334 * 40, 3333 * 40, 3333 * 40]


@pytest.mark.parametrize("extra", [["--no_pairwise", "-e", "200", "--seed", "7"], ["--collapse", "-c", "I", "-p", "0.05"]],
                         ids=["fisher-permutations", "pairwise-collapse"])
def test_command_line_two_ranks_share_the_gpu(exampledir, tmp_path, extra):
    """`python -m scoary_amd` under torch.distributed.run with two ranks on the one GPU
    (SCOARY_SHARE_GPU=1, SCOARY_DIST_BACKEND=gloo): every rank parses one byte range of the
    gene table, takes one stride gene shard through the kernels, the per-gene records are
    all-gathered and woven back into file order, rank 0 writes -- and the result files are byte-identical to a single-process
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


# ---- the 8-GPU configs of BASELINE.json at 8 ranks ---------------------
# No multi-GPU node reaches the build or `pytest -m gpu`, so the first run on eight devices will be
# the driver's: these rehearse exactly that launch -- bench.py's own launcher, eight processes,
# eight gene shards, labels regenerated from the seed on every rank, the asynchronous gather of
# eight blocks overlapped with the next step, rank 0's checks -- with the eight ranks time-sharing
# the one device over gloo.  Reference analogue: scoary/methods.py:1076-1097 (stride domains),
# :1115-1122 (result weave).

def test_bench_eight_ranks_cfg4_strong_split():
    """cfg4 is DEFINED as gene-sharded 8x (BASELINE.json configs[3]): 200 000 variants x 5000
    isolates, --permute 10000, 25 000 genes per rank; every gathered record equals rank 0's
    own recomputation of that rank's shard."""
    d = _bench(["--scaling", "strong", "--no-cpu-baseline"], ranks=8, config="cfg4", steps=2)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["scaling"] == "strong"
    assert d["gather_matches_single_rank"] is True
    assert d["config"]["genes_total"] == 200_000 and d["config"]["parallelism"] == "gene-shard x8"
    assert d["config"]["isolates"] == 5000 and d["config"]["permutations"] == 10_000
    pr = d["per_rank"]
    assert [r["rank"] for r in pr] == list(range(8)) and [r["genes"] for r in pr] == [25_000] * 8
    assert all(r["exchange_bytes"] == 25_000 * 40 for r in pr)
    assert all(r["kernel_ms"]["k_permute_lists"] > 0 for r in pr)
    assert abs(d["value"] - 200_000 * 10_000 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]


@pytest.mark.parametrize("partition", ["stride", "contiguous"])
def test_bench_eight_ranks_frequency_sorted_genes(partition):
    """Roary writes its table sorted by gene frequency, and the list kernel's cost
    follows a gene's minority count.  cfg3 with its rows in that order (--gene-order sorted),
    split 8 ways: the stride domains (the reference's own partition, scoary/methods.py:1076-1078)
    give every rank the same list work to within 1 %; the contiguous equal-count blocks of rounds
    1-5 are > 1.4x apart (kept as --partition contiguous, the A/B).  Either way the gathered blocks
    equal rank 0's recomputation shard by shard AND, woven back into gene order, one run over the
    whole sorted matrix."""
    d = _bench(["--scaling", "strong", "--no-cpu-baseline", "--gene-order", "sorted", "--partition", partition],
               ranks=8, config="cfg3", steps=2)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["gather_matches_single_rank"] is True
    assert d["config"]["gene_order"] == "sorted" and d["config"]["gene_partition"].startswith(partition)
    pr = d["per_rank"]
    assert [r["genes"] for r in pr] == [6250] * 8
    work = [r["list_entries"] for r in pr]
    spread = max(work) / (sum(work) / 8.0)
    if partition == "stride":
        assert spread <= 1.01
    else:
        assert spread > 1.4


def test_bench_two_ranks_strong_curve_next_to_the_weak_line():
    """`scaling_strong`: next to the weak line, the same process group times the
    strong split of cfg3 (50 000 genes) and cfg4 (200 000 variants) over its ranks -- exchange
    included, a launch-bound shard replayed from a hipGraph -- so the driver's 1 / 2 / 4 / 8 runs
    yield the strong curve of SURVEY 8e as well."""
    d = _bench(["--scaling", "weak", "--no-cpu-baseline", "--strong-extra", "on"], ranks=2, config="cfg2", steps=2)
    ss = d["scaling_strong"]
    for cfg, G, T in (("cfg3", 50_000, 10), ("cfg4", 200_000, 1)):
        e = ss[cfg]
        assert e["n_gpus"] == 2 and e["rccl_ranks"] == 2 and e["genes_per_gpu"] == [G // 2, G // 2]
        assert e["gene_partition"] == "stride" and e["steps"] == 2
        assert abs(e["value"] - G * T * 10_000 / (e["ms_per_step"] * 1e-3)) < 1e-6 * e["value"]
        assert [r["rank"] for r in e["per_rank"]] == [0, 1]
        for r in e["per_rank"]:
            assert r["kernel_ms"]["k_permute_lists"] > 0 and r["exchange_exposed_ms"] is not None
            assert r["ms_per_step"] <= e["ms_per_step"] * (1 + 1e-9)
        work = [r["list_entries"] for r in e["per_rank"]]
        assert max(work) / (sum(work) / 2.0) < 1.02
    assert ss["cfg3"]["hip_graph"] is False and ss["cfg4"]["hip_graph"] is False     # 2.5e9 / 1e9 tests per rank


def test_bench_eight_ranks_cfg4_label_tile_shards():
    """The same launch with PERMUTATION SHARDS of the label tiles (bench.py --label-shards,
    dist.LabelShards): every rank generates 5 of the 40 tiles of 256 permutations
    (scoary_perm_generate_tiles_range) and one all_gather_into_tensor, in place, supplies the
    other 35; --verify-gather then compares every rank's records with rank 0's recomputation
    from tiles it generated all by itself -- so a tile that arrived wrong, or in the wrong
    place, shows up as a wrong exceedance count."""
    d = _bench(["--scaling", "strong", "--no-cpu-baseline", "--label-shards"], ranks=8, config="cfg4", steps=2)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["gather_matches_single_rank"] is True
    assert d["config"]["label_tiles"].startswith("1/8 per rank")
    tile_bytes = 4 * ((5000 + 1) * 8)                      # one tile: 5001 rows of 8 dwords
    assert all(r["label_tile_bytes_gathered"] == 7 * 5 * tile_bytes for r in d["per_rank"])


def test_bench_three_ranks_label_tile_shards_many_traits():
    """Label-tile shards where the flat (trait, tile) array does not divide by the ranks: 4 traits x
    3 tiles of 512 permutations over 3 ranks = 4 tiles each, a share that crosses a trait boundary."""
    d = _bench(["--scaling", "weak", "--no-cpu-baseline", "--label-shards", "--traits", "4",
                "--permutations", "1500", "--genes", "3000"], ranks=3, config="cfg2", steps=2)
    assert d["n_gpus"] == 3 and d["rccl_ranks"] == 3 and d["gather_matches_single_rank"] is True
    assert d["config"]["label_tiles"].startswith("1/3 per rank")


def test_bench_eight_ranks_cfg5_weak_shards():
    """cfg5 (BASELINE.json configs[4], 8 x MI355X): every rank its own shard of the 10 000-isolate
    x 50-trait x 100 000-permutation problem (8000 genes per rank here instead of 125 000 -- the
    full shard is tests/test_gpu_full_size.py's), three label-tile batches per step, weak scaling."""
    d = _bench(["--scaling", "weak", "--genes", "8000", "--no-cpu-baseline"], ranks=8, config="cfg5", steps=2)
    assert d["n_gpus"] == 8 and d["rccl_ranks"] == 8 and d["scaling"] == "weak"
    assert d["gather_matches_single_rank"] is True
    assert d["config"]["genes_total"] == 64_000 and d["config"]["genes_per_gpu"] == 8000
    assert d["config"]["isolates"] == 10_000 and d["config"]["traits"] == 50
    assert d["config"]["permutations"] == 100_000
    assert len(d["per_rank"]) == 8 and all(r["exchange_bytes"] == 50 * 8000 * 40 for r in d["per_rank"])


def _torchrun(nproc, args, env, timeout=900):
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node",
                           str(nproc), "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + args,
                          capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)


def _clean_env():
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_one_rank_through_rccl(scaling):
    """The real backend in the driver-run evidence: one rank under torch.distributed.run with
    --backend nccl (= RCCL) -- init_process_group("nccl", device_id=...), scoary_pack_records,
    dist.gather on device tensors, the asynchronous drain, the MAX all_reduce on a device tensor,
    rank 0's record checks -- everything the 8-GPU run does except a second device."""
    out = _torchrun(1, [os.path.join(ROOT, "bench.py"), "--exercise-exchange", "--backend", "nccl",
                        "--verify-gather", "--config", "cfg2", "--scaling", scaling, "--steps", "4",
                        "--warmup", "1", "--no-cpu-baseline"] +
                    (["--strong-extra", "on"] if scaling == "weak" else []), _clean_env())
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["gather_matches_single_rank"] is True
    assert d["config"]["exchange"].startswith("rccl gather")
    assert d["per_rank"][0]["exchange_bytes"] == 10_000 * 40
    assert d["per_rank"][0]["exchange_exposed_ms"] is not None
    # cfg2 is launch-bound: the sharded rank replays its local step (record packing included) from
    # a hipGraph, the gather stays outside (round 6)
    assert d["config"]["hip_graph"] is True and d["config"]["hip_graph_auto"] is True
    if scaling == "weak":
        # the strong-curve entries of the multi-GPU line through the same RCCL group (one rank here)
        for cfg, G in (("cfg3", 50_000), ("cfg4", 200_000)):
            e = d["scaling_strong"][cfg]
            assert e["n_gpus"] == 1 and e["rccl_ranks"] == 1 and e["genes_per_gpu"] == [G]
            assert e["per_rank"][0]["exchange_exposed_ms"] is not None and e["value"] > 1e11
    else:
        assert d["scaling_strong"] is None


def test_bench_rank_of_eight_cfg4_through_rccl_replays_a_graph():
    """One rank of cfg4's 8-way split (25 000 variants) under torch.distributed.run through real
    RCCL: 2.5e8 tests per step, launch-bound -- the local step runs as a hipGraph replay with the
    records packed inside it (two graphs, alternating record buffers), the gather outside; the
    gathered records equal an eager single-rank run's."""
    out = _torchrun(1, [os.path.join(ROOT, "bench.py"), "--exercise-exchange", "--backend", "nccl",
                        "--verify-gather", "--config", "cfg4", "--genes", "25000", "--scaling", "strong",
                        "--steps", "20", "--warmup", "3", "--no-cpu-baseline"], _clean_env())
    assert out.returncode == 0, (out.stdout[-1000:], out.stderr[-3000:])
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["rccl_ranks"] == 1 and d["gather_matches_single_rank"] is True
    assert d["config"]["hip_graph"] is True and d["config"]["genes_per_gpu"] == 25_000
    assert d["per_rank"][0]["exchange_exposed_ms"] is not None and d["ms_per_step"] < 1.0


def test_command_line_one_rank_through_rccl(exampledir, tmp_path):
    """`python -m scoary_amd` under torch.distributed.run with one rank and SCOARY_EXERCISE_DIST=1:
    the command line joins an RCCL group, all-gathers the per-gene records and the bit rows on
    device tensors and shuts the group down; outputs byte-identical to the plain run."""
    inputs = ["-g", os.path.join(exampledir, "Gene_presence_absence.csv"),
              "-t", os.path.join(exampledir, "Tetracycline_resistance.csv"),
              "--no_pairwise", "-e", "100", "--seed", "3", "--no-time"]
    env = _clean_env()
    one, two = tmp_path / "one", tmp_path / "two"
    out = subprocess.run([sys.executable, "-m", "scoary_amd"] + inputs + ["-o", str(one)],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    out = _torchrun(1, ["-m", "scoary_amd"] + inputs + ["-o", str(two)],
                    dict(env, SCOARY_EXERCISE_DIST="1"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    names = sorted(f for f in os.listdir(one) if f.endswith(".results.csv"))
    assert names and names == sorted(f for f in os.listdir(two) if f.endswith(".results.csv"))
    for f in names:
        assert (one / f).read_bytes() == (two / f).read_bytes(), f


def test_command_line_eight_ranks_share_the_gpu(exampledir, tmp_path):
    """The command line at the node's full width: eight ranks under torch.distributed.run on the one
    GPU (gloo), each parsing one byte range of the gene table and taking one of eight stride gene
    shards (9 001 frequency-sorted rows: uneven) through the kernels; result files and Tree.nwk byte-identical to one process."""
    inputs = ["-g", os.path.join(exampledir, "Gene_presence_absence.csv"),
              "-t", os.path.join(exampledir, "Tetracycline_resistance.csv"),
              "-e", "100", "--seed", "11", "--no-time", "-u", "-c", "I", "EPW", "-p", "0.05", "0.5"]
    env = _clean_env()
    one, eight = tmp_path / "one", tmp_path / "eight"
    out = subprocess.run([sys.executable, "-m", "scoary_amd"] + inputs + ["-o", str(one)],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    out = _torchrun(8, ["-m", "scoary_amd"] + inputs + ["-o", str(eight)],
                    dict(env, SCOARY_SHARE_GPU="1", SCOARY_DIST_BACKEND="gloo"))
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    names = sorted(f for f in os.listdir(one) if not f.startswith("scoary"))
    assert any(f.endswith(".results.csv") for f in names) and "Tree.nwk" in names
    assert names == sorted(f for f in os.listdir(eight) if not f.startswith("scoary"))
    for f in names:
        assert (one / f).read_bytes() == (eight / f).read_bytes(), f
    assert all(os.path.exists(eight / ("scoary.rank%d.log" % r)) for r in range(1, 8))


def test_a_failed_gather_check_ends_every_rank(tmp_path):
    """when rank 0's --verify-gather finds a difference, its verdict is broadcast
    and EVERY rank exits non-zero at once -- before, rank 0 raised while the others sat in a
    barrier until the launcher's timeout.  A corrupted record is injected on rank 0."""
    import time
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t0 = time.perf_counter()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu",
                          "--backend", "gloo", "--config", "cfg2", "--steps", "2", "--warmup", "1",
                          "--verify-gather", "--inject-gather-fault"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "gathered records differ from a single-rank run" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]      # no bench line
    assert time.perf_counter() - t0 < 300                                        # nobody waited for a timeout
