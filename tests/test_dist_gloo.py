"""Multi-process path on CPU: world_size 2 (and 3, uneven shards) over gloo.
Each rank computes its gene shard -- here with the CPU oracle standing in for
the GPU engine, which is what the rank-local step is on a GPU box -- and the
per-gene records are all-gathered by scoary_amd.dist exactly as under RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _case():
    rng = np.random.default_rng(21)
    G, N, T, P = 203, 90, 3, 40
    genes = (rng.random((G, N)) < rng.uniform(0.05, 0.95, (G, 1))).astype(np.uint8)
    traits = (rng.random((T, N)) < 0.45).astype(np.uint8)
    traits[1, ::11] = 2
    return genes, traits, N, P, 77


def _worker(rank, world, port, outq, kind="stride"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), SCOARY_GENE_PARTITION=kind)
    from oracle import oracle as orc
    from scoary_amd import dist as sd
    from scoary_amd.engine import pack_bits_rows
    w, r, _ = sd.init_from_env()            # gloo: no GPU visible here
    assert (w, r) == (world, rank) and sd.is_distributed()
    genes, traits, N, P, seed = _case()
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    T = traits.shape[0]

    def local(sel):
        if len(range(*sel.indices(genes.shape[0]))) == 0:
            return torch.zeros((T, 0, sd.REC_WORDS), dtype=torch.int32)
        gb = orc.pack_rows(np.ascontiguousarray(genes[sel]))
        c = orc.counts_packed(gb, tb, mb).transpose(1, 0, 2).copy()
        o, p = orc.fisher_many(c.reshape(-1, 4))
        rr = orc.permute_r(gb, tb, mb, N, P, seed).T.copy()
        return sd.pack_records(torch.from_numpy(c), torch.from_numpy(p.reshape(T, -1)),
                               torch.from_numpy(o.reshape(T, -1)),
                               torch.from_numpy(rr.view(np.int32)))

    rec = sd.associate_sharded(local, genes.shape[0])
    out = sd.numpy_records(rec)
    outq.put((rank, out["counts"], out["p"], out["odds"], out["r"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,kind", [(2, "stride"), (3, "stride"), (3, "contiguous")])
def test_gene_shard_all_gather_gloo(world, kind):
    """Every rank computes its shard (stride domains by default, the reference's partition;
    uneven at world 3) and the all-gathered, re-woven records equal the single-process result."""
    from oracle import oracle as orc
    from scoary_amd.engine import pack_bits_rows
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    genes, traits, N, P, seed = _case()
    gb = orc.pack_rows(genes)
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    c = orc.counts_packed(gb, tb, mb).transpose(1, 0, 2)
    o, p = orc.fisher_many(np.ascontiguousarray(c).reshape(-1, 4))
    r = orc.permute_r(gb, tb, mb, N, P, seed).T
    for rank, gc, gp, go, gr in got:         # every rank holds the full result
        assert np.array_equal(gc, c)
        assert np.array_equal(gp.ravel(), p)
        np.testing.assert_array_equal(go.ravel(), o)
        assert np.array_equal(gr, r)


def test_shard_bounds_cover_exactly():
    from scoary_amd.dist import max_shard, shard_bounds
    for G in (1, 7, 8, 9, 50000, 200000, 1000003):
        for world in (1, 2, 3, 8):
            b = shard_bounds(G, world)
            assert b[0][0] == 0 and b[-1][1] == G and len(b) == world
            assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
            sizes = [y - x for x, y in b]
            assert max(sizes) - min(sizes) <= 1 and max(sizes) == max_shard(G, world)


def test_gene_partition_covers_and_weaves():
    """GenePartition: the shards of every rank are disjoint and cover the genes; lengths add up and
    the longest is `cap`; weave() puts gathered (padded) shards back into gene order -- numpy and
    torch, stride and contiguous, even and uneven splits, more ranks than genes."""
    from scoary_amd.dist import GenePartition
    for kind in ("stride", "contiguous"):
        for G in (1, 5, 8, 9, 203, 1000):
            for world in (1, 2, 3, 8):
                part = GenePartition(G, world, kind)
                seen = np.zeros(G, dtype=np.int64)
                T, W = 2, 3
                full = np.arange(T * G * W, dtype=np.int32).reshape(T, G, W)
                recv = np.full((world, T, part.cap, W), -1, dtype=np.int32)
                for r in range(world):
                    sel = part.index(r)
                    seen[sel] += 1
                    n = part.length(r)
                    assert n == len(range(G)[sel]) and n <= part.cap
                    recv[r, :, :n] = full[:, sel]
                assert (seen == 1).all() and sum(part.lengths()) == G and max(part.lengths()) == part.cap
                assert np.array_equal(part.weave(recv), full)
                assert torch.equal(part.weave(torch.from_numpy(recv)), torch.from_numpy(full))
    # the stride shard IS the reference's domain (scoary/methods.py:1076-1078)
    part = GenePartition(11, 4)
    assert [list(range(11))[part.index(t)] for t in range(4)] == [list(range(t, 11, 4)) for t in range(4)]


def _list_work(rows_ones, N, part):
    """Minority entries per rank: what the list-driven permutation kernel's time follows."""
    minority = np.minimum(rows_ones, N - rows_ones)
    return np.array([minority[part.index(r)].sum() for r in range(part.world)], dtype=np.float64)


def test_stride_shards_balance_list_work_on_roary_order(exampledir):
    """Roary writes its table sorted by gene frequency (the reference's own
    exampledata: 100, ..., 0 isolates), and the list kernel's cost follows min(ones, N - ones).
    An 8-way contiguous equal-count split of that table leaves two ranks idle (max / mean 1.84);
    the stride domains -- the reference's own partition -- are within 10 % of the mean (in fact
    within 1 %)."""
    from scoary_amd import methods as m
    from scoary_amd.dist import GenePartition
    with open(os.path.join(exampledir, "Gene_presence_absence.csv"), "r", newline=None) as f:
        table = m.Csv_to_dic_Roary(f, ",", [], startcol=14)["Roarydic"]
    N = len(table.strains)
    ones = np.unpackbits(table.rows64.view(np.uint8), axis=1).sum(axis=1).astype(np.int64)
    assert (np.diff(ones) <= 0).all()                         # the file really is frequency-sorted
    G = len(ones)
    old = _list_work(ones, N, GenePartition(G, 8, "contiguous"))
    new = _list_work(ones, N, GenePartition(G, 8, "stride"))
    assert old.min() == 0 and old.max() / old.mean() > 1.8    # what rounds 1-5 did
    assert new.max() / new.mean() <= 1.10
    assert new.max() / new.mean() <= 1.01
    # and on a synthetic cfg3-like spectrum sorted the same way, 2 ... 8 ranks
    rng = np.random.default_rng(3)
    ones = np.sort(rng.binomial(2000, rng.uniform(0.02, 0.98, 50000)))[::-1]
    for world in (2, 4, 8):
        w = _list_work(ones, 2000, GenePartition(50000, world, "stride"))
        assert w.max() / w.mean() <= 1.001
        c = _list_work(ones, 2000, GenePartition(50000, world, "contiguous"))
        assert c.max() / c.mean() > (1.4 if world > 2 else 0.99)


def test_record_pack_roundtrip():
    from scoary_amd import dist as sd
    rng = np.random.default_rng(0)
    T, G = 3, 17
    c = torch.from_numpy(rng.integers(0, 5000, (T, G, 4)).astype(np.int32))
    p = torch.from_numpy(rng.random((T, G)))
    o = torch.from_numpy(np.where(rng.random((T, G)) < 0.2, np.inf, rng.random((T, G)) * 50))
    o[0, 0] = float("nan")
    r = torch.from_numpy(rng.integers(0, 2**31 - 1, (T, G)).astype(np.int32))
    rec = sd.pack_records(c, p, o, r)
    d = sd.unpack_records(rec)
    assert torch.equal(d["counts"], c) and torch.equal(d["p"], p) and torch.equal(d["r"], r)
    assert torch.equal(d["odds"].view(torch.int64), o.view(torch.int64))   # bit pattern, nan incl.
    # numpy_records: one copy of the record tensor, numpy slices (no torch kernels) -- same content
    h = sd.numpy_records(rec)
    assert np.array_equal(h["counts"], c.numpy()) and np.array_equal(h["p"], p.numpy())
    assert np.array_equal(h["odds"].view(np.int64), o.numpy().view(np.int64))
    assert np.array_equal(h["r"], r.numpy().view(np.uint32)) and not h["nstop"].any()
    assert all(v.flags["C_CONTIGUOUS"] for v in h.values())


def _gather_worker(rank, world, port, outq):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from scoary_amd import dist as sd
    sd.init_from_env()
    G, T = 11, 2
    full = torch.arange(T * G * sd.REC_WORDS, dtype=torch.int32).view(T, G, sd.REC_WORDS)
    outs = []
    for kind in ("stride", "contiguous"):
        part = sd.GenePartition(G, world, kind)
        for async_op in (False, True):
            _, finish = sd.gather_genes(full[:, part.index(rank)].contiguous(), G, dst=0, async_op=async_op,
                                        partition=part)
            outs.append(finish())
    # weave=False: the blocks as they arrived; the reader weaves them (what bench.py's Exchange does)
    part = sd.GenePartition(G, world)
    _, finish = sd.gather_genes(full[:, part.index(rank)].contiguous(), G, dst=0, partition=part, weave=False)
    raw = finish()
    if rank == 0:
        assert tuple(raw.shape) == (world, T, part.cap, sd.REC_WORDS) and torch.equal(part.weave(raw), full)
    else:
        assert raw is None
    outq.put((rank, [None if o is None else o.numpy() for o in outs]))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_to_rank0_gloo():
    """bench.py's exchange step: a true gather (uneven shards; stride and contiguous), sync and async."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gather_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = np.arange(2 * 11 * 10, dtype=np.int32).reshape(2, 11, 10)
    assert all(np.array_equal(o, want) for o in got[0])
    assert len(got[0]) == 4 and got[1] == [None] * 4 and got[2] == [None] * 4


def _label_shard_worker(rank, world, port, outq):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from scoary_amd import dist as sd
    sd.init_from_env()
    sh = sd.LabelShards()
    res = []
    for nflat, tw in ((7, 5), (3, 4), (12, 1), (1, 6)):        # fewer tiles than ranks included
        per, first, count = sh.share(nflat)
        tiles = torch.full((sh.padded_words(nflat, tw) + 3,), -1, dtype=torch.int32)
        full = torch.arange(nflat * tw, dtype=torch.int32) + 100 * nflat
        tiles[first * tw:(first + count) * tw] = full[first * tw:(first + count) * tw]    # this rank's share
        sh.all_gather(tiles, nflat, tw)
        res.append((per, first, count, tiles[:nflat * tw].numpy().copy(), full.numpy()))
    outq.put((rank, res, sh.bytes_gathered))
    dist.barrier()
    dist.destroy_process_group()


def test_label_tile_shards_all_gather_gloo():
    """dist.LabelShards: every rank fills its contiguous share of the flat (trait, tile) array and
    one all_gather_into_tensor, in place, supplies the others' -- the permutation shards of the
    label generator (engine.label_shards, bench.py --label-shards)."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_label_shard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, res, nbytes in got:
        for per, first, count, tiles, full in res:
            assert np.array_equal(tiles, full), rank
            assert first <= rank * per and 0 <= count <= per
        assert nbytes > 0


def _exchange_worker(rank, world, port, outq):
    """bench.py's Exchange (async gather, two receive buffers, drained at the
    barrier) driven for several steps on CPU tensors."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import types
    import bench
    from scoary_amd import dist as sd
    sd.init_from_env()
    T, G, steps = 2, 5, 5
    ex = bench.Exchange(torch, types.SimpleNamespace(device="cpu"), world, rank, T,
                        sd.GenePartition(G * world, world, "contiguous"))
    for step in range(steps):
        base = 1000 * step + 100 * rank
        res = {"counts": torch.full((T, G, 4), base, dtype=torch.int32),
               "p": torch.full((T, G), float(base), dtype=torch.float64),
               "odds": torch.full((T, G), float(base) + 0.5, dtype=torch.float64),
               "r": torch.full((T, G), base + 7, dtype=torch.int32)}
        ex.submit(res)
        assert len(ex.pending) <= 1 + (ex.kind == "gather")
    ex.drain()
    dist.barrier()
    out = None
    if rank == 0:
        last = ex.recv[(steps - 1) % 2]                        # [world, T, G, words] of the last step
        d = sd.unpack_records(last.reshape(world * T, G, sd.REC_WORDS))
        out = (ex.kind, d["counts"].view(world, T, G, 4)[:, 0, 0, 0].tolist(),
               d["r"].view(world, T, G)[:, 1, 4].tolist(), d["p"].view(world, T, G)[:, 0, 0].tolist())
    outq.put((rank, out))
    dist.destroy_process_group()


def test_bench_exchange_pipeline_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kind, counts, r, pv = got[0]
    assert kind == "gather"
    assert counts == [4000, 4100] and r == [4007, 4107] and pv == [4000.0, 4100.0]
    assert got[1] is None


def test_bench_launcher_runs_its_own_ranks_gloo():
    """`python bench.py --gpus 2` outside a launcher re-executes itself under
    torch.distributed.run with two ranks.  --dry-exchange keeps it on
    CPU: gloo process group, the bench Exchange pipeline with fabricated records, and
    rank 0's check that a valid block arrived from every rank -- weak and strong."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    for scaling in ("weak", "strong"):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2",
                              "--dry-exchange", "--steps", "3", "--scaling", scaling],
                             capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
        assert out.returncode == 0, out.stderr[-2000:]
        line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
        d = json.loads(line)
        assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == scaling
        assert d["exchange"] == "gather"


def test_bench_refuses_more_gpus_than_visible():
    """--gpus N with fewer than N devices must fail loudly, not run one rank and report
    n_gpus: 1 (what round 1 did)."""
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"],
                         capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0
    assert "--gpus 64" in out.stderr and "visible" in out.stderr


# ------------------------------------------------ rank-sharded table read -----
def _read_worker(rank, world, port, path, outq):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import logging
    from scoary_amd import dist as sd
    from scoary_amd import methods as m
    sd.init_from_env()
    seen = []

    class Grab(logging.Handler):
        def emit(self, rec):
            seen.append(rec.getMessage())
    m.log.addHandler(Grab())
    m.log.setLevel(logging.INFO)
    with open(path, "r", newline=None) as f:
        gd = m.Csv_to_dic_Roary(f, ",", [3], startcol=14)
    t = gd["Roarydic"]
    outq.put((rank, list(t.ids), list(t.annotation), t.rows64.copy(), gd["Strains"],
              any("byte ranges" in s for s in seen)))
    dist.barrier()
    dist.destroy_process_group()


def _sharded_read(path, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_read_worker, args=(r, world, port, path, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict((x[0], x[1:]) for x in (q.get(timeout=180) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return got


def test_ranks_read_the_table_in_byte_ranges_gloo(exampledir, tmp_path):
    """Under torchrun every rank tokenises one byte range of the gene presence/absence
    file and the parts are all-gathered: the table every rank ends up
    with equals the single-process read -- identifiers, text columns, bit rows -- for 2 and
    3 ranks; a file whose part boundary falls inside a quoted multi-line cell makes the
    ranks agree to read it whole."""
    from scoary_amd import methods as m
    path = os.path.join(exampledir, "Gene_presence_absence.csv")
    with open(path, "r", newline=None) as f:
        want = m.Csv_to_dic_Roary(f, ",", [3], startcol=14)
    wt = want["Roarydic"]
    for world in (2, 3):
        got = _sharded_read(path, world)
        for rank in range(world):
            ids, ann, rows64, strains, sharded = got[rank]
            assert sharded, "rank %d did not take the sharded path" % rank
            assert ids == list(wt.ids) and ann == list(wt.annotation) and strains == want["Strains"]
            assert np.array_equal(rows64, wt.rows64)
    # a quoted cell with line breaks across the middle of the file
    header = ["Gene", "Non-unique Gene name", "Annotation"] + ["c%d" % i for i in range(11)] + ["s1", "s2", "s3"]
    rows = [",".join(header)]
    for i in range(40):
        ann = '"line one\nline two\n' + "x\n" * 400 + 'end"' if i == 20 else "plain"
        rows.append(",".join(["g%d" % i, "", ann] + [""] * 11 + ["1", "0" if i % 3 else "1", "1" if i % 2 else ""]))
    tricky = tmp_path / "tricky.csv"
    tricky.write_text("\n".join(rows) + "\n")
    with open(tricky, "r", newline=None) as f:
        w2 = m.Csv_to_dic_Roary(f, ",", [], startcol=14)
    got = _sharded_read(str(tricky), 2)
    for rank in range(2):
        ids, ann, rows64, strains, sharded = got[rank]
        assert not sharded                                  # fell back, on both ranks
        assert ids == list(w2["Roarydic"].ids) and len(ids) == 40
        assert np.array_equal(rows64, w2["Roarydic"].rows64)
        assert ann[20].startswith("line one\nline two")
