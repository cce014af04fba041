"""Gene sharding across the GPUs of one node (one process per GPU).

Genes are independent units of the path, so the packed matrix is split by rows, one shard
per rank; trait, mask and permutation vectors are replicated (permutations are regenerated
identically on every rank from the counter-based seed -- zero traffic).  The path's only
exchange step is the gather of per-gene result records at the end: one ``all_gather`` over
RCCL (xGMI) on GPU tensors, or gloo on CPU tensors in the tests.

The shards are the reference's own STRIDE domains, ``range(rank, G, world)``
(scoary/methods.py:1076-1078), and the gathered records are woven back into file order the way
the reference weaves its workers' results (:1115-1122) -- ``GenePartition``.  Rounds 1-5 cut
contiguous equal-count blocks: the cost of the list-driven permutation kernel follows a gene's
minority count, and Roary writes its table sorted by gene frequency, so a contiguous 8-way
split of the reference's own exampledata had two idle ranks and max / mean = 1.84 of list work
(VERDICT r5 #2).  A stride shard is a 1-in-world sample of ANY monotone (or slowly varying)
order: every rank gets the same length spectrum, so list work, per-gene fixed cost and the
Fisher pass balance together without a cost model.
"""
import os

import numpy as np

REC_WORDS = 10  # int32 words per (trait, gene): 4 counts, p (2), odds (2), r (1), nstop (1)


def _torch():
    import torch
    return torch


def shard_bounds(G, world):
    """Contiguous, balanced [start, stop) per rank; the first G % world ranks
    get one extra gene."""
    base, extra = divmod(int(G), int(world))
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append((start, start + n))
        start += n
    return out


def max_shard(G, world):
    return -(-int(G) // int(world))


class GenePartition:
    """Which genes each rank works on, and how gathered shards go back into gene order.

    kind "stride" (default): rank r owns genes r, r + world, r + 2 world, ... -- the reference's
    domains (scoary/methods.py:1076-1078).  kind "contiguous": equal-count blocks (shard_bounds),
    kept for weak scaling (every rank brings its own block) and as the A/B of the rehearsals."""

    def __init__(self, G, world, kind="stride"):
        if kind not in ("stride", "contiguous"):
            raise ValueError("unknown gene partition %r" % (kind,))
        self.G, self.world, self.kind = int(G), int(world), kind
        self.cap = max_shard(G, world)
        self._bounds = shard_bounds(G, world) if kind == "contiguous" else None

    def index(self, rank):
        """slice of the gene axis that is rank's shard."""
        if self.kind == "stride":
            return slice(int(rank), self.G, self.world)
        a, b = self._bounds[rank]
        return slice(a, b, 1)

    def length(self, rank):
        return len(range(*self.index(rank).indices(self.G)))

    def lengths(self):
        return [self.length(r) for r in range(self.world)]

    def weave(self, recv):
        """recv [world, T, cap, W] (torch tensor or numpy array; shard r in recv[r, :, :length(r)])
        -> [T, G, W] in gene order."""
        world, T, cap, W = recv.shape
        if world != self.world or cap != self.cap:
            raise ValueError("gathered block is %s, partition wants (%d, T, %d, W)"
                             % (tuple(recv.shape), self.world, self.cap))
        if self.kind == "stride":
            # gene j * world + r = recv[r, :, j]; the pad slots of the short shards land at >= G
            if isinstance(recv, np.ndarray):
                return np.ascontiguousarray(recv.transpose(1, 2, 0, 3).reshape(T, cap * world, W)[:, :self.G])
            return recv.permute(1, 2, 0, 3).reshape(T, cap * world, W)[:, :self.G].contiguous()
        parts = [recv[r, :, :self.length(r)] for r in range(world)]
        if isinstance(recv, np.ndarray):
            return np.ascontiguousarray(np.concatenate(parts, axis=1))
        return _torch().cat(parts, dim=1).contiguous()


def pack_records(counts, p, odds, r, nstop=None):
    """counts int32 [T,Gs,4], p/odds float64 [T,Gs], r int32 [T,Gs] or None, nstop int32
    [T,Gs] or None (the permutation count a gene's sequential estimator stopped at, 0 = it
    ran to the end; only with --permute-early-abort) -> int32 [T,Gs,10] (bit patterns
    preserved)."""
    torch = _torch()
    T, Gs = p.shape
    if r is None:
        r = torch.zeros((T, Gs), dtype=torch.int32, device=p.device)
    if nstop is None:
        nstop = torch.zeros((T, Gs), dtype=torch.int32, device=p.device)
    def i32(x):
        return x.reshape(-1).clone().view(torch.int32).view(T, Gs, 2)
    return torch.cat([counts.reshape(T, Gs, 4), i32(p), i32(odds),
                      r.reshape(T, Gs, 1), nstop.reshape(T, Gs, 1)], dim=2).contiguous()


def unpack_records(rec):
    torch = _torch()
    T, G, _ = rec.shape
    rec = rec.contiguous()
    def f64(lo):        # flat copy first: a (1, 1, 2) slice keeps the parent's odd strides
        return rec[:, :, lo:lo + 2].reshape(-1).clone().view(torch.float64).view(T, G)
    return {"counts": rec[:, :, 0:4].reshape(-1).clone().view(T, G, 4),
            "p": f64(4), "odds": f64(6),
            "r": rec[:, :, 8].reshape(-1).clone().view(T, G),
            "nstop": rec[:, :, 9].reshape(-1).clone().view(T, G)}


def is_distributed():
    """A process group with more than one rank exists.  Looks at sys.modules instead of importing:
    no group can exist if torch.distributed was never imported, and a command-line process imports
    torch on a helper thread while the main thread reads the gene table (methods.main) -- an import
    of torch.distributed from here would then deadlock against it (the importer detects that and
    raises inside torch's C++ start-up)."""
    import sys
    td = sys.modules.get("torch.distributed")
    if td is None or not hasattr(td, "is_initialized") or not hasattr(td, "get_world_size"):
        return False
    return td.is_available() and td.is_initialized() and td.get_world_size() > 1


def world_rank():
    if not is_distributed():
        return 1, 0
    import torch.distributed as dist
    return dist.get_world_size(), dist.get_rank()


def init_from_env():
    """Join the process group described by torchrun's environment (RANK,
    LOCAL_RANK, WORLD_SIZE, MASTER_*), backend nccl (= RCCL on ROCm).  No-op for
    a single process.  Returns (world, rank, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # SCOARY_EXERCISE_DIST=1 under torchrun with one rank: still join a group and go
    # through the exchange step (a 1-GPU check of the N>1 code path)
    if world > 1 or (os.environ.get("SCOARY_EXERCISE_DIST") == "1" and "RANK" in os.environ):
        torch = _torch()
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # SCOARY_SHARE_GPU=1: rank r uses device r % visible devices, and SCOARY_DIST_BACKEND=gloo
        # replaces RCCL (which wants one device per rank): a functional check of the sharded
        # command line on a single-GPU box (tests/test_gpu_two_ranks.py), not a way to run it
        backend = os.environ.get("SCOARY_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            if os.environ.get("SCOARY_SHARE_GPU") == "1":
                local_rank %= torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
        if not dist.is_initialized():
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
    return world, rank, local_rank


def all_gather_genes(rec_local, G, group=None, partition=None):
    """rec_local: this rank's [T, Gs, W] block (Gs = its shard length) ->
    the full [T, G, W] in gene order on every rank.  Shards are padded to a common length
    for the collective; ``partition`` (default: stride) says which genes they are."""
    torch = _torch()
    import torch.distributed as dist
    world = dist.get_world_size(group)
    part = partition or GenePartition(G, world)
    T, Gs, W = rec_local.shape
    cap = part.cap
    if Gs != part.length(dist.get_rank(group)):
        raise ValueError("this rank's shard has %d genes, the partition says %d"
                         % (Gs, part.length(dist.get_rank(group))))
    send = rec_local
    if Gs != cap:
        send = torch.zeros((T, cap, W), dtype=rec_local.dtype, device=rec_local.device)
        send[:, :Gs] = rec_local
    # concatenated-along-dim-0 output form: accepted by both RCCL and gloo; gloo moves host
    # memory, so device tensors are staged through the host for it (functional checks only)
    dev = rec_local.device
    staged = dist.get_backend(group) != "nccl" and dev.type == "cuda"
    send = send.contiguous().cpu() if staged else send.contiguous()
    recv = torch.empty((world * T, cap, W), dtype=rec_local.dtype, device=send.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    if staged:
        recv = recv.to(dev)
    return part.weave(recv.view(world, T, cap, W))


def gather_genes(rec_local, G, dst=0, group=None, async_op=False, recv=None, partition=None, weave=True):
    """The path's one exchange step as a true gather: every rank sends its
    [T, Gs, W] block to ``dst`` only (1/world of an all_gather's traffic; over
    xGMI each sender uses its own link to dst).  Returns (work, finish) where
    finish() -> the full [T, G, W] tensor in gene order on dst, None elsewhere.
    ``weave=False``: finish() returns the gathered blocks [world, T, cap, W] as they arrived
    (shard r in block r, GenePartition's layout) and the reader applies ``partition.weave``
    when -- and if -- it wants gene order."""
    torch = _torch()
    import torch.distributed as dist
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    part = partition or GenePartition(G, world)
    T, Gs, W = rec_local.shape
    cap = part.cap
    send = rec_local
    if Gs != cap:
        send = torch.zeros((T, cap, W), dtype=rec_local.dtype, device=rec_local.device)
        send[:, :Gs] = rec_local
    send = send.contiguous()
    parts = None
    if rank == dst:
        if recv is None:
            recv = torch.empty((world, T, cap, W), dtype=rec_local.dtype, device=rec_local.device)
        parts = [recv[r] for r in range(world)]
    work = dist.gather(send, parts, dst=dst, group=group, async_op=async_op)

    def finish():
        # a gather that has already completed needs no stream-level wait (work.wait() would put a barrier
        # on the compute stream in front of the next step: ~20 us per step on a launch-bound shard)
        if work is not None and async_op and not work.is_completed():
            work.wait()
        if rank != dst:
            return None
        return part.weave(recv) if weave else recv
    return work, finish


class LabelShards:
    """Permutation shards of the label tiles (engine.label_shards): every rank generates one
    contiguous share of a batch's flat (trait, tile) array and ONE all_gather_into_tensor over
    RCCL / xGMI supplies the rest, in place (the tile array is padded to world equal chunks).

    Spec S4's generator has no chain over the isolates, so a rank can also simply generate
    every tile itself (cfg4: ~0.03 ms on one MI355X against ~0.8 MB per rank through a
    collective whose latency alone is of that order).  Replication therefore stays the default;
    the shards are opt-in (bench.py --label-shards, SCOARY_LABEL_SHARDS=1) for shapes where the
    generator outweighs an all-gather of T * P * N / 8 bytes -- many traits x permutations on
    long rows -- and are rehearsed at 8 ranks in tests/test_gpu_two_ranks.py."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.staged = dist.get_backend(group) != "nccl"      # gloo moves host memory
        self.bytes_gathered = 0

    def share(self, nflat):
        """(tiles per rank, first tile of this rank, its tile count) for nflat flat tiles."""
        per = -(-int(nflat) // self.world)
        first = min(self.rank * per, int(nflat))
        return per, first, min(per, int(nflat) - first)

    def padded_words(self, nflat, tile_words):
        return self.world * self.share(nflat)[0] * int(tile_words)

    def all_gather(self, tiles, nflat, tile_words):
        """tiles: int32 device tensor of >= padded_words(nflat, tile_words) words whose chunk
        `rank` this rank has filled; on return every chunk is filled."""
        torch = _torch()
        import torch.distributed as dist
        per = self.share(nflat)[0]
        chunk = per * int(tile_words)
        full = tiles[:self.world * chunk]
        mine = full[self.rank * chunk:(self.rank + 1) * chunk]
        if self.staged and tiles.device.type == "cuda":
            recv = torch.empty(full.shape, dtype=full.dtype)
            dist.all_gather_into_tensor(recv, mine.cpu(), group=self.group)
            full.copy_(recv)
        else:
            dist.all_gather_into_tensor(full, mine, group=self.group)
        self.bytes_gathered += 4 * chunk * (self.world - 1)


def all_gather_host_rows(rows, group=None):
    """rows: this rank's (R_k, W) numpy array (R_k differs between ranks) -> the (sum R_k, W)
    concatenation in rank order on every rank.  Used to put the bit rows of a table the
    ranks parsed in byte ranges back together; goes over RCCL on GPU tensors when a GPU
    backend is up, gloo otherwise."""
    torch = _torch()
    import torch.distributed as dist
    world = dist.get_world_size(group)
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    as64 = np.ascontiguousarray(rows).view(np.int64) if rows.dtype == np.uint64 else np.ascontiguousarray(rows)
    n = torch.tensor([as64.shape[0]], dtype=torch.int64, device=dev)
    counts = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(counts, n, group=group)
    counts = counts.cpu().tolist()
    cap, W = max(counts + [1]), as64.shape[1]
    send = torch.zeros((cap, W), dtype=torch.from_numpy(as64[:0]).dtype, device=dev)
    send[:as64.shape[0]] = torch.from_numpy(as64).to(dev)
    recv = torch.empty((world * cap, W), dtype=send.dtype, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    recv = recv.view(world, cap, W).cpu().numpy()
    out = np.concatenate([recv[r, :counts[r]] for r in range(world)], axis=0)
    return (out.view(np.uint64) if rows.dtype == np.uint64 else out), counts


def all_gather_objects(obj, group=None):
    """[obj of rank 0, obj of rank 1, ...] on every rank (pickled; small host metadata)."""
    import torch.distributed as dist
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, obj, group=group)
    return out


def barrier(group=None):
    """Process-group barrier; no-op for a single process."""
    if is_distributed():
        import torch.distributed as dist
        dist.barrier(group=group)


def shutdown():
    """Leave the process group (if any) before the interpreter exits."""
    if _initialized():
        import torch.distributed as dist
        dist.destroy_process_group()


def _initialized():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def _initialized_safe():
    """_initialized() without importing torch when nobody has yet (a plain single-process run)."""
    import sys
    if "torch" not in sys.modules:
        return False
    return _initialized()


def associate_sharded(local_compute, G, group=None, kind=None):
    """Run ``local_compute(sel) -> int32 records [T, len(sel), REC_WORDS]`` on this rank's gene
    shard (``sel``: a slice of the gene axis, GenePartition.index) and gather the records of
    all ranks in gene order (every rank gets the full result: the host-side B/BH needs the
    globally sorted p).  ``kind``: "stride" (default; SCOARY_GENE_PARTITION overrides) or
    "contiguous"."""
    world, rank = world_rank()
    if world == 1 and not _initialized():
        return local_compute(slice(0, int(G), 1))
    part = GenePartition(G, world, kind or os.environ.get("SCOARY_GENE_PARTITION", "stride"))
    return all_gather_genes(local_compute(part.index(rank)), G, group, partition=part)


def numpy_records(rec):
    """Device (or host) records [T, G, REC_WORDS] -> the host arrays of unpack_records.  ONE copy
    of the record tensor, then numpy slices: slicing and cloning on the device would be the first
    torch elementwise kernels of a command-line process -- 0.2-0.4 s of lazy code-object loading
    for 8 MB of results (profiles/r05_e2e_cli_cfg4_vcf.txt 'results D2H')."""
    h = rec.contiguous().cpu().numpy()          # contiguous() of a contiguous tensor launches nothing
    T, G, _ = h.shape

    def f64(lo):
        return np.ascontiguousarray(h[:, :, lo:lo + 2]).view(np.float64).reshape(T, G)
    return {"counts": np.ascontiguousarray(h[:, :, 0:4]),
            "p": f64(4), "odds": f64(6),
            "r": np.ascontiguousarray(h[:, :, 8]).view(np.uint32),
            "nstop": np.ascontiguousarray(h[:, :, 9]).view(np.uint32)}
