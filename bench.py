#!/usr/bin/env python3
"""bench.py -- gene x permutation Fisher tests/s on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over the headline synthetic batch
(BASELINE.json configs[2]: 50k genes x 2000 isolates x 10 traits, 10k label
permutations): contingency counts -> Fisher p + rejection regions -> label
permutation generation -> permutation exceedance counts, inputs resident in
HBM.  tests per step = G*T*P per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Multi-GPU: genes shard across ranks (every rank holds its own G-gene shard of a
G*N-gene matrix: weak scaling); trait / permutation vectors are regenerated
identically on every rank from the seed (no broadcast); the one exchange step
of the path -- gathering per-gene results on rank 0 -- is an RCCL gather inside
the timed region (asynchronous, overlapped with the next step's kernels).

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LISTS_DEFAULT = True         # list-driven kernel: 4.9 ms vs 16.1 ms (dense) on the headline config
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
VALU_LANE_OPS_PER_S = 256 * 4 * 32 * 2.4e9   # 256 CU x 4 SIMD-32 x 2.4 GHz (upper bound)
# measured issue rates (lane-ops/s, whole chip): the dense kernel's op pair (v_and_b32 with an SGPR
# operand + v_bcnt_u32_b32 accumulate, tools/valu_peak.hip) and the list kernel's v_bitop3_b32 with
# VGPR operands in distinct banks (tools/valu_banks.hip: 2.5 cycles per wave-instruction at 8 waves
# per SIMD, 2.8 at the 4 the list kernel runs with)
VALU_PEAK_AND_BCNT = 4.1e13
VALU_PEAK_BITOP3 = 6.2e13


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg4"])
    ap.add_argument("--genes", type=int, default=None, help="override G (per GPU)")
    ap.add_argument("--permutations", type=int, default=None, help="override P")
    ap.add_argument("--isolates", type=int, default=None, help="override N (shape experiments)")
    ap.add_argument("--traits", type=int, default=None, help="override T (shape experiments)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exercise-exchange", action="store_true",
                    help="run the RCCL exchange step even at world size 1 (launched under "
                         "torch.distributed.run with one rank): a 1-GPU check of the N>1 code path")
    ap.add_argument("--kernel", default="auto", choices=["auto", "dense", "lists"],
                    help="permutation kernel: dense (k_permute_reg/chunked) or list-driven")
    ap.add_argument("--cpu-seconds", type=float, default=10.0,
                    help="target CPU time of the cpu_baseline sample")
    return ap.parse_args()


def load_traffic(config, default_sizes, kernel="k_permute"):
    """HBM bytes per k_permute launch from the committed PMC summary
    (profiles/r*_pmc.json, produced by tools/profile.sh + tools/rocpd_summary.py
    from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same
    command).  None when no summary matches the workload being run."""
    import glob
    if not default_sizes:
        return None, None, None
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json"))):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        if d.get("_meta", {}).get("workload") != config:
            continue
        for k, v in d.items():
            is_lists = k.startswith("k_permute_lists")
            if k.startswith("k_permute") and is_lists == (kernel == "k_permute_lists") \
                    and "hbm_traffic_bytes_per_launch" in v:
                best = (v["hbm_traffic_bytes_per_launch"], os.path.relpath(path, ROOT),
                        v.get("SQ_INSTS_VALU"))
    return best if best else (None, None, None)


def cpu_baseline(genes, traits, N, seed, target_s):
    """Time the CPU oracle (the C restatement, OpenMP over genes) on a bounded
    sample of the same workload: all T traits, a gene subsample, P_s
    permutations; whole path (counts + Fisher weights + permutations)."""
    from oracle import oracle as orc
    from scoary_amd.engine import pack_bits_rows
    cores = orc.num_threads()
    T = traits.shape[0]
    Gs = min(genes.shape[0], 4096)
    gb = orc.pack_rows(genes[:Gs])
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    orc.permute_r(gb, tb, mb, N, 64, seed)             # warm the thread pool / caches
    t0 = time.perf_counter()
    orc.permute_r(gb, tb, mb, N, 1024, seed)
    probe = time.perf_counter() - t0
    rate = Gs * T * 1024 / probe
    Ps = int(max(64, min(400000, target_s * rate / (Gs * T))))
    t0 = time.perf_counter()
    orc.permute_r(gb, tb, mb, N, Ps, seed)
    dt = time.perf_counter() - t0
    return {"value": Gs * T * Ps / dt, "unit": "gene-permutation Fisher tests/s",
            "cores": cores, "kind": "port",
            "sample": "oracle/oracle.c orc_permute_r (counts + Fisher weights + label "
                      "permutations + exceedance), %d genes x %d isolates x %d traits x %d "
                      "permutations, %d OpenMP threads, %.1f s" % (Gs, N, T, Ps, cores, dt)}


class Exchange:
    """The path's one exchange step: the per-gene records of every shard are
    gathered on rank 0 over RCCL/xGMI (north_star: "only an RCCL gather of
    per-gene results").  It is issued asynchronously and drained before its
    buffers are reused, so step i's gather overlaps step i+1's kernels; every
    gather has completed before the closing barrier of the timed region."""

    def __init__(self, torch, eng, world, rank, T, G):
        from scoary_amd import dist as sdist
        self.sdist, self.world, self.rank, self.G = sdist, world, rank, G
        self.pending, self.step_no, self.kind = [], 0, "gather"
        self.recv = [None, None]
        if rank == 0:
            self.recv = [torch.empty((world, T, G, sdist.REC_WORDS), dtype=torch.int32,
                                     device=eng.device) for _ in range(2)]

    def drain(self, keep=0):
        while len(self.pending) > keep:
            self.pending.pop(0)()

    def submit(self, res):
        sdist = self.sdist
        self.drain(keep=1)
        rec = sdist.pack_records(res["counts"], res["p"], res["odds"], res["r"])
        if self.kind == "gather":
            try:
                _, finish = sdist.gather_genes(rec, self.G * self.world, dst=0, async_op=True,
                                               recv=self.recv[self.step_no % 2])
                self.pending.append(finish)
            except (RuntimeError, NotImplementedError) as e:   # backend without gather
                if self.rank == 0:
                    print("bench: dist.gather unavailable (%s); using all_gather" % e,
                          file=sys.stderr)
                self.kind = "all_gather"
        if self.kind == "all_gather":
            res["gathered"] = sdist.all_gather_genes(rec, self.G * self.world)
        self.step_no += 1


def measured_copy_peak(device):
    """Device-to-device copy rate (read + write bytes per second) of a 1 GiB buffer: the
    measured HBM figure SURVEY 8d asks for next to the 8 TB/s datasheet peak."""
    import torch
    n = 1 << 30
    src = torch.empty(n, dtype=torch.uint8, device=device)
    dst = torch.empty_like(src)
    src.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(3):
        dst.copy_(src)
    e0.record()
    reps = 10
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    e1.synchronize()
    return 2.0 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def roofline_report(args, eng, use_lists, G, N, T, P, pbatch, k3_name, k3_ms):
    """The `roofline` object: operand-bandwidth model, measured HBM
    traffic and VALU instruction count from the committed PMC passes."""
    W64 = (N + 63) // 64
    tests_per_launch = G * T * (P if use_lists else min(P, pbatch))
    launches_per_step = 1 if use_lists else -(-P // pbatch)
    alg_bytes = 16.0 * W64 * tests_per_launch        # 16*W bytes / test
    achieved = alg_bytes / (k3_ms * 1e-3) / 1e9
    default_sizes = (args.genes is None and args.permutations is None
                     and args.isolates is None and args.traits is None)
    traffic, traffic_src, valu_insts = load_traffic(args.config, default_sizes, k3_name)
    w32 = -(-N // 32)
    valu_ops = tests_per_launch * (2.0 * w32 + 6)     # dense-kernel op model (reference point)
    copy_gbs = measured_copy_peak(eng.device)
    return {
        "bound": "hbm",
        "kernel": k3_name,
        "achieved": achieved,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic,
        "traffic_unit": "bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024",
        "traffic_source": traffic_src,
        "traffic_gbs": None if not traffic else traffic / (k3_ms * 1e-3) / 1e9,
        "measured_copy_peak_gbs": copy_gbs,            # torch D2D copy of 1 GiB, read + write
        "frac_of_measured_copy": achieved / copy_gbs,
        "algorithmic_bytes_per_launch": alg_bytes,
        "model": "operand bytes 16*ceil(N/64) B per test (SURVEY 8d); frac > 1 means the "
                 "kernel left the HBM-bound regime (operands reused from VGPR/SGPR)",
        "kernel_ms": k3_ms,
        "launches_per_step": launches_per_step,
        "tests_per_s_kernel": tests_per_launch / (k3_ms * 1e-3),
        "dense_model_valu_frac_of_2.4GHz_simd32_peak":
            None if use_lists else valu_ops / (k3_ms * 1e-3) / VALU_LANE_OPS_PER_S,
        # what actually binds: VALU instruction issue (figure iii).  Instruction
        # count from the committed SQ_INSTS_VALU pass, duration measured live; the ceiling
        # is the measured chip-wide issue rate of the kernel's own op (see the constants).
        "valu": None if not valu_insts else {
            "wave_insts_per_launch": valu_insts,
            "ops_per_test": valu_insts * 64.0 / tests_per_launch,
            "lane_ops_per_s": valu_insts * 64.0 / (k3_ms * 1e-3),
            "peak_lane_ops_per_s": VALU_PEAK_BITOP3 if use_lists else VALU_PEAK_AND_BCNT,
            "peak_source": "tools/valu_banks.hip (v_bitop3_b32, 8 waves/SIMD)" if use_lists
                           else "tools/valu_peak.hip (v_and_b32 + v_bcnt_u32_b32)",
            "frac": valu_insts * 64.0 / (k3_ms * 1e-3)
                    / (VALU_PEAK_BITOP3 if use_lists else VALU_PEAK_AND_BCNT),
            "source": traffic_src,
            "clock_note": "the list kernel runs at the socket power cap: 1.37 kW, shader clock "
                          "2.15 of 2.4 GHz (profiles/r01_clock_power.txt)" if use_lists else None},
    }


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sharded = world > 1 or (args.exercise_exchange and "RANK" in os.environ)
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); none visible")
    torch.cuda.set_device(local_rank)
    if sharded:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from scoary_amd import synth
    from scoary_amd.engine import AssociationEngine, pack_bits_rows

    # every rank: its own gene shard (different seed offset), same traits
    genes, traits, P, seed = synth.make_config(args.config, G=args.genes, N=args.isolates,
                                               T=args.traits)
    if rank > 0:
        rng = np.random.default_rng(seed + 1000 * rank)
        genes = synth.make_genes(genes.shape[0], genes.shape[1], rng,
                                 kind="rare" if args.config == "cfg4" else "uniform",
                                 core_frac=0.05 if args.config == "cfg3" else 0.0)
    if args.permutations:
        P = args.permutations
    G, N = genes.shape
    T = traits.shape[0]

    eng = AssociationEngine(local_rank)
    gm = eng.pack_dense(genes)                     # bit-packed once into HBM
    trv = eng.vecrows(pack_bits_rows((traits == 1).astype(np.uint8)), N)
    mkv = eng.vecrows(pack_bits_rows((traits != 2).astype(np.uint8)), N)
    use_lists = args.kernel == "lists" or (args.kernel == "auto" and LISTS_DEFAULT
                                           and eng.lists_supported(N))
    if use_lists:
        eng.build_lists(gm)    # once per dataset, like the packing
    pbatch = eng.perm_batch(T, N, P)
    perm_buf = torch.empty((T, pbatch, eng.row_words(N)), dtype=torch.int32, device=eng.device)
    exchange = Exchange(torch, eng, world, rank, T, G) if sharded else None

    def step():
        res = eng.associate(gm, trv, mkv, permutations=P, seed=seed, perm_buffer=perm_buf,
                            use_lists=use_lists)
        if exchange:
            exchange.submit(res)
        return res

    def barrier():
        if exchange:
            exchange.drain()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    eng.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    k3_name = "k_permute_lists" if use_lists else "k_permute"
    k3_ms = eng.kernel_ms(k3_name)
    names = ("k_margins", "k_counts", "k_fisher") + (
        ("k_perm_generate_tiles", "k_lists_crit", "k_permute_lists") if use_lists else
        ("k_perm_generate", "k_permute"))
    kernel_ms = {k: eng.kernel_ms(k) for k in names}
    eng.set_timing(False)

    if sharded:                                      # MAX over ranks
        tmax = torch.tensor([dt], dtype=torch.float64, device=eng.device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        out = {
            "metric": "gene x permutation Fisher tests/sec",
            "value": G * T * P * world * args.steps / dt,
            "unit": "tests/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 bit-words (bit-sliced adders / AND + popcount), f64 for Fisher p",
            "data": "synthetic",
            "config": {"workload": "%s: %d genes x %d isolates x %d traits, --permute %d per GPU; "
                                   "counts + Fisher + label permutations + exceedance counts"
                                   % (args.config, G, N, T, P),
                       "genes_per_gpu": G, "isolates": N, "traits": T, "permutations": P,
                       "parallelism": "gene-shard x%d" % world,
                       "exchange": ("rccl %s of per-gene records" % exchange.kind) if exchange
                       else "none (single GPU)"},
            "roofline": roofline_report(args, eng, use_lists, G, N, T, P, pbatch, k3_name, k3_ms),
            "kernel_ms": kernel_ms,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(genes, traits, N, seed, args.cpu_seconds)
        print(json.dumps(out))
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
