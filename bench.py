#!/usr/bin/env python3
"""bench.py -- gene x permutation Fisher tests/s on MI355X (BASELINE.json metric).

One "step" = one pass of the whole hot path over the headline synthetic batch
(BASELINE.json configs[2]: 50k genes x 2000 isolates x 10 traits, 10k label
permutations): contingency counts -> Fisher p + rejection regions -> label
permutation generation -> permutation exceedance counts, inputs resident in
HBM.  tests per step = G*T*P.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config cfg3] [--scaling weak|strong]

``--gpus N`` with N > 1 runs N ranks by itself (re-executes under
``torch.distributed.run --nproc-per-node N``) unless it is already running under a
launcher (WORLD_SIZE set), and refuses to run with fewer than N visible GPUs.

Multi-GPU: genes shard across ranks -- weak scaling: every rank holds its own G-gene
shard (a G*N-gene matrix); strong scaling: the config's G genes are split into the
reference's stride domains, rank r = genes r, r + N, r + 2N, ... (scoary_amd.dist.GenePartition;
scoary/methods.py:1076-1078).  Trait / permutation vectors are regenerated identically on every
rank from the seed (no broadcast); the one exchange step of the path -- gathering per-gene
results on rank 0 -- is an RCCL gather inside the timed region (asynchronous, overlapped with
the next step's kernels), and rank 0 checks the records it received from every rank.

The default line is the WEAK one (`value`, `scaling`); next to it `scaling_strong` carries, timed
in the same process group, the strong split of the headline config (cfg3's 50 000 genes over the
N ranks) and of cfg4 (200 000 variants over the N ranks), so the 1/2/4/8 runs give the strong
curve SURVEY 8e asks for as well (at N = 1 they are the denominators).

Prints ONE JSON line (rank 0).
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LISTS_DEFAULT = True         # list-driven kernel: 4.6 ms vs 16.1 ms (dense) on the headline config
HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# nominal integer-VALU peak: 256 CU x 4 SIMD x 32 lanes/clk x 2.4 GHz (a wave64 op = 2 cycles)
VALU_NOMINAL_LANE_OPS = 256 * 4 * 32 * 2.4e9
# measured issue rates (lane-ops/s, whole chip): the dense kernel's op pair (v_and_b32 with an SGPR
# operand + v_bcnt_u32_b32 accumulate, tools/valu_peak.hip) and the list kernel's v_bitop3_b32 with
# VGPR operands in distinct banks (tools/valu_banks.hip: 2.5 cycles per wave-instruction at 8 waves
# per SIMD, 2.8 at the 4 the list kernel runs with)
VALU_PEAK_AND_BCNT = 4.1e13
# sustained v_bitop3_b32, conflict-free banks, no memory traffic (profiles/r02_clock_valu_both_occupancies.txt):
# 5.61e13 at the 4 waves per SIMD the list kernel can have (127 VGPRs, one 1024-thread block per CU),
# 6.06e13 at 8 waves per SIMD (out of its reach: <= 64 VGPRs)
VALU_PEAK_BITOP3_4WAVES = 5.61e13
VALU_PEAK_BITOP3_8WAVES = 6.06e13
KERNEL_SOURCES = ("scoary_lists.hip", "scoary_list_walk.inc", "scoary_assoc.hip", "scoary_labels.hip", "scoary_common.hpp",
                  "scoary_ctr_regs.inc", "scoary_vgpr_banks.inc")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg3", choices=["cfg2", "cfg3", "cfg4", "cfg5"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank its own G-gene shard; strong: the config's G genes "
                         "split across the ranks")
    ap.add_argument("--partition", default="stride", choices=["stride", "contiguous"],
                    help="strong scaling: which genes a rank owns -- stride (rank r: genes r, r + N, ...: the "
                         "reference's domains, balanced for any gene order) or contiguous equal-count blocks "
                         "(rounds 1-5; unbalanced on frequency-sorted tables, kept as the A/B)")
    ap.add_argument("--gene-order", default="config", choices=["config", "sorted"],
                    help="sorted: the matrix rows in descending gene frequency, the order Roary writes its "
                         "table in (scoary/exampledata/Gene_presence_absence.csv: 100, ..., 0 isolates)")
    ap.add_argument("--strong-extra", default="auto", choices=["auto", "on", "off"],
                    help="also time the strong split of cfg3 and cfg4 over the ranks of this run "
                         "(`scaling_strong`); auto: on for the default cfg3 weak line")
    ap.add_argument("--genes", type=int, default=None, help="override G (per GPU / total)")
    ap.add_argument("--permutations", type=int, default=None, help="override P")
    ap.add_argument("--isolates", type=int, default=None, help="override N (shape experiments)")
    ap.add_argument("--traits", type=int, default=None, help="override T (shape experiments)")
    ap.add_argument("--gene-kind", default=None, choices=["uniform", "rare", "ushaped", "balanced"],
                    help="gene-frequency spectrum instead of the config's own (evidence lines next to the "
                         "BASELINE configs: ushaped = Beta(0.15, 0.15), a pan-genome-like spectrum)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exercise-exchange", action="store_true",
                    help="run the RCCL exchange step even at world size 1 (launched under "
                         "torch.distributed.run with one rank): a 1-GPU check of the N>1 code path")
    ap.add_argument("--kernel", default="auto", choices=["auto", "dense", "lists"],
                    help="permutation kernel: dense (k_permute_reg/chunked) or list-driven")
    ap.add_argument("--graph", action="store_true",
                    help="capture the step into a hipGraph and replay it; per-kernel times then come from "
                         "extra eager steps after the timed region.  Launch-bound workloads (fewer than 5e8 "
                         "tests per step: cfg2) do this by default, as the engine itself does "
                         "(AssociationEngine.auto_graph_eligible)")
    ap.add_argument("--no-graph", action="store_true", help="never replay a hipGraph (eager launches)")
    ap.add_argument("--cpu-seconds", type=float, default=6.0,
                    help="target wall time of each cpu_baseline sample")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend of a multi-rank run: nccl (= RCCL over xGMI, the real "
                         "thing) or gloo (host-staged; lets several ranks share ONE GPU with --share-gpu, "
                         "a functional check of the multi-rank path on a single-GPU box)")
    ap.add_argument("--share-gpu", action="store_true",
                    help="map rank r to device r %% visible devices instead of requiring one GPU per "
                         "rank (functional check only: the ranks time-share the device)")
    ap.add_argument("--verify-gather", action="store_true",
                    help="after the timed region rank 0 recomputes every rank's shard by itself and "
                         "compares the gathered records with it bit for bit (gather_matches_single_rank)")
    ap.add_argument("--telemetry-ms", type=float, default=4.0,
                    help="sampling period of the shader-clock / socket-power side thread during the "
                         "timed region (amdsmi or hwmon); 0 switches it off")
    ap.add_argument("--inject-gather-fault", action="store_true", help=argparse.SUPPRESS)   # tests: corrupt one
    # received record before --verify-gather looks at it; every rank must then exit non-zero, together
    ap.add_argument("--label-shards", action="store_true",
                    help="multi-GPU: every rank generates 1/world of each batch of label tiles and one "
                         "all_gather_into_tensor supplies the rest (dist.LabelShards); default: every rank "
                         "generates all tiles (spec S4's generator costs less than the collective's latency "
                         "on the BASELINE shapes)")
    ap.add_argument("--k1-cold", action="store_true", help=argparse.SUPPRESS)     # the default since round 5
    ap.add_argument("--no-k1-cold", action="store_true",
                    help="skip roofline_k1.cold: k_counts timed on a working set the 256 MiB Infinity Cache cannot "
                         "hold (cfg5's 125 000 x 10 000 shard shape with T = 1 and 4, eight 160 MB matrices in "
                         "rotation; single GPU only, ~1 s and 1.3 GB)")
    ap.add_argument("--sustain-seconds", type=float, default=6.0,
                    help="after the timed region (single GPU): run the step back to back this long and report "
                         "the clock / power the box sustains (the `sustained` object); 0 skips it")
    ap.add_argument("--dry-exchange", action="store_true",
                    help="CPU-only check of the launcher + exchange path (gloo, fabricated "
                         "records, no kernels): what tests/test_dist_gloo.py drives")
    return ap.parse_args(argv)


# --------------------------------------------------------------- launcher -----
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def maybe_relaunch(args):
    """--gpus N > 1 outside a launcher: run N ranks ourselves (one process per GPU)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    if not args.dry_exchange:
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < (1 if args.share_gpu else args.gpus):
            raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible on this node" % (args.gpus, have))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    raise SystemExit(subprocess.call(cmd, env=env))


# ------------------------------------------------------- counters on file -----
def kernel_source_sha():
    h = hashlib.sha256()
    for fn in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "scoary_amd", "csrc", fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def load_counters(config, default_sizes, kernel):
    """PMC counters of the dominant kernel from the committed rocprofv3 summary
    (profiles/r*_pmc.json: separate --pmc passes of this same command, tools/profile.sh +
    tools/rocpd_summary.py).  Only a summary whose _meta.kernel_source_sha256 equals the
    hash of the kernel sources in this tree is used -- counters of another kernel
    version are refused, not reported.  Returns (dict | None, reason)."""
    import glob
    if not default_sizes:
        return None, "shape overridden on the command line: no committed PMC profile applies"
    sha = kernel_source_sha()
    stale = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")), reverse=True):
        try:
            with open(path) as f:
                d = json.load(f)
        except (OSError, ValueError):
            continue
        meta = d.get("_meta", {})
        if meta.get("workload") != config:
            continue
        for k, v in d.items():
            if k.startswith(kernel) and (kernel != "k_permute" or not k.startswith("k_permute_lists")) \
                    and isinstance(v, dict) and "SQ_INSTS_VALU" in v:
                if meta.get("kernel_source_sha256") != sha:
                    stale = os.path.relpath(path, ROOT)
                    continue
                out = dict(v)
                out["source"] = os.path.relpath(path, ROOT)
                return out, None
    if stale:
        return None, ("%s was collected from other kernel sources (sha256 mismatch): re-run "
                      "tools/profile.sh" % stale)
    return None, "no PMC summary for %s / %s under profiles/" % (config, kernel)


# ------------------------------------------------------------- telemetry -----
class Telemetry:
    """Shader clock and socket power of one GPU, sampled from a side thread WHILE the timed
    region runs (the same kernel sources gave 9.5e11 on the driver's box
    and 1.04e12 on the builder's -- the list kernel sits at the socket power cap, so its time
    follows the clock the box grants).  With the mean clock of THIS run in the line,
    `ops_per_clock` (VALU lane-ops per shader clock) separates a slow box from a slow kernel:
    it is a property of the code, the clock is a property of the box.
    Sources, first that works: amdsmi (gpu_metrics: current_gfxclk(s), current_socket_power),
    then the amdgpu hwmon files (freq1_input, power1_average / power1_input)."""

    def __init__(self, device_index=0, period_s=0.004, reader=None):
        self.period_s = period_s
        self.samples = []                   # (t, sclk_mhz | None, power_w | None)
        self.source = None
        self._stop = None
        self._thread = None
        self._read = reader or self._pick_reader(device_index)
        if reader is not None:
            self.source = "injected"

    # -- sources -----------------------------------------------------------
    def _pick_reader(self, device_index):
        for make in (self._amdsmi_reader, self._hwmon_reader):
            try:
                rd = make(device_index)
                if rd is not None and any(v is not None for v in rd()):
                    return rd
            except Exception:               # a missing tool must never cost the bench line
                continue
        self.source = None
        return None

    @staticmethod
    def _bdf_of_torch_device(device_index):
        try:
            import torch
            pr = torch.cuda.get_device_properties(device_index)
            return "%04x:%02x:%02x" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        except Exception:
            return None

    def _amdsmi_reader(self, device_index):
        import amdsmi
        try:
            amdsmi.amdsmi_init()
        except Exception:
            return None
        handles = amdsmi.amdsmi_get_processor_handles()
        if not handles:
            return None
        h = handles[min(device_index, len(handles) - 1)]
        want = self._bdf_of_torch_device(device_index)
        if want:
            for cand in handles:
                try:
                    if str(amdsmi.amdsmi_get_gpu_device_bdf(cand)).lower().startswith(want):
                        h = cand
                        break
                except Exception:
                    pass

        def num(x):
            return float(x) if isinstance(x, (int, float)) and x >= 0 else None

        def read():
            m = amdsmi.amdsmi_get_gpu_metrics_info(h)
            clk = None
            per_xcd = [c for c in (m.get("current_gfxclks") or []) if isinstance(c, (int, float)) and c > 0] \
                if isinstance(m.get("current_gfxclks"), (list, tuple)) else []
            if per_xcd:
                clk = float(sum(per_xcd)) / len(per_xcd)
            if clk is None:
                clk = num(m.get("current_gfxclk"))
            pw = num(m.get("current_socket_power"))
            if pw is None:
                pw = num(m.get("average_socket_power"))
            return clk, pw
        self.source = "amdsmi gpu_metrics (current_gfxclks mean over XCDs, current_socket_power)"
        return read

    def _hwmon_reader(self, device_index):
        import glob
        cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device/hwmon/hwmon*"))
        if not cards:
            return None
        want = self._bdf_of_torch_device(device_index)
        hw = cards[min(device_index, len(cards) - 1)]
        if want:
            for c in cards:
                if want in os.path.realpath(os.path.join(c, "..", "..")).lower():
                    hw = c
                    break
        fclk = os.path.join(hw, "freq1_input")
        fpow = next((os.path.join(hw, n) for n in ("power1_average", "power1_input")
                     if os.path.exists(os.path.join(hw, n))), None)

        def rd(path, scale):
            try:
                with open(path) as f:
                    return float(f.read().strip()) * scale
            except (OSError, ValueError, TypeError):
                return None

        def read():
            return (rd(fclk, 1e-6) if os.path.exists(fclk) else None,
                    rd(fpow, 1e-6) if fpow else None)
        self.source = "amdgpu hwmon (%s: freq1_input, %s)" % (hw, os.path.basename(fpow) if fpow else "-")
        return read

    # -- sampling ------------------------------------------------------------
    def start(self):
        if self._read is None:
            return self
        import threading
        self.samples = []
        self._stop = threading.Event()

        def loop():
            while not self._stop.is_set():
                try:
                    clk, pw = self._read()
                except Exception:
                    clk = pw = None
                self.samples.append((time.perf_counter(), clk, pw))
                self._stop.wait(self.period_s)
        self._thread = threading.Thread(target=loop, name="scoary-telemetry", daemon=True)
        self._thread.start()
        return self

    def stop(self, t_begin=None, t_end=None):
        """Ends the sampling; summary of the samples taken inside [t_begin, t_end]
        (perf_counter times of the timed region; all samples if not given)."""
        if self._thread is not None:
            self._stop.set()
            self._thread.join(timeout=2.0)
            self._thread = None
        rows = [s for s in self.samples
                if (t_begin is None or s[0] >= t_begin) and (t_end is None or s[0] <= t_end)]
        clk = [c for _, c, _ in rows if c]
        pw = [w for _, _, w in rows if w]

        def stat(v):
            return (None, None, None) if not v else (sum(v) / len(v), min(v), max(v))
        cm, cmin, cmax = stat(clk)
        pm, pmin, pmax = stat(pw)
        return {"source": self.source, "samples": len(rows), "period_ms": self.period_s * 1e3,
                "sclk_mhz_mean": cm, "sclk_mhz_min": cmin, "sclk_mhz_max": cmax,
                "socket_power_w_mean": pm, "socket_power_w_min": pmin, "socket_power_w_max": pmax}


# ------------------------------------------------------------ CPU baselines ---
def cpu_baseline_port(genes, traits, N, seed, target_s):
    """The C restatement (oracle/oracle.c, OpenMP over genes) on a bounded sample of the
    same workload: all T traits, a gene subsample, P_s permutations; whole path."""
    from oracle import oracle as orc
    from oracle.scipy_baseline import usable_cpus
    from scoary_amd.engine import pack_bits_rows
    # OpenMP would start one thread per logical CPU of the HOST (256); the container's cgroup
    # grants far fewer (16 on the pool's boxes): oversubscribed threads only add switching
    orc.set_num_threads(min(orc.num_threads(), usable_cpus()))
    cores = orc.num_threads()
    T = traits.shape[0]
    Gs = min(genes.shape[0], 4096)
    gb = orc.pack_rows(genes[:Gs])
    tb = pack_bits_rows((traits == 1).astype(np.uint8))
    mb = pack_bits_rows((traits != 2).astype(np.uint8))
    orc.permute_r(gb, tb, mb, N, 64, seed)             # warm the thread pool / caches
    t0 = time.perf_counter()
    orc.permute_r(gb, tb, mb, N, 4096, seed)
    probe = time.perf_counter() - t0
    rate = Gs * T * 4096 / probe
    Ps = int(max(64, min(400000, target_s * rate / (Gs * T))))
    t0 = time.perf_counter()
    orc.permute_r(gb, tb, mb, N, Ps, seed)
    dt = time.perf_counter() - t0
    return {"value": Gs * T * Ps / dt, "unit": "gene-permutation Fisher tests/s",
            "cores": cores, "cores_are": "OpenMP threads = the logical CPUs this process may use (affinity mask and "
                                         "cgroup quota applied to os.cpu_count(); SMT siblings included)",
            "kind": "port",
            "sample": "oracle/oracle.c orc_permute_r (bit-packed popcount counts + Fisher weights + "
                      "label permutations + exceedance), %d genes x %d isolates x %d traits x %d "
                      "permutations, %d OpenMP threads, %.1f s" % (Gs, N, T, Ps, cores, dt)}


def cpu_baseline_scipy(eng, genes, traits, N, seed, target_s):
    """The reference-structured CPU path (oracle/scipy_baseline.py: per-isolate Python
    counting + memoised scipy.stats.fisher_exact over a multiprocessing.Pool of
    os.cpu_count() stride domains; scoary/methods.py:791-857, :1076-1078) on a gene
    subsample x trait 0 x 20 permutations, CHECKED against the GPU on the same sample and
    extrapolated linearly to tests/s."""
    from oracle import oracle as orc
    from scoary_amd.engine import pack_bits_rows
    Ps = 20
    tr = traits[0]
    mb = pack_bits_rows((tr != 2).astype(np.uint8)[None])
    npos = int((tr == 1).sum())
    labels = np.stack([np.unpackbits(orc.perm_labels(seed, 0, pi, mb[0], npos, N).view(np.uint8),
                                     bitorder="little")[:N] for pi in range(Ps)])
    G = genes.shape[0]
    # candidates: <= 16384 genes spread over the matrix; the baseline process sizes its sample
    # from them (4-gene probe -> about target_s seconds on all cores).  It runs in a fresh
    # interpreter: a multiprocessing Pool must not be forked from this process (HIP context,
    # runtime threads).
    import tempfile
    cand = np.unique(np.linspace(0, G - 1, min(G, 16384)).astype(np.int64))
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.npz"), os.path.join(tmp, "out.npz")
        np.savez(fin, genes=genes[cand], trait=tr, labels=labels)
        env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
        subprocess.run([sys.executable, "-m", "oracle.scipy_baseline", fin, fout, str(target_s)],
                       check=True, cwd=ROOT, env=env, timeout=600)
        d = np.load(fout)
        sub = cand[d["order"]]
        counts, p, r, dt, n = d["counts"], d["p"], d["r"], float(d["dt"]), int(d["n"])
        cpu_count, usable = int(d["cpu_count"]), int(d["usable"])
    Gs = len(sub)
    # the GPU on the same sample, same labels (spec S4, same seed, trait 0)
    gm = eng.pack_dense(genes[sub])
    tb = pack_bits_rows((tr == 1).astype(np.uint8)[None])
    res = eng.associate(gm, eng.vecrows(tb, N), eng.vecrows(mb, N), permutations=Ps, seed=seed,
                        use_lists=False)
    gc = res["counts"].cpu().numpy()[0]
    gp = res["p"].cpu().numpy()[0]
    gr = res["r"].cpu().numpy().view(np.uint32)[0]
    ok = bool(np.array_equal(gc, counts) and np.array_equal(gr.astype(np.int64), r)
              and np.max(np.abs(gp - p)) < 1e-12)
    tests = Gs * (Ps + 1)
    return {"value": tests / dt, "unit": "gene-permutation Fisher tests/s", "cores": n,
            "cores_are": "Pool workers = the logical CPUs this process may use (os.cpu_count() = %d, after "
                         "affinity mask and cgroup quota: %d; SMT siblings included)" % (cpu_count, usable),
            "per_worker_tests_per_s": tests / dt / max(n, 1),
            "genes_per_worker": Gs / max(n, 1),
            "kind": "scipy-restatement", "matches_gpu": ok,
            "sample": "oracle/scipy_baseline.py: per-isolate Python counting + memoised "
                      "scipy.stats.fisher_exact, multiprocessing.Pool(%d) over stride domains "
                      "range(k, G, %d) (reference structure, scoary/methods.py:791-857, :1076-1078); "
                      "%d genes x %d isolates x trait 0 x (1 + %d permutations) = %d tests in %.2f s "
                      "(pool start-up excluded; %.1f genes and %.0f tests/s per worker -- a single core "
                      "alone does ~350 tests/s at N = 2000, so a lower figure is workers sharing cores "
                      "or a short, skewed sample), extrapolated linearly; counts / r equal and "
                      "|dp| < 1e-12 against the GPU on this sample: %s"
                      % (n, n, Gs, N, Ps, tests, dt, Gs / max(n, 1), tests / dt / max(n, 1), ok)}


# ------------------------------------------------------------------ exchange --
class Exchange:
    """The path's one exchange step: the per-gene records of every shard are
    gathered on rank 0 over RCCL/xGMI (north_star: "only an RCCL gather of
    per-gene results").  It is issued asynchronously and drained before its
    buffers are reused, so step i's gather overlaps step i+1's kernels; every
    gather has completed before the closing barrier of the timed region.
    ``partition``: scoary_amd.dist.GenePartition of ALL the genes of the run (weak
    scaling: world contiguous blocks of G; strong: the config's genes, stride)."""

    def __init__(self, torch, eng, world, rank, T, partition):
        from scoary_amd import dist as sdist
        self.sdist, self.world, self.rank, self.part = sdist, world, rank, partition
        self.eng = eng if hasattr(eng, "lib") else None
        self.total = partition.G
        self.pending, self.step_no, self.kind = [], 0, "gather"
        self.recv = [None, None]
        if rank == 0:
            self.recv = [torch.empty((world, T, partition.cap, sdist.REC_WORDS), dtype=torch.int32,
                                     device=eng.device) for _ in range(2)]

    def drain(self, keep=0):
        while len(self.pending) > keep:
            self.pending.pop(0)()

    def submit(self, res):
        sdist = self.sdist
        self.drain(keep=1)
        if res.get("records") is not None:                # packed inside the (replayed) step
            rec = res["records"]
        elif self.eng is not None and hasattr(self.eng, "pack_records"):
            rec = self.eng.pack_records(res)              # one kernel (scoary_pack_records)
        else:                                             # CPU tensors (gloo tests)
            rec = sdist.pack_records(res["counts"], res["p"], res["odds"], res["r"])
        if self.kind == "gather":
            try:
                # the blocks stay as they arrive ([rank][T][cap][10], GenePartition's layout): gene order
                # is the READER's indexing (GenePartition.weave -- --verify-gather applies it to compare
                # with a one-GPU run; the command line applies it once, on the host), as the reference
                # weaves its workers' lists when it writes the results (scoary/methods.py:1115-1122)
                _, finish = sdist.gather_genes(rec, self.total, dst=0, async_op=True,
                                               recv=self.recv[self.step_no % 2], partition=self.part, weave=False)
                self.pending.append(finish)
            except (RuntimeError, NotImplementedError) as e:   # backend without gather
                if self.rank == 0:
                    print("bench: dist.gather unavailable (%s); using all_gather" % e,
                          file=sys.stderr)
                self.kind = "all_gather"
        if self.kind == "all_gather":
            res["gathered"] = sdist.all_gather_genes(rec, self.total, partition=self.part)
        self.step_no += 1

    def last_block(self):
        """rank 0: the receive buffer of the last step, [world, T, cap, words]."""
        return self.recv[(self.step_no - 1) % 2]

    def check(self, T, nval):
        """Rank 0, after the last drain: the receive buffer of the last step holds one
        block per rank, every record a plausible result of THIS workload -- the four
        counts of every (trait, gene) add up to the trait's valid isolates (a
        size-independent property only a real record satisfies).  Returns the number of
        ranks whose block passed."""
        if self.rank != 0 or self.kind != "gather":
            return None
        last = self.last_block()
        ok = 0
        for rk, n in enumerate(self.part.lengths()):
            counts = last[rk, :, :n, 0:4].sum(dim=2).cpu().numpy()     # [T, shard]
            if counts.shape == (T, n) and np.array_equal(
                    counts, np.broadcast_to(np.asarray(nval)[:, None], counts.shape)):
                ok += 1
        return ok


def order_genes(genes, how):
    """--gene-order sorted: rows in descending gene frequency (stable), Roary's table order."""
    if how != "sorted":
        return genes
    ones = genes.sum(axis=1, dtype=np.int64)
    return np.ascontiguousarray(genes[np.argsort(-ones, kind="stable")])


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


def list_adder_work(eng, gm, genes, T, P, batch):
    """Lane-ops of one k_permute_lists launch that are the algorithm itself: the bit-sliced
    full adders.  A list entry adds one tile-row word into the counter planes of every
    permutation word: 32 entries -> 31 full adders per word, a full adder = 2 v_bitop3_b32
    (sum 0x96, carry 0xE8).  A wave group of GPW genes walks its lists padded to the group's
    common length (nhalf half-steps of 16 entries, scoary_lists_plan), 64 lanes x NW words:
        adder_lane_ops = sum_groups nhalf*16 * 64*NW * (31/32)*2  x  T * tiles per launch
    `unpadded` takes the genes' true minority counts instead (padding is not useful either).
    Everything else the kernel issues -- address adds, ripple, region test, epilogue, index and
    tile loads -- is overhead."""
    tw, _stride, gpw, _classes, _piece = eng.list_params(gm.N)
    nw = tw // (64 // gpw)
    nhalf = gm.lists.ngroups.cpu().numpy().astype(np.int64).reshape(-1, gm.G)[:, ::gpw]   # per (segment,) wave group
    tile_perms = 32 * tw
    launches = -(-P // batch)
    tiles_per_launch = sum(-(-min(batch, P - b) // tile_perms) for b in range(0, P, batch)) / launches
    per_tile_padded = float(nhalf.sum()) * 16 * 64 * nw * (31.0 / 32.0) * 2.0
    ones = genes.sum(axis=1, dtype=np.int64)
    minority = np.minimum(ones, gm.N - ones)
    per_tile_true = float(minority.sum()) * tw * (31.0 / 32.0) * 2.0
    return {"padded": per_tile_padded * T * tiles_per_launch,
            "unpadded": per_tile_true * T * tiles_per_launch,
            "padded_entries_per_gene": float(nhalf.sum()) * 16 * gpw / max(gm.G, 1),
            "minority_entries_per_gene": float(minority.mean())}


def roofline_report(args, eng, use_lists, G, N, T, P, pbatch, k3_name, k3_ms, adders=None, telemetry=None):
    """The `roofline` object.  What binds the permutation kernel is integer-VALU issue,
    not HBM, so: bound = "valu", achieved = lane-ops/s from the
    SQ_INSTS_VALU count of the committed PMC pass of THIS kernel version (sha-checked)
    and the live hipEvent duration, peak = the nominal SIMD peak (frac <= 1 by
    construction).  `frac` is a UTILISATION (every overhead instruction raises it);
    `useful_frac` counts only the full-adder lane-ops the algorithm needs (list_adder_work)
    against the same peak.  The measured v_bitop3 ceilings, the SURVEY 8d
    operand-bandwidth figure and the measured HBM traffic ride along."""
    W64 = (N + 63) // 64
    launches_per_step = -(-P // pbatch)                # label tiles / label rows come in batches
    tests_per_launch = G * T * P / launches_per_step   # average over the launches of a step
    operand_bytes = 16.0 * W64 * tests_per_launch        # 16*W bytes / test
    default_sizes = (args.genes is None and args.permutations is None and args.isolates is None
                     and args.traits is None and getattr(args, "gene_kind", None) is None)
    ctr, why = load_counters(args.config, default_sizes, k3_name)
    sec = k3_ms * 1e-3
    valu = traffic = None
    if ctr:
        valu = ctr["SQ_INSTS_VALU"] * 64.0                # lane-ops per launch
        traffic = ctr.get("hbm_traffic_bytes_per_launch")
    copy_gbs = measured_copy_peak(eng.device)
    measured_peak = VALU_PEAK_BITOP3_4WAVES if use_lists else VALU_PEAK_AND_BCNT
    useful = adders["padded"] if adders else None
    out = {
        "bound": "valu",
        "kernel": k3_name,
        "achieved": None if valu is None else valu / sec / 1e12,
        "peak": VALU_NOMINAL_LANE_OPS / 1e12,
        "unit": "T lane-ops/s (SQ_INSTS_VALU x 64 / kernel time; peak = 256 CU x 4 SIMD x 32 lanes x 2.4 GHz)",
        "frac": None if valu is None else valu / sec / VALU_NOMINAL_LANE_OPS,
        # how much of the issue is the algorithm: full-adder lane-ops only (list_adder_work)
        "useful_lane_ops": useful,
        "useful_frac": None if useful is None else useful / sec / VALU_NOMINAL_LANE_OPS,
        "useful_frac_unpadded": None if not adders else adders["unpadded"] / sec / VALU_NOMINAL_LANE_OPS,
        "adder_ops_per_test": None if useful is None else useful / tests_per_launch,
        "overhead_ops_per_test": None if (useful is None or valu is None)
                                 else (valu - useful) / tests_per_launch,
        "padded_entries_per_gene": None if not adders else adders["padded_entries_per_gene"],
        "minority_entries_per_gene": None if not adders else adders["minority_entries_per_gene"],
        "frac_of_measured_op_peak": None if valu is None else valu / sec / measured_peak,
        "measured_op_peak": measured_peak / 1e12,
        "measured_op_peak_source": ("profiles/r02_clock_valu_both_occupancies.txt (sustained v_bitop3_b32 at "
                                    "the 4 waves/SIMD this kernel can have)") if use_lists
                                   else "tools/valu_peak.hip (v_and_b32 + v_bcnt_u32_b32)",
        "measured_op_peak_8waves": VALU_PEAK_BITOP3_8WAVES / 1e12 if use_lists else None,
        "ops_per_test": None if valu is None else valu / tests_per_launch,
        "counters_source": ctr["source"] if ctr else None,
        "counters_refused": why,
        "kernel_ms": k3_ms,
        "launches_per_step": launches_per_step,
        "tests_per_s_kernel": tests_per_launch / sec,
        # HBM side: measured traffic (PMC, gfx950-corrected) against the 8 TB/s peak
        "traffic": traffic,
        "traffic_unit": "bytes per launch, (2*FETCH_SIZE + WRITE_SIZE)*1024",
        "write_bytes": None if not ctr else ctr.get("WRITE_SIZE_bytes"),
        "hbm_frac": None if not traffic else traffic / sec / 1e9 / HBM_PEAK_GBS,
        "hbm_peak_gbs": HBM_PEAK_GBS,
        "measured_copy_peak_gbs": copy_gbs,
        # SURVEY 8d figure (i): operand bandwidth, 16*ceil(N/64) B per test -- > 1 means the
        # operands are reused on chip (the kernel left the HBM-bound regime)
        "operand_bw_gbs": operand_bytes / sec / 1e9,
        "operand_bw_frac": operand_bytes / sec / 1e9 / HBM_PEAK_GBS,
        "operand_bytes_per_launch": operand_bytes,
    }
    # the clock THIS run had (side-thread samples over the timed region): lane-ops per shader
    # clock is the code's figure, the clock is the box's -- a slow box shows up as a lower
    # sclk_mhz_mean at an unchanged ops_per_clock, a slow kernel the other way round
    sclk = telemetry and telemetry.get("sclk_mhz_mean")
    out["sclk_mhz_mean"] = sclk
    out["socket_power_w_mean"] = telemetry and telemetry.get("socket_power_w_mean")
    out["telemetry_samples"] = telemetry and telemetry.get("samples")
    out["ops_per_clock"] = None if (valu is None or not sclk) else valu / (sec * sclk * 1e6)
    out["ops_per_clock_unit"] = ("VALU lane-ops per shader clock, whole chip (SQ_INSTS_VALU x 64 / "
                                 "(kernel time x sclk_mhz_mean)); nominal 256 CU x 4 SIMD x 32 = 32768")
    out["ops_per_clock_frac"] = None if out["ops_per_clock"] is None else out["ops_per_clock"] / 32768.0
    out["useful_ops_per_clock"] = None if (useful is None or not sclk) else useful / (sec * sclk * 1e6)
    out["clock_note"] = ("sclk_mhz_mean / socket_power_w_mean are amdsmi's readings during this run's timed region; "
                         "inside a 0.1 s region after an idle GPU they are not yet calibrated (the power is a moving "
                         "average on its way up, the clock reading varies 1.7-2.2 GHz between runs of equal kernel time): "
                         "the `sustained` object of the line is the box's figure -- under sustained load the list kernel "
                         "sits at the socket power cap, 1.33-1.36 kW at 2.19-2.21 GHz (cfg5's kernel: 1.90 GHz)") \
        if use_lists else None
    return out


def ctr_lane_ops(roofline):
    """VALU lane-ops per launch behind roofline['achieved'] (T lane-ops/s x kernel seconds)."""
    return roofline["achieved"] * 1e12 * roofline["kernel_ms"] * 1e-3


def small_kernel_rooflines(args, G, N, T, n_mask_classes, kernel_ms):
    """`roofline_k1` / `roofline_k2`: K1 (k_counts) is the path's genuine HBM
    stream -- SURVEY 8d's B1 = 8 W G + 16 W T + 16 G T bytes, each gene word read once -- with an
    integer-VALU floor next to it (2 ops, AND + popcount-accumulate, per 32 isolates and (gene,
    vector); vectors = T label rows + one validity row per mask class); whichever floor is
    higher is its bound.  K2 (k_fisher) moves B2 = 32 G T bytes and is bound by its fp64
    divide chains (trip count = half the support of every table), so it is reported as a rate.
    Counter traffic comes from the committed PMC summary when it matches these sources."""
    W64 = (N + 63) // 64
    default_sizes = (args.genes is None and args.permutations is None and args.isolates is None
                     and args.traits is None and getattr(args, "gene_kind", None) is None)
    out = {}
    t1 = kernel_ms.get("k_counts")
    if t1:
        sec = t1 * 1e-3
        b1 = 8.0 * W64 * G + 16.0 * W64 * T + 16.0 * G * T
        vectors = T + n_mask_classes
        ops = 2.0 * ((N + 31) // 32) * G * vectors                     # lane-ops
        ctr, _why = load_counters(args.config, default_sizes, "k_counts")
        traffic = ctr.get("hbm_traffic_bytes_per_launch") if ctr else None
        hbm_floor, valu_floor = b1 / (HBM_PEAK_GBS * 1e9), ops / VALU_PEAK_AND_BCNT
        out["roofline_k1"] = {
            "kernel": "k_counts", "bound": "hbm" if hbm_floor >= valu_floor else "valu",
            "bytes": b1, "bytes_formula": "8*W*G + 16*W*T + 16*G*T (SURVEY 8d B1), W = ceil(N/64)",
            "kernel_ms": t1, "timed": "five back-to-back launches of the kernel alone, after the timed region",
            "gbs": b1 / sec / 1e9, "hbm_frac": b1 / sec / 1e9 / HBM_PEAK_GBS,
            "passes_over_matrix": -(-T // 32), "vectors": vectors, "mask_classes_over_passes": n_mask_classes,
            "valu_lane_ops": ops, "valu_frac": ops / sec / VALU_NOMINAL_LANE_OPS,
            "valu_frac_of_measured_and_bcnt_peak": ops / sec / VALU_PEAK_AND_BCNT,
            "hbm_floor_ms": hbm_floor * 1e3, "valu_floor_ms": valu_floor * 1e3,
            "frac_of_bound": max(hbm_floor, valu_floor) / sec,
            "traffic": traffic, "traffic_over_bytes": None if not traffic else traffic / b1,
            "counters_source": ctr["source"] if ctr else None}
    t2 = kernel_ms.get("k_fisher")
    if t2:
        sec = t2 * 1e-3
        b2 = 32.0 * G * T
        ctr, _why = load_counters(args.config, default_sizes, "k_fisher")
        out["roofline_k2"] = {
            "kernel": "k_fisher", "bound": "fp64 valu latency (divide chains, trip count = half the support)",
            "tables": G * T, "tables_per_s": G * T / sec, "kernel_ms": t2,
            "bytes": b2, "bytes_formula": "32*G*T (SURVEY 8d B2)", "gbs": b2 / sec / 1e9,
            "hbm_frac": b2 / sec / 1e9 / HBM_PEAK_GBS,
            "valu_lane_ops_per_table": None if not ctr else ctr["SQ_INSTS_VALU"] * 64.0 / (G * T),
            "in_step": "runs on the main stream while the label-tile generator has the side stream: "
                       "kernel_ms here is the kernel alone, kernel_ms['k_fisher'] of the line the shared one",
            "counters_source": ctr["source"] if ctr else None}
    return out


def k1_cold_report(eng, args, G=125_000, N=10_000, copies=8, rounds=3):
    """K1 as an HBM stream: k_counts on cfg5's per-GPU shard shape with T = 1 ... 50
    traits (round 6: the whole sweep, to locate where the kernel leaves the HBM roofline for the AND +
    popcount one), over `copies` different matrices launched in rotation.  One tiled matrix is 160 MB, eight are
    1.28 GB: when a matrix comes round again, 1.1 GB of other matrices have gone through the 256 MiB
    Infinity Cache since its last use, so every launch reads HBM.  The content does not enter the timing
    (random bits, made on the device).  The same kernel on ONE matrix back to back rides along as `warm`."""
    import torch
    from scoary_amd import synth
    from scoary_amd.engine import GeneMatrix, pack_bits_rows
    W64 = (N + 63) // 64
    Qp, Gp = eng.quads(N), eng.padded_genes(G)
    gen = torch.Generator(device=eng.device)
    gen.manual_seed(5)
    mats = [GeneMatrix(torch.randint(-2**31, 2**31 - 1, (Qp, Gp, 4), dtype=torch.int32, device=eng.device,
                                     generator=gen), G, N) for _ in range(copies)]
    copy_gbs = measured_copy_peak(eng.device)
    out = {"shape": "%d genes x %d isolates (cfg5's per-GPU shard), %d tiled matrices of %.0f MB in rotation"
                    % (G, N, copies, Qp * Gp * 16 / 1e6),
           "measured_copy_peak_gbs": copy_gbs, "hbm_peak_gbs": HBM_PEAK_GBS, "runs": []}
    rng = np.random.default_rng(11)
    for T in (1, 2, 4, 8, 16, 32, 50):
        traits = synth.make_traits(T, N, rng)
        trv = eng.vecrows(pack_bits_rows((traits == 1).astype(np.uint8)), N)
        mkv = eng.vecrows(pack_bits_rows((traits != 2).astype(np.uint8)), N)
        plan = eng.trait_plan(trv, mkv, N)
        counts = eng._empty((T, G, 4), torch.int32)
        b1 = 8.0 * W64 * G + 16.0 * W64 * T + 16.0 * G * T

        def timed(seq):
            for m in seq[:copies]:
                eng.counts(m, trv, mkv, out=(counts,), plan=plan)
            torch.cuda.synchronize()
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(seq) + 1)]
            evs[0].record()
            for i, m in enumerate(seq):
                eng.counts(m, trv, mkv, out=(counts,), plan=plan)
                evs[i + 1].record()
            torch.cuda.synchronize()
            ms = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(len(seq)))
            return sum(ms) / len(ms), ms[len(ms) // 2]
        cold_mean, cold_med = timed(mats * rounds)
        warm_mean, warm_med = timed([mats[0]] * (copies * rounds))
        # the two floors of this launch: the B1 bytes at the HBM peak, and 2 lane-ops (AND, popcount-accumulate)
        # per 32 isolates and (gene, vector) at the measured rate of that op pair; vectors = T label rows + one
        # validity row per pass (no missing values here: one mask class)
        passes = -(-T // int(eng.lib.scoary_counts_traits_per_pass(T)))
        ops = 2.0 * ((N + 31) // 32) * G * (T + passes)
        hbm_floor, valu_floor = b1 / (HBM_PEAK_GBS * 1e9), ops / VALU_PEAK_AND_BCNT
        out["runs"].append({
            "traits": T, "bytes": b1, "bytes_formula": "8*W*G + 16*W*T + 16*G*T (SURVEY 8d B1)",
            "passes_over_matrix": passes, "hbm_floor_ms": hbm_floor * 1e3, "valu_floor_ms": valu_floor * 1e3,
            "bound": "hbm" if hbm_floor >= valu_floor else "valu",
            "frac_of_bound": max(hbm_floor, valu_floor) / (cold_med * 1e-3),
            "cold_ms_mean": cold_mean, "cold_ms_median": cold_med,
            "gbs": b1 / (cold_med * 1e-3) / 1e9,
            "hbm_frac": b1 / (cold_med * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "frac_of_measured_copy_peak": b1 / (cold_med * 1e-3) / 1e9 / copy_gbs,
            "warm_ms_median": warm_med, "warm_gbs": b1 / (warm_med * 1e-3) / 1e9,
            "launches": copies * rounds})
    hb = [r["traits"] for r in out["runs"] if r["bound"] == "hbm"]
    out["crossover"] = ("HBM floor above the AND + popcount floor up to T = %s traits on this shape (B1 at 8 TB/s against "
                        "2 lane-ops per 32 isolates and vector at %.1e lane-ops/s)" % (max(hb) if hb else 0,
                                                                                    VALU_PEAK_AND_BCNT))
    return out


def make_step(torch, eng, gm, trv, mkv, P, seed, ws, use_lists, plan, exchange, use_graph):
    """-> (step(i), graphs): one pass of the hot path on this rank's genes + (sharded) the submit of
    its records to the exchange.  With ``use_graph`` the local step is a hipGraph replay; a sharded
    rank records TWO graphs that pack their records into alternating buffers, because step i's
    gather is still reading its buffer while step i + 1 runs."""
    graphs = []
    if use_graph:
        # recorded here, outside the warm-up count: what engine.associate does by itself on the
        # second call of a launch-bound step
        if exchange is not None:
            from scoary_amd import dist as sdist
            for _ in range(2):
                rec = torch.empty((trv.shape[0], gm.G, sdist.REC_WORDS), dtype=torch.int32, device=eng.device)
                graphs.append(eng.capture(gm, trv, mkv, P, seed, ws, use_lists=use_lists, plan=plan, records=rec))
        else:
            graphs.append(eng.capture(gm, trv, mkv, P, seed, ws, use_lists=use_lists, plan=plan))

    def step(i):
        if exchange is not None:
            # before anything of this step is launched: all gathers but the newest are complete, so
            # the record buffer this step packs into (graph i % 2) is no longer being read
            exchange.drain(keep=1)
        if graphs:
            g, res = graphs[i % len(graphs)]
            g.launch()
        else:
            res = eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=use_lists,
                                workspace=ws, plan=plan, graph=False)
        if exchange is not None:
            exchange.submit(res)
        return res
    return step, graphs


def strong_split_case(torch, dist, eng, args, cfg, world, rank, local_rank, sharded):
    """One point of the strong-scaling curve: config `cfg` at its full size, its genes split over the
    ranks of this run (GenePartition, --partition), timed like the line itself -- warm-up, barrier,
    `--steps` steps with the asynchronous gather on rank 0, barrier, MAX over ranks.  A launch-bound
    shard replays its local step from a hipGraph.  Returns the entry of `scaling_strong` (rank 0)."""
    from scoary_amd import synth
    from scoary_amd import dist as sdist
    from scoary_amd.engine import pack_bits_rows
    base, traits, P, seed = synth.make_config(cfg)
    base = order_genes(base, args.gene_order)
    part = sdist.GenePartition(base.shape[0], world, args.partition)
    genes = np.ascontiguousarray(base[part.index(rank)])
    del base
    G, N = genes.shape
    T = traits.shape[0]
    ones = genes.sum(axis=1, dtype=np.int64)
    list_entries = int(np.minimum(ones, N - ones).sum())
    gm = eng.tile_rows(pack_bits_rows(genes), N)
    del genes
    trv = eng.vecrows(pack_bits_rows((traits == 1).astype(np.uint8)), N)
    mkv = eng.vecrows(pack_bits_rows((traits != 2).astype(np.uint8)), N)
    plan = eng.trait_plan(trv, mkv, N)
    use_lists = args.kernel != "dense" and LISTS_DEFAULT and eng.lists_supported(N)
    if use_lists:
        eng.build_lists(gm)
    ws = eng.workspace(gm, T, P, use_lists=use_lists)
    exchange = Exchange(torch, eng, world, rank, T, part) if sharded else None
    use_graph = (not args.no_graph and not (sharded and args.label_shards)
                 and eng.auto_graph_eligible(gm, T, P))
    step, graphs = make_step(torch, eng, gm, trv, mkv, P, seed, ws, use_lists, plan, exchange, use_graph)

    def barrier():
        if exchange:
            exchange.drain()
            dist.barrier()
        torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    barrier()
    last = torch.cuda.Event()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    host_issue_ms = (time.perf_counter() - t0) * 1e3 / max(args.steps, 1)
    exposed_ms = None
    if exchange:
        last.record()
        last.synchronize()
        t_kernels = time.perf_counter()
    barrier()
    dt_own = dt = time.perf_counter() - t0
    if exchange:
        exposed_ms = (time.perf_counter() - t_kernels) * 1e3
    # per-kernel durations: a few eager steps with the library's event timers on
    eng.set_timing(True)
    for _ in range(3):
        eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=use_lists, workspace=ws, plan=plan,
                      graph=False)
    torch.cuda.synchronize()
    k3_name = eng.list_kernel_name(N) if use_lists else "k_permute"
    names = ("k_counts", "k_fisher") + (("k_perm_generate_tiles", k3_name, "k_lists_reduce") if use_lists
                                        else ("k_perm_generate", "k_permute"))
    kernel_ms = {k: eng.kernel_ms(k) for k in names}
    eng.set_timing(False)
    per_rank = rccl_ranks = None
    if sharded:
        tmax = torch.tensor([dt], dtype=torch.float64, device=eng.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        per_rank = sdist.all_gather_objects({
            "rank": rank, "genes": G, "list_entries": list_entries, "ms_per_step": dt_own / args.steps * 1e3,
            "kernel_ms": kernel_ms, "exchange_exposed_ms": exposed_ms, "host_issue_ms_per_step": host_issue_ms,
            "exchange_bytes": T * G * sdist.REC_WORDS * 4})
        rccl_ranks = exchange.check(T, (traits != 2).sum(1))
        if rank == 0 and exchange.kind == "gather" and rccl_ranks != world:
            raise SystemExit("bench.py: scaling_strong %s: valid records from %s of %d ranks"
                             % (cfg, rccl_ranks, world))
    for g_ in graphs:
        g_[0].close()
    tests = float(part.G) * T * P
    return {"workload": "%s: %d genes x %d isolates x %d traits, --permute %d in all, split over %d GPU(s)"
                        % (cfg, part.G, N, T, P, world),
            "n_gpus": world, "value": tests * args.steps / dt, "unit": "tests/s",
            "ms_per_step": dt / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup,
            "genes_per_gpu": part.lengths(), "gene_partition": part.kind, "gene_order": args.gene_order,
            "hip_graph": bool(graphs), "kernel_ms": kernel_ms, "exchange_exposed_ms": exposed_ms,
            "host_issue_ms_per_step": host_issue_ms, "rccl_ranks": rccl_ranks, "per_rank": per_rank}


def dry_exchange(args, world, rank):
    """CPU-only: the launcher, process group, Exchange pipeline and rank-0 check with
    fabricated records (gloo)."""
    import types
    import torch
    import torch.distributed as dist
    from scoary_amd import dist as sdist
    sdist.init_from_env()
    T, G, nval = 2, 7, [11, 13]
    part = sdist.GenePartition(G, world, args.partition) if args.scaling == "strong" \
        else sdist.GenePartition(G * world, world, "contiguous")
    Gs = part.length(rank)
    ex = Exchange(torch, types.SimpleNamespace(device="cpu"), world, rank, T, part)
    for step in range(args.steps):
        counts = torch.zeros((T, Gs, 4), dtype=torch.int32)
        for t in range(T):
            counts[t, :, 0] = nval[t] - 3
            counts[t, :, 1] = 1 + (rank + step) % 2
            counts[t, :, 2] = 2 - (rank + step) % 2
        ex.submit({"counts": counts, "p": torch.full((T, Gs), 0.5, dtype=torch.float64),
                   "odds": torch.ones((T, Gs), dtype=torch.float64),
                   "r": torch.full((T, Gs), rank, dtype=torch.int32)})
    ex.drain()
    dist.barrier()
    ok = ex.check(T, nval)
    if rank == 0:
        print(json.dumps({"dry_exchange": True, "n_gpus": world, "rccl_ranks": ok,
                          "scaling": args.scaling, "exchange": ex.kind, "steps": args.steps}))
        if ok != world:
            raise SystemExit("dry exchange: only %s of %d ranks delivered valid records" % (ok, world))
    dist.destroy_process_group()


def main():
    args = parse()
    maybe_relaunch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.dry_exchange:
        return dry_exchange(args, world, rank)

    import torch
    import torch.distributed as dist

    sharded = world > 1 or (args.exercise_exchange and "RANK" in os.environ)
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); none visible")
    if args.share_gpu:
        local_rank %= torch.cuda.device_count()
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    if sharded:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from scoary_amd import synth
    from scoary_amd import dist as sdist
    from scoary_amd.engine import AssociationEngine, pack_bits_rows

    c = synth.CONFIGS[args.config]
    G_cfg = args.genes or (c["G"] // 8 if args.config == "cfg5" else c["G"])   # cfg5: the per-GPU shard
    strong = args.scaling == "strong" and world > 1
    base_genes, traits, P, seed = synth.make_config(args.config, G=G_cfg, N=args.isolates, T=args.traits,
                                                    gene_kind=args.gene_kind)
    base_genes = order_genes(base_genes, args.gene_order)
    # the genes of the whole run: strong -- the config's, in the reference's stride domains (or
    # --partition contiguous); weak -- world blocks of G_cfg, one per rank
    part = sdist.GenePartition(G_cfg, world, args.partition) if strong \
        else sdist.GenePartition(G_cfg * world, world, "contiguous")

    def shard_of(rk):
        """The gene rows rank rk works on.  strong: the config's genes, split -- every rank
        generates the same matrix and keeps its rows; weak: every rank its own G-gene shard
        (different seed offset), same traits."""
        if strong:
            return np.ascontiguousarray(base_genes[part.index(rk)])
        if rk == 0:
            return base_genes
        rng = np.random.default_rng(seed + 1000 * rk)
        return order_genes(synth.make_genes(
            base_genes.shape[0], base_genes.shape[1], rng,
            kind=args.gene_kind or ("rare" if args.config == "cfg4" else "uniform"),
            core_frac=0.05 if args.config in ("cfg3", "cfg5") else 0.0), args.gene_order)
    genes = shard_of(rank)
    if args.permutations:
        P = args.permutations
    G, N = genes.shape
    T = traits.shape[0]
    G_total = part.G

    eng = AssociationEngine(local_rank)
    if sharded and args.label_shards:
        eng.label_shards = sdist.LabelShards()     # 1 / world of every batch of label tiles + one all-gather
    rows64 = pack_bits_rows(genes)                 # host packing: excluded like file parsing
    tbits = pack_bits_rows((traits == 1).astype(np.uint8))
    mbits = pack_bits_rows((traits != 2).astype(np.uint8))
    torch.cuda.synchronize()

    def setup():
        gm_ = eng.tile_rows(rows64, N)             # H2D of the packed bits + device tiling
        trv_, mkv_ = eng.vecrows(tbits, N), eng.vecrows(mbits, N)
        plan_ = eng.trait_plan(trv_, mkv_, N)      # trait margins + mask classes: once per trait set
        ul = args.kernel == "lists" or (args.kernel == "auto" and LISTS_DEFAULT
                                        and eng.lists_supported(N))
        if ul:
            eng.build_lists(gm_)                   # on the device, from the tiled matrix
        torch.cuda.synchronize()
        return gm_, trv_, mkv_, ul, plan_
    setup()                                        # first call pays module loads; time the second
    t0 = time.perf_counter()
    gm, trv, mkv, use_lists, plan = setup()
    setup_ms = (time.perf_counter() - t0) * 1e3

    pbatch = eng.perm_batch(T, N, P)
    ws = eng.workspace(gm, T, P, use_lists=use_lists)
    exchange = Exchange(torch, eng, world, rank, T, part) if sharded else None
    # a launch-bound step is replayed from a hipGraph -- also on a gene-sharded rank (round 6: a rank of
    # cfg4's 8-way split has 0.1 ms of launch-bound kernels in 0.35 ms): the LOCAL step is recorded,
    # record packing included; the exchange (a collective) stays outside the graph.  Label shards put
    # a collective INSIDE the step: those run eagerly.
    auto_graph = (not args.graph and not args.no_graph and not (sharded and args.label_shards)
                  and eng.auto_graph_eligible(gm, T, P))
    step_fn, graphs = make_step(torch, eng, gm, trv, mkv, P, seed, ws, use_lists, plan, exchange,
                                use_graph=args.graph or auto_graph)
    graph = graphs[0][0] if graphs else None
    step_no = [0]

    def step():
        step_no[0] += 1
        return step_fn(step_no[0])

    def barrier():
        if exchange:
            exchange.drain()
            dist.barrier()
        torch.cuda.synchronize()

    # shader clock / socket power of THIS run, sampled from a side thread during the timed region.
    # The sensor is opened BEFORE the warm-up (amdsmi's start-up takes tens of milliseconds: the GPU
    # must not sit idle between the warm-up steps and the timed region).
    # A launch-bound run (cfg2: a 1-2 ms timed region of 0.06 ms graph launches) is not sampled from
    # inside: one amdsmi read costs a few hundred microseconds of interpreter time and would be the
    # largest thing in the region.  It gets one reading right before and one right after instead.
    tele = Telemetry(local_rank, period_s=max(0.001, args.telemetry_ms * 1e-3)) if args.telemetry_ms > 0 else None
    tele_inside = tele is not None and not eng.auto_graph_eligible(gm, T, P)
    for _ in range(args.warmup):
        step()
    barrier()
    if graph is None:
        eng.set_timing(True)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    edge = []
    if tele is not None and not tele_inside and tele._read:
        edge.append((time.perf_counter(),) + tuple(tele._read()))
    if tele_inside:
        tele.start()
    t0 = time.perf_counter()
    ev[0].record()
    for i in range(args.steps):
        step()
        ev[i + 1].record()
    # how long the HOST took to issue the steps (launches, graph replays, the exchange's submit): when this
    # is close to the whole timed region the run is bound by the interpreter, not by the GPU
    host_issue_ms = (time.perf_counter() - t0) * 1e3 / max(args.steps, 1)
    exposed_ms = None
    if exchange:
        # what the exchange step costs beyond the kernels: wall clock from the moment this
        # rank's last kernel (the record packing of the last step) has finished to the end of
        # the timed region -- draining the gathers still in flight + the closing barrier
        ev[args.steps].synchronize()
        t_kernels = time.perf_counter()
    barrier()
    dt = time.perf_counter() - t0
    if exchange:
        exposed_ms = (time.perf_counter() - t_kernels) * 1e3
    if tele is not None and not tele_inside:
        if tele._read:
            edge.append((time.perf_counter(),) + tuple(tele._read()))
        tele.samples = edge
        telemetry = tele.stop()
        telemetry["sampled"] = "one reading before and one after the timed region (launch-bound run)"
    else:
        telemetry = tele.stop(t0, t0 + dt) if tele is not None else None
        if telemetry is not None:
            telemetry["sampled"] = "inside the timed region"
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2] if step_ms else None
    if graph is not None:                           # per-kernel times: a few eager steps afterwards
        eng.set_timing(True)
        for _ in range(5):
            eng.associate(gm, trv, mkv, permutations=P, seed=seed, use_lists=use_lists, workspace=ws, plan=plan,
                          graph=False)
        torch.cuda.synchronize()
    k3_name = eng.list_kernel_name(N) if use_lists else "k_permute"
    k3_ms = eng.kernel_ms(k3_name)
    names = ("k_counts", "k_fisher") + (
        ("k_perm_generate_tiles", k3_name, "k_lists_reduce") if use_lists else
        ("k_perm_generate", "k_permute"))
    kernel_ms = {k: eng.kernel_ms(k) for k in names}
    eng.set_timing(False)
    # K1 and K2 by themselves (roofline_k1 / roofline_k2): inside a step k_fisher shares the chip with
    # the label-tile generator on the side stream, so its in-step duration is not its own
    iso = {}
    saved_shards, ws.label_shards = ws.label_shards, None     # the generator alone: all tiles, no collective
    gen = ()
    if use_lists and P > 0:
        gen = (("k_perm_generate_tiles",
                lambda: eng.perm_generate_tiles(mkv, plan.margins, N, min(ws.batch, P), 0, seed, out=ws.tiles)),)
    for name, fn in gen + (("k_counts", lambda: eng.counts(gm, trv, mkv, out=(ws.counts,), plan=plan)),
                     ("k_fisher", lambda: eng.fisher(ws.counts, out=(ws.p, ws.odds, ws.crit),
                                                     **({"lists": gm.lists, "lcrit": ws.lcrit} if use_lists and P > 0
                                                        else {})))):
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        e1.synchronize()
        iso[name] = e0.elapsed_time(e1) / 5
    ws.label_shards = saved_shards
    # The command line's output-fidelity pass (scoary_fisher_scipy: SciPy's own digits for what gets printed above
    # 170 isolates) is NOT in the timed step -- the path's p is k_fisher's, within 1e-12 --; its cost, for the record:
    scipy_digits = None
    if N > 170 and hasattr(eng, "fisher_scipy"):
        pcopy = ws.p.clone()
        eng.fisher_scipy(ws.counts, pcopy)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        eng.fisher_scipy(ws.counts, pcopy)
        e1.record()
        e1.synchronize()
        scipy_digits = {"kernel": "k_fisher_scipy", "kernel_ms": e0.elapsed_time(e1), "tables": int(G) * int(T),
                        "in_timed_step": False,
                        "what": "scipy.stats.fisher_exact's own double for every table (Boost's prime-factorised pmf "
                                "restated): run by the command line over what it prints, so that its result files are "
                                "the reference's bytes; the benchmarked step keeps k_fisher's p (within 1e-12)",
                        "max_abs_change_of_p": float((pcopy - ws.p).abs().max().item())}
        del pcopy
    k1_cold = k1_cold_report(eng, args) if (not args.no_k1_cold and not sharded) else None

    # The box under SUSTAINED load (single GPU, outside the timed region, GPU still warm): the same
    # step back to back for --sustain-seconds, clock and power from the second half of the window.
    # amdsmi's readings inside a 0.1 s region that follows an idle GPU are not a calibrated clock
    # (1.73 ... 2.22 GHz reported across runs whose kernel times agree to 0.5 %; the socket power
    # is a moving average still on its way up); after a second at the power cap they are
    # (2.19-2.21 GHz at 1.33-1.36 kW, as rocm-smi says).  This is what tells a slow box from a
    # slow kernel: the sustained clock the box grants this kernel, and lane-ops per that clock.
    sustained = None
    if args.sustain_seconds > 0 and not sharded and median_ms and median_ms < 200.0 \
            and tele is not None and tele._read is not None:
        tele2 = Telemetry(local_rank, period_s=0.01).start()
        s0 = time.perf_counter()
        nsteps = 0
        burst = max(1, int(50.0 / median_ms))                 # ~50 ms of steps per synchronize
        while time.perf_counter() - s0 < args.sustain_seconds:
            for _ in range(burst):
                step()
            nsteps += burst
            torch.cuda.synchronize()
        s1 = time.perf_counter()
        sustained = tele2.stop(s0 + 0.5 * (s1 - s0), s1)
        sustained.update({"steps": nsteps, "seconds": s1 - s0, "ms_per_step": (s1 - s0) / nsteps * 1e3,
                          "value": G * T * P * nsteps / (s1 - s0),
                          "what": "the same step back to back after the timed region; clock / power over the "
                                  "second half of the window"})
    dt_own = dt
    per_rank = None
    if sharded:                                      # MAX over ranks
        tmax = torch.tensor([dt], dtype=torch.float64,
                            device=eng.device if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        # one line per rank, so that a sub-linear point of the scaling curve explains itself:
        # its own wall clock, its kernels, the bytes it sends per step and how long the
        # exchange kept it waiting after its last kernel
        ones = genes.sum(axis=1, dtype=np.int64)
        mine = {"rank": rank, "device": local_rank, "genes": G, "ms_per_step": dt_own / args.steps * 1e3,
                "ms_per_step_median": median_ms, "kernel_ms": kernel_ms,
                # what the list kernel's cost follows: the minority entries of this rank's genes
                "list_entries": int(np.minimum(ones, N - ones).sum()),
                "exchange_exposed_ms": exposed_ms, "host_issue_ms_per_step": host_issue_ms,
                "exchange_bytes": T * G * sdist.REC_WORDS * 4,
                "label_tile_bytes_gathered": None if eng.label_shards is None
                else eng.label_shards.bytes_gathered // max(args.steps + args.warmup, 1),
                "sclk_mhz_mean": telemetry and telemetry["sclk_mhz_mean"],
                "socket_power_w_mean": telemetry and telemetry["socket_power_w_mean"]}
        per_rank = sdist.all_gather_objects(mine)
    nval = (traits != 2).sum(1)
    rccl_ranks = exchange.check(T, nval) if exchange else None
    failure = None                                   # rank 0's verdict; every rank leaves together (below)
    if exchange and rank == 0 and exchange.kind == "gather" and rccl_ranks != world:
        failure = "bench.py: rank 0 received valid records from %s of %d ranks" % (rccl_ranks, world)
    gather_ok = None
    if args.verify_gather and exchange and rank == 0 and exchange.kind == "gather" and failure is None:
        # rank 0 alone recomputes every rank's shard and compares the block it received from
        # that rank in the last step, record for record (counts, p, odds, r: 40 bytes each)
        last = exchange.last_block()
        if args.inject_gather_fault:
            last[world - 1, 0, 0, 8] += 1                    # r of the last rank's first record
        gather_ok = True
        saved_label_shards, eng.label_shards = eng.label_shards, None   # rank 0 alone: every tile generated here
        for rk, n in enumerate(part.lengths()):
            g_rk = eng.tile_rows(pack_bits_rows(shard_of(rk)), N)
            if use_lists:
                eng.build_lists(g_rk)
            alone = eng.pack_records(eng.associate(g_rk, trv, mkv, permutations=P, seed=seed,
                                                   use_lists=use_lists, plan=plan))
            gather_ok = gather_ok and bool(torch.equal(last[rk, :, :n], alone))
        if gather_ok and strong:
            # ... and woven back into gene order (GenePartition.weave) they are the records of ONE
            # GPU working on the whole matrix
            g_all = eng.tile_rows(pack_bits_rows(base_genes), N)
            if use_lists:
                eng.build_lists(g_all)
            whole = eng.pack_records(eng.associate(g_all, trv, mkv, permutations=P, seed=seed,
                                                   use_lists=use_lists, plan=plan))
            gather_ok = bool(torch.equal(part.weave(last), whole))
        eng.label_shards = saved_label_shards
        if not gather_ok:
            failure = "bench.py: the gathered records differ from a single-rank run"
    if sharded:
        # the other ranks wait for rank 0's checks, and rank 0's verdict reaches all of them:
        # a failed check ends every rank with a non-zero status instead of leaving the others
        # blocked in a barrier until the launcher's timeout
        verdict = [failure]
        dist.broadcast_object_list(verdict, src=0)
        if verdict[0] is not None:
            dist.destroy_process_group()
            raise SystemExit(verdict[0] if rank == 0 else 1)
    elif failure is not None:
        raise SystemExit(failure)

    # ---- the strong curve next to the weak line (same process group, after the line's own timing) ----
    default_line = (args.config == "cfg3" and args.genes is None and args.permutations is None
                    and args.isolates is None and args.traits is None and args.gene_kind is None
                    and args.scaling == "weak")
    scaling_strong = None
    if args.strong_extra == "on" or (args.strong_extra == "auto" and default_line):
        scaling_strong = {"what": "the config's genes split over the n_gpus ranks of this run (stride domains, "
                                  "scoary_amd.dist.GenePartition), exchange included; value = the config's "
                                  "G x T x P tests / the slowest rank's time; speed-up over one GPU = value / the "
                                  "same entry of the n_gpus = 1 line"}
        for cfg_ in ("cfg3", "cfg4"):
            if world == 1 and cfg_ == args.config and default_line and not sharded:
                scaling_strong[cfg_] = {"same_as": "the line's own value (one GPU holds all genes)",
                                        "value": G_total * T * P * args.steps / dt,
                                        "ms_per_step": dt / args.steps * 1e3, "n_gpus": 1}
                continue
            scaling_strong[cfg_] = strong_split_case(torch, dist, eng, args, cfg_, world, rank, local_rank, sharded)

    if rank == 0:
        tests_per_step = G_total * T * P
        out = {
            "metric": "gene x permutation Fisher tests/sec",
            "value": tests_per_step * args.steps / dt,
            "unit": "tests/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_median": median_ms,
            "host_issue_ms_per_step": host_issue_ms,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "u32 bit-words (bit-sliced adders / AND + popcount), f64 for Fisher p",
            "data": "synthetic",
            "config": {"workload": "%s%s: %d genes x %d isolates x %d traits, --permute %d %s; "
                                   "counts + Fisher + label permutations + exceedance counts"
                                   % (args.config, "" if not args.gene_kind else " shape, %s gene frequencies"
                                      % args.gene_kind, G_total if strong else G, N, T, P,
                                      "in all (split across the GPUs)" if strong else "per GPU"),
                       "gene_kind": args.gene_kind or "config", "gene_order": args.gene_order,
                       "gene_partition": ("%s (scoary_amd.dist.GenePartition)" % part.kind) if strong
                       else "every rank its own genes (weak scaling)",
                       "genes_per_gpu": G, "genes_total": G_total, "isolates": N, "traits": T,
                       "permutations": P, "parallelism": "gene-shard x%d" % world,
                       "hip_graph": bool(graph), "hip_graph_auto": bool(auto_graph),
                       "label_tiles": "1/%d per rank + all_gather_into_tensor" % world
                       if (sharded and args.label_shards) else "generated on every rank",
                       "exchange": ("%s %s of per-gene records" % (
                           "rccl" if args.backend == "nccl" else "gloo (shared-GPU functional check)",
                           exchange.kind)) if exchange
                       else "none (single GPU)"},
            "rccl_ranks": rccl_ranks,
            "gather_matches_single_rank": gather_ok,
            "per_rank": per_rank,
            "scaling_strong": scaling_strong,
            # once per data set, outside the timed region: H2D of the packed bits + device tiling +
            # device list build + trait vectors and their plan (margins, mask classes); host
            # bit-packing and parsing excluded
            "setup_ms": setup_ms,
            "value_incl_setup": tests_per_step * args.steps / (dt + setup_ms * 1e-3),
            "value_single_step_incl_setup": tests_per_step / (dt / args.steps + setup_ms * 1e-3),
            # SURVEY 8d's end-to-end figure: the observed tables count as tests too, G*T*(P+1)
            "value_incl_observed_tables": G_total * T * (P + 1) * args.steps / dt,
            "roofline": roofline_report(
                args, eng, use_lists, G, N, T, P, ws.batch if use_lists else pbatch, k3_name, k3_ms,
                adders=list_adder_work(eng, gm, genes, T, P, ws.batch) if use_lists else None,
                telemetry=telemetry),
            "telemetry": telemetry,
            "kernel_ms": kernel_ms,
        }
        cls = plan.mask_class.cpu().numpy()
        tpp = int(eng.lib.scoary_counts_traits_per_pass(T))       # classes are counted once per PASS
        out["kernel_source_sha256"] = kernel_source_sha()      # same hash + another time = another box
        if sustained is not None:
            v = out["roofline"].get("achieved")
            lane_ops = None if v is None else ctr_lane_ops(out["roofline"])
            sclk = sustained.get("sclk_mhz_mean")
            sustained["ops_per_clock_frac"] = None if (lane_ops is None or not sclk) else (
                lane_ops / (out["roofline"]["kernel_ms"] * 1e-3 * sclk * 1e6) / 32768.0)
            sustained["ops_per_clock_note"] = ("SQ_INSTS_VALU x 64 / (the timed region's kernel time x the sustained "
                                               "clock) / 32768: what the kernel issues per clock if the timed region "
                                               "already ran at the sustained clock")
        out["sustained"] = sustained
        # (the driver's record keeps `roofline` and `config` whole and only the NAMES of other keys:
        # the sustained window and the strong-curve entries ride along inside them, in short form)
        out["roofline"]["sustained"] = None if sustained is None else {
            k: sustained.get(k) for k in ("value", "ms_per_step", "seconds", "steps", "sclk_mhz_mean",
                                          "socket_power_w_mean", "ops_per_clock_frac")}
        out["config"]["scaling_strong"] = None if scaling_strong is None else {
            k: {f: v.get(f) for f in ("n_gpus", "value", "ms_per_step", "hip_graph", "rccl_ranks", "genes_per_gpu")}
            for k, v in scaling_strong.items() if isinstance(v, dict)}
        out["fisher_scipy_digits"] = scipy_digits
        out["kernel_ms_isolated"] = iso            # five back-to-back launches of the kernel alone
        out["kernel_ms_note"] = ("kernel_ms: hipEvent durations inside the timed steps -- there the label generator "
                                 "(k_perm_generate_tiles) runs on a side stream WHILE k_fisher runs on the main one, so "
                                 "both durations are those of two kernels sharing the chip; kernel_ms_isolated: each of "
                                 "them (and k_counts) launched alone, five times back to back")
        out.update(small_kernel_rooflines(
            args, G, N, T, sum(len(np.unique(cls[a:a + tpp])) for a in range(0, T, tpp)), iso))
        if "roofline_k1" in out:
            out["roofline_k1"]["note"] = (
                "timed on ONE matrix launched back to back: a matrix below the 256 MiB Infinity Cache (every "
                "BASELINE shape but cfg5's shard) is served from it after the first launch, so gbs here is a "
                "cache-resident rate, not an HBM rate; roofline_k1.cold (bench.py --k1-cold, "
                "profiles/r05_k1_stream.json) is the HBM figure")
            if k1_cold is not None:
                out["roofline_k1"]["cold"] = k1_cold
        if world == 1 and not args.no_cpu_baseline:
            port = cpu_baseline_port(genes, traits, N, seed, args.cpu_seconds)
            try:
                out["cpu_baseline"] = cpu_baseline_scipy(eng, genes, traits, N, seed, args.cpu_seconds)
            except Exception as e:                        # the GPU line must not die with the CPU leg
                out["cpu_baseline"] = {"value": None, "kind": "scipy-restatement",
                                       "error": "%s: %s" % (type(e).__name__, e)}
            out["cpu_baseline_port"] = port
        print(json.dumps(out))
    if sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
