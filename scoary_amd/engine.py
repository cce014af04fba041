"""Device-side association engine: thin Python over the C-ABI.

PyTorch-ROCm is used for device memory and streams only; every computation of
the hot path happens in libscoary_hip.so (scoary_amd/csrc/scoary_*.hip).
"""
import ctypes

import numpy as np

from . import _abi


def _torch():
    import torch
    return torch


def pack_bits_rows(dense01):
    """Host helper (spec S1): (R, N) 0/1 array -> (R, W64) uint64 rows, bit i
    of word w = column 64*w+i.  Pure numpy; this is input preparation, the
    same bit-packing the CSV reader produces."""
    dense01 = np.ascontiguousarray(dense01, dtype=np.uint8)
    R, N = dense01.shape
    W = (N + 63) // 64
    by = np.packbits(dense01, axis=1, bitorder="little")
    out = np.zeros((R, W * 8), dtype=np.uint8)
    out[:, :by.shape[1]] = by
    return out.view("<u8")


class GeneMatrix:
    """The bit-packed gene presence/absence matrix resident in HBM (tiled)."""

    def __init__(self, tiled, G, N):
        self.tiled = tiled      # torch.int32 [Qp, Gp, 4]
        self.G = int(G)
        self.N = int(N)
        self.lists = None       # GeneLists, for the list-driven permutation kernel


class ListMemoryError(_abi.ScoaryHipError):
    """The index lists of a gene matrix would not fit the memory budget (build_lists): the
    caller falls back to the dense permutation kernels, which need no lists."""


class GeneLists:
    """Minority index lists of a gene matrix on the device (scoary_lists_plan / _fill).
    start / ngroups: int32 [G] per list slot, or [segments, G] for N > 20479."""

    def __init__(self, idx, start, ngroups, order, flipped, entries):
        self.idx, self.start, self.ngroups = idx, start, ngroups
        self.order, self.flipped, self.entries = order, flipped, entries


class TraitPlan:
    """What the counts need that depends on the traits alone (scoary_trait_plan): margins
    int32 [T, 2] = (positives, valid isolates), mask_class int32 [T] = first trait OF THE SAME PASS with the same
    validity row, buf = the opaque plan buffer (class slots per pass + the label / validity
    rows gathered quad-major).  Built once per trait set; valid for exactly the traits / masks
    tensors it was built from (a snapshot: later in-place changes of them are not seen)."""

    def __init__(self, margins, mask_class, buf, traits, masks, N):
        self.margins, self.mask_class, self.buf, self.N = margins, mask_class, buf, int(N)
        # strong references: the memory the plan describes cannot be recycled for other data
        # while the plan lives, and `is` identifies the tensors (an address can be reused);
        # _version sees in-place edits made through torch after the snapshot
        self.traits, self.masks = traits, masks
        self.versions = (traits._version, masks._version)

    def fits(self, traits, masks):
        return (traits is self.traits and masks is self.masks
                and self.versions == (traits._version, masks._version))


class Workspace:
    """Device buffers of one associate() step (engine.workspace)."""

    def __init__(self, eng, genes, T, permutations, use_lists, perm_buffer=None):
        torch = _torch()
        G, N = genes.G, genes.N
        if use_lists is None:
            use_lists = genes.lists is not None and eng.lists_supported(N)
        self.key = (G, N, int(T), int(permutations), bool(use_lists and permutations > 0))
        self.eng = eng
        self.counts = eng._empty((T, G, 4), torch.int32)
        self.margins = eng._empty((T, 2), torch.int32)
        self.mask_class = eng._empty((T,), torch.int32)
        self.plan_buf = eng._empty((int(eng.lib.scoary_trait_plan_bytes(int(T), N)) // 4,), torch.int32)
        self.p = eng._empty((T, G), torch.float64)
        self.odds = eng._empty((T, G), torch.float64)
        self.crit = eng._empty((T, G, 2), torch.int32) if permutations > 0 else None
        self.r = eng._empty((T, G), torch.int32) if permutations > 0 else None
        self.tiles = self.scratch = self.perms = self.lcrit = None
        self.label_shards = None
        self.auto = None        # engine.associate's cached hipGraph of a launch-bound step
        self.batch = 0
        if permutations > 0 and use_lists:
            self.batch = eng.list_batch(T, N, permutations, G)
            nb0 = min(self.batch, permutations)
            words = int(eng.lib.scoary_list_tiles_words(N, nb0, T))
            if eng.label_shards is not None:          # room for world equal chunks (dist.LabelShards)
                words = max(words, eng.label_shards.padded_words(*eng.tiles_per_batch(N, nb0, T)))
            self.tiles = eng._empty((words,), torch.int32)
            self.label_shards = eng.label_shards
            self.scratch = eng.permute_lists_scratch(G, T, N, nb0)
            self.lcrit = eng._empty((T, G, 2), torch.int32)
        elif permutations > 0:
            if perm_buffer is not None:
                self.perms = perm_buffer
            else:
                self.perms = eng._empty((T, eng.perm_batch(T, N, permutations), eng.row_words(N)),
                                        torch.int32)

    def fits(self, genes, T, permutations, use_lists):
        # the label shards are part of the shape: a workspace made before eng.label_shards was set
        # (or kept after it was reset) has the wrong tile padding, and ranks that disagree about
        # the shards would block in the all-gather of _label_tiles
        if self.tiles is not None and self.label_shards is not self.eng.label_shards:
            return False
        return self.key == (genes.G, genes.N, int(T), int(permutations),
                            bool(use_lists and permutations > 0))


class StepGraph:
    """A captured associate() step (engine.capture)."""

    def __init__(self, eng, graph, stream):
        self.eng, self.graph, self.stream = eng, graph, stream
        self.done = None        # event behind the most recent launch

    def launch(self):
        eng = self.eng
        eng._check(eng.lib.scoary_graph_launch(eng.h, self.graph, eng._stream()),
                   "scoary_graph_launch")
        if self.done is None:
            self.done = _torch().cuda.Event()
        self.done.record(_torch().cuda.current_stream(eng.device))

    def close(self):
        """Destroys the executable graph -- after its last launch has finished: the runtime
        does not keep a destroyed graph alive for a launch that is still running."""
        if self.graph:
            if self.done is not None:
                self.done.synchronize()
            self.eng.lib.scoary_graph_destroy(self.graph)
            self.graph = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class AssociationEngine:
    def __init__(self, device=None):
        torch = _torch()
        if not torch.cuda.is_available():
            raise _abi.ScoaryHipError(
                "no GPU visible: scoary_amd runs on MI355X (gfx950) only and has no CPU path")
        self.lib = _abi.load()
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", int(device) if not isinstance(device, torch.device)
                                   else device.index or 0)
        h = ctypes.c_void_p()
        rc = self.lib.scoary_create(self.device.index, ctypes.byref(h))
        if rc != 0:
            raise _abi.ScoaryHipError("scoary_create(device=%d) failed: %d" % (self.device.index, rc))
        self.h = h
        # dist.LabelShards: generate one share of every batch of label tiles and all-gather the
        # rest (multi-GPU, opt-in); None: every rank generates all tiles (the default)
        self.label_shards = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.scoary_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- plumbing -----------------------------------------------------------
    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            self._side = _torch().cuda.Stream(device=self.device)
        return self._side

    def _stream(self):
        return ctypes.c_void_p(_torch().cuda.current_stream(self.device).cuda_stream)

    def _check(self, rc, what):
        if rc != 0:
            msg = self.lib.scoary_last_error(self.h)
            raise _abi.ScoaryHipError("%s failed (%d): %s" % (what, rc, (msg or b"").decode()))

    @staticmethod
    def _ptr(t):
        return ctypes.c_void_p(t.data_ptr())

    def quads(self, N):
        return int(self.lib.scoary_tiled_quads(int(N)))

    def row_words(self, N):
        return int(self.lib.scoary_row_words(int(N)))

    def padded_genes(self, G):
        return int(self.lib.scoary_tiled_genes(int(G)))

    def _empty(self, shape, dtype):
        return _torch().empty(shape, dtype=dtype, device=self.device)

    # -- a1: packing ----------------------------------------------------------
    def pack_dense(self, dense):
        """dense: (G, N) uint8 (numpy or device tensor), non-zero = present."""
        torch = _torch()
        if isinstance(dense, np.ndarray):
            dense = torch.from_numpy(np.ascontiguousarray(dense, dtype=np.uint8))
        dense = dense.to(self.device, dtype=torch.uint8).contiguous()
        G, N = dense.shape
        tiled = self._empty((self.quads(N), self.padded_genes(G), 4), torch.int32)
        self._check(self.lib.scoary_pack_dense(self.h, self._ptr(dense), G, N, self._ptr(tiled),
                                               self._stream()), "scoary_pack_dense")
        return GeneMatrix(tiled, G, N)

    def tile_rows(self, rows64, N):
        """rows64: (G, W64) uint64 bit rows (numpy or int64 device tensor)."""
        torch = _torch()
        if isinstance(rows64, np.ndarray):
            rows64 = torch.from_numpy(np.ascontiguousarray(rows64).view(np.int64))
        rows64 = rows64.to(self.device).contiguous()
        G, W = rows64.shape
        if W != (N + 63) // 64:
            raise ValueError("rows64 has %d words per row, N=%d needs %d" % (W, N, (N + 63) // 64))
        tiled = self._empty((self.quads(N), self.padded_genes(G), 4), torch.int32)
        self._check(self.lib.scoary_tile_rows(self.h, self._ptr(rows64), G, N, self._ptr(tiled),
                                              self._stream()), "scoary_tile_rows")
        return GeneMatrix(tiled, G, N)

    def vecrows(self, rows64, N):
        """(R, W64) uint64 host rows -> device vecrows int32 [R, Wp] (zero padded)."""
        torch = _torch()
        rows64 = np.ascontiguousarray(rows64, dtype=np.uint64)
        R, W = rows64.shape
        Wp = self.row_words(N)
        buf = np.zeros((R, Wp), dtype=np.uint32)
        buf[:, :2 * W] = rows64.view(np.uint32).reshape(R, 2 * W)
        return torch.from_numpy(buf.view(np.int32)).to(self.device)

    def list_budget_bytes(self):
        """Bytes build_lists may spend on one matrix's index array: SCOARY_LIST_BUDGET_MB if set
        (tests), else 60 % of the device memory that is free right now (the driver's figure plus what
        torch's caching allocator holds unused) -- label tiles (<= 8 GB)
        and the count scratch (<= 4 GB) of a step still have to fit next to it."""
        import os
        mb = os.environ.get("SCOARY_LIST_BUDGET_MB")
        if mb:
            return int(float(mb) * (1 << 20))
        torch = _torch()
        free, _total = torch.cuda.mem_get_info(self.device)
        # blocks the caching allocator holds but has not handed out are as good as free
        cached = torch.cuda.memory_reserved(self.device) - torch.cuda.memory_allocated(self.device)
        return int((free + max(cached, 0)) * 0.6)

    def list_kernel_name(self, N):
        """Timer label (scoary_last_kernel_ms) of the list-driven permutation kernel that
        takes N isolates: k_permute_lists (one tile in LDS) or k_permute_seglists (N > 20479)."""
        return "k_permute_seglists" if int(self.lib.scoary_list_segments(int(N))) > 1 else "k_permute_lists"

    def build_lists(self, genes, budget_bytes=None):
        """Attach the minority index lists of ``genes`` (spec S6) for the list-driven
        permutation kernel.  Built on the device from the tiled matrix that is
        already in HBM (scoary_lists_plan + scoary_lists_fill): nothing crosses
        PCIe but the 8-byte entry count.  The plan returns the size of the index array
        before it is allocated: more than ``budget_bytes`` (default list_budget_bytes())
        raises ListMemoryError and leaves ``genes.lists`` None."""
        torch = _torch()
        lanes, _stride, _gpw, _classes, _piece = self.list_params(genes.N)
        if not lanes:
            raise ValueError("N=%d is too large for the list-driven kernel" % genes.N)
        G, N = genes.G, genes.N
        scratch = self._empty((int(self.lib.scoary_lists_scratch_bytes(G, N)) // 8 + 1,),
                              torch.int64)
        nseg = max(1, int(self.lib.scoary_list_segments(N)))      # N > 20479: sub-lists per segment
        shape = (G,) if nseg == 1 else (nseg, G)
        start = self._empty(shape, torch.int32)
        ngroups = self._empty(shape, torch.int32)
        order = self._empty((G,), torch.int32)
        flipped = self._empty((G,), torch.uint8)
        entries = ctypes.c_int64()
        self._check(self.lib.scoary_lists_plan(
            self.h, self._ptr(genes.tiled), G, N, self._ptr(scratch), self._ptr(start),
            self._ptr(ngroups), self._ptr(order), self._ptr(flipped), ctypes.byref(entries),
            self._stream()), "scoary_lists_plan")
        total = int(entries.value)                     # 32-bit words of index array
        need = 4 * (total + int(self.lib.scoary_lists_slack_entries()))
        if budget_bytes is None:
            budget_bytes = self.list_budget_bytes()
        if need > budget_bytes:
            raise ListMemoryError("index lists of a %d x %d matrix need %.1f MB, budget %.1f MB"
                                  % (G, N, need / 2**20, budget_bytes / 2**20))
        idx = self._empty((total + int(self.lib.scoary_lists_slack_entries()),), torch.int32)
        self._check(self.lib.scoary_lists_fill(
            self.h, self._ptr(genes.tiled), G, N, self._ptr(scratch), self._ptr(order),
            self._ptr(flipped), total, self._ptr(idx), self._stream()), "scoary_lists_fill")
        genes.lists = GeneLists(idx, start, ngroups, order, flipped, total)
        return genes.lists

    def list_params(self, N):
        """(tile row dwords, row stride bytes, genes per wavefront, classes,
        interleave piece) -- scoary_list_params."""
        out = (ctypes.c_int64 * 5)()
        self.lib.scoary_list_params(int(N), out)
        return tuple(int(x) for x in out)

    def lists_supported(self, N):
        return int(N) <= int(self.lib.scoary_list_max_isolates())

    def perm_generate_tiles(self, masks, margins, N, P, perm_base, seed, out=None, trait_base=0,
                            tile_range=None):
        """Label tiles of permutations perm_base .. perm_base + P - 1 (perm_base a multiple of
        32).  ``tile_range`` = (first, count): only these flat (trait, tile) indices of the
        [T][tiles] array are written (scoary_perm_generate_tiles_range) -- one rank's share of a
        run that all-gathers the rest (dist.LabelShards)."""
        torch = _torch()
        T = masks.shape[0]
        if out is None:
            out = self._empty((int(self.lib.scoary_list_tiles_words(N, P, T)),), torch.int32)
        if tile_range is None:
            self._check(self.lib.scoary_perm_generate_tiles(
                self.h, self._ptr(masks), self._ptr(margins), T, N, P, perm_base, trait_base,
                ctypes.c_uint64(seed), self._ptr(out), self._stream()), "scoary_perm_generate_tiles")
        else:
            self._check(self.lib.scoary_perm_generate_tiles_range(
                self.h, self._ptr(masks), self._ptr(margins), T, N, P, perm_base, trait_base,
                ctypes.c_uint64(seed), int(tile_range[0]), int(tile_range[1]), self._ptr(out),
                self._stream()), "scoary_perm_generate_tiles_range")
        return out

    def tiles_per_batch(self, N, P, T):
        """Flat (trait, tile) count and dwords per tile of a batch of P permutations."""
        tw = self.list_params(N)[0]
        return int(T) * (-(-int(P) // (32 * tw))), int(self.lib.scoary_list_tile_words(int(N)))

    def permute_lists_scratch(self, G, T, N, P):
        """Scratch tensor for permute_lists (list-order regions + per-part counts)."""
        torch = _torch()
        nbytes = int(self.lib.scoary_permute_lists_scratch_bytes(G, T, N, P))
        return self._empty(((nbytes + 3) // 4,), torch.int32)

    def permute_lists(self, genes, tiles, crit, margins, P, r, scratch=None, lcrit=None,
                      accumulate=True):
        """r (+)= exceedance counts of P permutations (label tiles ``tiles``).  The regions
        come in gene order (``crit`` from fisher) or, one launch cheaper, in slot order
        (``lcrit`` from fisher(..., lists=...)); accumulate=False overwrites r."""
        L = genes.lists
        T = (lcrit if lcrit is not None else crit).shape[0]
        if scratch is None:
            scratch = self.permute_lists_scratch(genes.G, T, genes.N, P)
        self._check(self.lib.scoary_permute_lists(
            self.h, self._ptr(tiles), self._ptr(L.idx), L.entries, self._ptr(L.start),
            self._ptr(L.ngroups), self._ptr(L.order), self._ptr(L.flipped),
            self._ptr(crit) if lcrit is None else None,
            self._ptr(lcrit) if lcrit is not None else None,
            self._ptr(margins), self._ptr(scratch), genes.G, T, genes.N, P, self._ptr(r),
            1 if accumulate else 0, self._stream()), "scoary_permute_lists")
        return r

    # -- a3: counts -----------------------------------------------------------
    def trait_plan(self, traits, masks, N, out=None):
        """Margins, mask classes and the gathered operand rows of a trait set (scoary_trait_plan),
        once per trait set.  ``out`` = (margins, mask_class, plan buffer) to fill."""
        torch = _torch()
        T = traits.shape[0]
        margins, mask_class, buf = out if out is not None else (
            self._empty((T, 2), torch.int32), self._empty((T,), torch.int32),
            self._empty((int(self.lib.scoary_trait_plan_bytes(int(T), int(N))) // 4,), torch.int32))
        self._check(self.lib.scoary_trait_plan(self.h, self._ptr(traits), self._ptr(masks), T, int(N),
                                               self._ptr(margins), self._ptr(mask_class), self._ptr(buf),
                                               self._stream()), "scoary_trait_plan")
        return TraitPlan(margins, mask_class, buf, traits, masks, N)

    def counts(self, genes, traits, masks, out=None, plan=None):
        """-> (counts int32 [T, G, 4], margins int32 [T, 2]).  ``plan``: the TraitPlan of these
        traits / masks (trait_plan); without one it is built here (three more small launches).
        ``out`` = (counts[, margins, mask_class, plan buffer]): buffers to fill."""
        torch = _torch()
        T = traits.shape[0]
        counts = out[0] if out is not None else self._empty((T, genes.G, 4), torch.int32)
        if plan is None:
            plan = self.trait_plan(traits, masks, genes.N,
                                   out=None if out is None or len(out) < 4 else tuple(out[1:4]))
        elif not plan.fits(traits, masks) or plan.N != genes.N:
            raise ValueError("the trait plan was built from other trait / mask tensors")
        self._check(self.lib.scoary_counts_planned(
            self.h, self._ptr(genes.tiled), self._ptr(plan.buf), self._ptr(plan.margins),
            genes.G, T, genes.N, self._ptr(counts), self._stream()), "scoary_counts_planned")
        return counts, plan.margins

    # -- a5: Fisher -----------------------------------------------------------
    def fisher(self, tables, want_crit=True, out=None, lists=None, lcrit=None):
        """tables: int32 device tensor [..., 4] -> (p, odds, crit) shaped [...].
        With ``lists`` (the GeneLists of the matrix the [T, G, 4] tables were counted on):
        scoary_fisher_lists -- tables visited in list-slot order, and the regions also
        written in the list kernel's form; returns (p, odds, crit, lcrit)."""
        torch = _torch()
        tables = tables.contiguous()
        shape = tables.shape[:-1]
        M = int(np.prod(shape)) if len(shape) else 1
        if out is not None:
            p, odds, crit = out
        else:
            p = self._empty(shape, torch.float64)
            odds = self._empty(shape, torch.float64)
            crit = self._empty(tuple(shape) + (2,), torch.int32) if want_crit else None
        if lists is not None:
            if len(shape) != 2:
                raise ValueError("fisher(lists=...) wants tables shaped [T, G, 4]")
            if lcrit is None:
                lcrit = self._empty(tuple(shape) + (2,), torch.int32)
            self._check(self.lib.scoary_fisher_lists(
                self.h, self._ptr(tables), shape[0], shape[1], self._ptr(lists.order),
                self._ptr(lists.flipped), self._ptr(p), self._ptr(odds),
                self._ptr(crit) if crit is not None else None, self._ptr(lcrit),
                self._stream()), "scoary_fisher_lists")
            return p, odds, crit, lcrit
        self._check(self.lib.scoary_fisher(self.h, self._ptr(tables), M, self._ptr(p),
                                           self._ptr(odds),
                                           self._ptr(crit) if crit is not None else None,
                                           self._stream()), "scoary_fisher")
        return p, odds, crit

    def fisher_scipy(self, tables, p):
        """SciPy's own double for every table of 171 ... fisher_scipy_max_isolates() isolates, written over the
        entries of ``p`` (float64 device tensor shaped like tables[..., 0]; scoary_fisher_scipy): what the command
        line prints.  Returns the number of tables above the maximum (left as they were)."""
        torch = _torch()
        tables = tables.contiguous()
        if not p.is_contiguous() or p.dtype != torch.float64 or tuple(p.shape) != tuple(tables.shape[:-1]):
            raise ValueError("fisher_scipy: p must be a contiguous float64 tensor shaped like the tables")
        M = int(np.prod(tables.shape[:-1])) if len(tables.shape) > 1 else 1
        skipped = torch.zeros(1, dtype=torch.int64, device=self.device)
        self._check(self.lib.scoary_fisher_scipy(self.h, self._ptr(tables), M, self._ptr(p), self._ptr(skipped),
                                                 self._stream()), "scoary_fisher_scipy")
        return int(skipped.item())

    def fisher_scipy_max_isolates(self):
        return int(self.lib.scoary_fisher_scipy_max_isolates())

    # -- a8 / a7: permutations -------------------------------------------------
    def perm_generate(self, masks, margins, N, P, perm_base, seed, out=None, trait_base=0):
        torch = _torch()
        T = masks.shape[0]
        Wp = self.row_words(N)
        if out is None:
            out = self._empty((T, P, Wp), torch.int32)
        self._check(self.lib.scoary_perm_generate(self.h, self._ptr(masks), self._ptr(margins), T,
                                                  N, P, perm_base, trait_base,
                                                  ctypes.c_uint64(seed),
                                                  self._ptr(out), self._stream()),
                    "scoary_perm_generate")
        return out

    def permute(self, genes, perms, crit, r, P=None):
        T = perms.shape[0]
        if P is None:
            P = perms.shape[1]
        self._check(self.lib.scoary_permute(self.h, self._ptr(genes.tiled), self._ptr(perms),
                                            self._ptr(crit), genes.G, T, genes.N, P,
                                            self._ptr(r), self._stream()), "scoary_permute")
        return r

    def permute_sequential(self, genes, masks, margins, crit, permutations, seed, thr):
        """The reference's sequential estimator with early abort on the Fisher statistic
        (scoary_permute_seq): returns (r, nstop) int32 device tensors [T, G]; Empirical_p =
        (r + 1) / ((nstop or P) + 1).  ``thr``: host array of abort thresholds per
        permutation index (tree._abort_thresholds)."""
        torch = _torch()
        T = masks.shape[0]
        r = torch.zeros((T, genes.G), dtype=torch.int32, device=self.device)
        nstop = torch.zeros((T, genes.G), dtype=torch.int32, device=self.device)
        th = np.minimum(np.asarray(thr, dtype=np.int64), 0xffffffff).astype(np.uint32)
        d_thr = torch.from_numpy(th.view(np.int32)).to(self.device)
        batch = self.perm_batch(T, genes.N, permutations, budget_bytes=1 << 30)
        buf = self._empty((T, batch, self.row_words(genes.N)), torch.int32)
        done = 0
        while done < permutations:
            nb = min(batch, permutations - done)
            self.perm_generate(masks, margins, genes.N, nb, done, seed, out=buf)
            self._check(self.lib.scoary_permute_seq(
                self.h, self._ptr(genes.tiled), self._ptr(buf), self._ptr(crit), self._ptr(d_thr),
                genes.G, T, genes.N, nb, done, self._ptr(r), self._ptr(nstop), self._stream()),
                "scoary_permute_seq")
            done += nb
        return r, nstop

    def perm_batch(self, T, N, P, budget_bytes=8 << 30):
        """Permutations per generate/permute round so the label buffer stays
        under budget_bytes."""
        per = T * self.row_words(N) * 4
        return int(max(1, min(P, budget_bytes // max(per, 1))))

    # -- the whole hot path ----------------------------------------------------
    def list_batch(self, T, N, permutations, G=0, budget_bytes=8 << 30, scratch_bytes=4 << 30):
        """Permutations per label-tile batch of the list-driven path (multiple of 512): the
        label tiles stay under budget_bytes and the per-(trait, tile, gene) 16-bit counts of
        scoary_permute_lists under scratch_bytes (one count per 512 / 256 / 128 ... permutations,
        trait and gene: 9.8 GB for a cfg5 shard in one batch)."""
        per = max(int(self.lib.scoary_list_tiles_words(N, 512, T)) * 4, 1)    # tile bytes / 512 perms
        batch = (budget_bytes // per) * 512
        if G > 0:
            tile_perms = 32 * self.list_params(N)[0]
            per_tile = 2 * int(T) * (int(G) + 64)                              # count bytes / tile column
            batch = min(batch, max(1, scratch_bytes // per_tile) * tile_perms // 512 * 512)
        return int(max(512, min(-(-permutations // 512) * 512, batch)))

    def workspace(self, genes, T, permutations=0, use_lists=None, perm_buffer=None):
        """Every device buffer one associate() step needs, allocated once: steps that
        reuse it allocate nothing (a precondition for hipGraph capture, and what
        small launch-bound workloads need anyway)."""
        return Workspace(self, genes, T, permutations, use_lists, perm_buffer)

    # -- launch-bound steps: automatic hipGraph replay -----------------------------
    AUTO_GRAPH_MAX_TESTS = 5e8      # ~0.5 ms of kernels at 1e12 tests/s: below it launches dominate

    def auto_graph_eligible(self, genes, T, permutations):
        """A step this small is bound by its five kernel launches (cfg2: 0.112 ms eager, 0.064 ms
        as one graph launch, profiles/r03_bench_cfg2*.json): associate() then records it into a
        hipGraph on its second call with the same buffers and replays it from the third on.
        The recording call synchronises the device once (capture() warms the step up and records
        on a fresh stream); SCOARY_AUTO_GRAPH=0 switches the whole mechanism off."""
        import os
        if os.environ.get("SCOARY_AUTO_GRAPH", "1") == "0":
            return False
        return float(genes.G) * int(T) * max(int(permutations), 1) <= self.AUTO_GRAPH_MAX_TESTS

    def _auto_graph(self, genes, traits, masks, permutations, seed, use_lists, ws, plan):
        """The cached-graph path of associate(): returns the result dict, or None for 'run eagerly'."""
        torch = _torch()
        if getattr(self, "_timing", False) or torch.cuda.is_current_stream_capturing():
            return None
        L = genes.lists
        # Everything a recorded step has baked in: the tensors themselves (compared by identity
        # and kept alive by `refs`, so their memory cannot be recycled under the graph), their
        # versions (in-place edits through torch since the recording) and the scalars.
        refs = (genes.tiled, traits, masks, plan, plan.margins, plan.mask_class, plan.buf) + \
            ((L.idx, L.start, L.ngroups, L.order, L.flipped) if L is not None else ())
        scalars = (genes.G, genes.N, int(traits.shape[0]), int(permutations), int(seed), bool(use_lists),
                   tuple(getattr(x, "_version", 0) for x in refs))
        st = ws.auto
        same = st is not None and st["scalars"] == scalars and len(st["refs"]) == len(refs) and \
            all(a is b for a, b in zip(st["refs"], refs))
        if not same:
            if st is not None and st["graph"] is not None:
                st["graph"].close()                       # waits for its last launch first
            ws.auto = {"refs": refs, "scalars": scalars, "graph": None, "res": None}
            return None                                   # first call with these buffers: eager
        if st["graph"] is None:                           # second call: record (runs the step as well)
            st["graph"], st["res"] = self.capture(genes, traits, masks, permutations, seed, ws,
                                                  use_lists=use_lists, plan=plan)
            # the capture ran on its own stream: order the caller's stream behind it
            torch.cuda.current_stream(self.device).wait_stream(st["graph"].stream)
            return st["res"]
        st["graph"].launch()
        return st["res"]

    def _label_tiles(self, ws, masks, margins, N, nb, base, seed):
        """One batch of label tiles into ws.tiles: all of them, or -- with label shards -- this
        rank's share followed by the all-gather of the others'."""
        sh = ws.label_shards
        if sh is None or sh.world == 1:
            self.perm_generate_tiles(masks, margins, N, nb, base, seed, out=ws.tiles)
            return
        nflat, tile_words = self.tiles_per_batch(N, nb, masks.shape[0])
        _per, first, count = sh.share(nflat)
        self.perm_generate_tiles(masks, margins, N, nb, base, seed, out=ws.tiles,
                                 tile_range=(first, count))
        sh.all_gather(ws.tiles, nflat, tile_words)

    def associate(self, genes, traits, masks, permutations=0, seed=0, perm_buffer=None,
                  use_lists=None, workspace=None, plan=None, graph=None, records=None):
        """counts -> Fisher -> (optional) permutation exceedance counts.
        Returns dict of device tensors: counts [T,G,4], margins [T,2],
        p / odds [T,G], r [T,G] (uint32 bit pattern in int32) or None.  With
        ``workspace`` the result tensors are the workspace's (overwritten by the
        next step that uses it).  ``plan``: the TraitPlan of these traits (trait_plan, once
        per trait set); without one every step rebuilds it (one more small launch).  With a
        workspace AND a plan, a launch-bound step (auto_graph_eligible) is recorded into a
        hipGraph on its second call and replayed afterwards; ``graph=False`` keeps it eager.
        ``records``: an int32 [T, G, 10] device tensor -- the result is also packed into it
        (pack_records) and returned as res["records"]."""
        res = self._associate(genes, traits, masks, permutations, seed, perm_buffer, use_lists,
                              workspace, plan, graph if records is None else False)
        if records is not None:
            # the exchange records of the step, packed as its last kernel (inside a captured step:
            # one launch less per replay for a gene-sharded rank)
            res = dict(res)
            res["records"] = self.pack_records(res, out=records)
        return res

    def _associate(self, genes, traits, masks, permutations, seed, perm_buffer, use_lists, workspace,
                   plan, graph):
        """The step behind associate() (its docstring)."""
        torch = _torch()
        T = traits.shape[0]
        if use_lists is None:
            use_lists = genes.lists is not None and self.lists_supported(genes.N)
        if use_lists and permutations > 0 and genes.lists is None:
            raise ValueError("use_lists=True but the gene matrix has no index lists: call "
                             "build_lists(genes) once per data set first")
        ws = workspace
        if ws is None:
            ws = Workspace(self, genes, T, permutations, use_lists, perm_buffer)
        elif not ws.fits(genes, T, permutations, use_lists):
            raise ValueError("workspace was made for another problem shape")
        # launch-bound steps with persistent buffers (workspace + plan): replay a cached hipGraph
        # (graph=None: automatic; False: never -- capture() itself, per-kernel timing)
        if graph is None and workspace is not None and plan is not None and perm_buffer is None \
                and ws.label_shards is None and self.auto_graph_eligible(genes, T, permutations):
            res = self._auto_graph(genes, traits, masks, permutations, seed, use_lists, ws, plan)
            if res is not None:
                return res
        counts, margins = self.counts(genes, traits, masks,
                                      out=(ws.counts, ws.margins, ws.mask_class, ws.plan_buf), plan=plan)
        if permutations > 0 and use_lists:
            # The first batch of label tiles needs only the trait margins, not the
            # Fisher pass: generate it on a side stream while k_fisher runs.  (With a plan the
            # margins are there before k_counts, and the fork could move in front of it: tried,
            # worth 0.2 % at cfg3 and nothing on the launch-bound shapes, while k_counts then
            # shares the chip with the generator and its own duration -- the path's one HBM
            # stream, reported as roofline_k1 -- can no longer be read off the step.)
            # (Round 5, with the 0.04-0.08 ms generator: the overlap is still worth 0.3 % at cfg3 and
            # 2.5 % on a 25 000-gene shard of cfg4, nothing on cfg4 itself; SCOARY_GEN_SIDE_STREAM=0
            # runs the two back to back on the main stream for such A/B runs.)
            import os
            main = torch.cuda.current_stream(self.device)
            nb0 = min(ws.batch, permutations)
            if os.environ.get("SCOARY_GEN_SIDE_STREAM", "1") == "1":
                side = self._side_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._label_tiles(ws, masks, margins, genes.N, nb0, 0, seed)
                p, odds, crit, lcrit = self.fisher(counts, out=(ws.p, ws.odds, ws.crit),
                                                   lists=genes.lists, lcrit=ws.lcrit)
                main.wait_stream(side)
            else:
                self._label_tiles(ws, masks, margins, genes.N, nb0, 0, seed)
                p, odds, crit, lcrit = self.fisher(counts, out=(ws.p, ws.odds, ws.crit),
                                                   lists=genes.lists, lcrit=ws.lcrit)
            done = 0
            while done < permutations:
                nb = min(ws.batch, permutations - done)
                if done > 0:
                    self._label_tiles(ws, masks, margins, genes.N, nb, done, seed)
                self.permute_lists(genes, ws.tiles, None, margins, nb, ws.r, scratch=ws.scratch,
                                   lcrit=lcrit, accumulate=done > 0)
                done += nb
            return {"counts": counts, "margins": margins, "p": p, "odds": odds, "crit": crit,
                    "r": ws.r}
        p, odds, crit = self.fisher(counts, out=(ws.p, ws.odds, ws.crit))
        r = None
        if permutations > 0:
            r = ws.r
            r.zero_()
            batch = ws.perms.shape[1]
            done = 0
            while done < permutations:
                nb = min(batch, permutations - done)
                self.perm_generate(masks, margins, genes.N, nb, done, seed, out=ws.perms)
                self.permute(genes, ws.perms[:, :nb] if nb == batch else ws.perms, crit, r, P=nb)
                done += nb
        return {"counts": counts, "margins": margins, "p": p, "odds": odds, "crit": crit, "r": r}

    def capture(self, genes, traits, masks, permutations, seed, workspace, use_lists=None, plan=None,
                records=None):
        """Record one associate() step into a hipGraph (scoary_graph_*): returns
        (StepGraph, result dict).  The results live in ``workspace``; ``launch()``
        recomputes them with a single graph launch.  The step is run once eagerly
        first (kernel attributes, side stream and lazy module loads happen outside the
        capture)."""
        torch = _torch()
        # checked before anything is launched: the warm-up step below ends in a device
        # synchronize, which is illegal on a capturing stream
        if torch.cuda.is_current_stream_capturing():
            raise _abi.ScoaryHipError("capture(): the current stream is already capturing")
        if workspace.label_shards is not None and workspace.label_shards.world > 1:
            raise _abi.ScoaryHipError("capture(): label shards are active -- their all-gather is a collective "
                                      "and cannot be recorded into a hipGraph")
        self.associate(genes, traits, masks, permutations=permutations, seed=seed,
                       use_lists=use_lists, workspace=workspace, plan=plan, graph=False, records=records)
        torch.cuda.synchronize(self.device)
        stream = torch.cuda.Stream(device=self.device)      # a fresh stream: never mid-capture
        with torch.cuda.stream(stream):
            self._check(self.lib.scoary_graph_begin(self.h, self._stream()), "scoary_graph_begin")
            failed = True
            try:
                res = self.associate(genes, traits, masks, permutations=permutations, seed=seed,
                                     use_lists=use_lists, workspace=workspace, plan=plan, graph=False,
                                     records=records)
                failed = False
            finally:
                # the capture must be ended either way; a graph that came out of a failed step
                # is destroyed here instead of leaking with the exception
                g = ctypes.c_void_p()
                rc = self.lib.scoary_graph_end(self.h, self._stream(), ctypes.byref(g))
                if (failed or rc != 0) and g.value:
                    self.lib.scoary_graph_destroy(g)
                    g = ctypes.c_void_p()
            self._check(rc, "scoary_graph_end")
        return StepGraph(self, g, stream), res

    def pack_records(self, res, nstop=None, out=None):
        """The exchange records of one associate() result (scoary_pack_records): int32
        device tensor [T, G, 10] = counts, p, odds, r, nstop -- the layout of
        scoary_amd.dist.pack_records, produced by one kernel."""
        torch = _torch()
        T, G = res["p"].shape
        if out is None:
            out = self._empty((T, G, 10), torch.int32)
        r = res.get("r")
        self._check(self.lib.scoary_pack_records(
            self.h, self._ptr(res["counts"]), self._ptr(res["p"]), self._ptr(res["odds"]),
            self._ptr(r) if r is not None else None,
            self._ptr(nstop) if nstop is not None else None, T * G, self._ptr(out),
            self._stream()), "scoary_pack_records")
        return out

    # -- --collapse support -----------------------------------------
    def row_hash(self, genes, masks):
        """(T, G, 2) uint64 numpy: 128-bit hash of every gene row AND each
        trait's validity mask."""
        torch = _torch()
        T = masks.shape[0]
        out = self._empty((T, genes.G, 2), torch.int64)
        self._check(self.lib.scoary_row_hash(self.h, self._ptr(genes.tiled), self._ptr(masks),
                                             genes.G, T, genes.N, self._ptr(out), self._stream()),
                    "scoary_row_hash")
        return out.cpu().numpy().view(np.uint64)

    # -- population-structure stage ----------------------------------
    def upgma_merges(self, rows01):
        """rows01: (R, N) 0/1 numpy (rows = isolates, columns = variable genes) -> the
        reference's UPGMA merge order as an (R-1, 2) int32 numpy array, or None when the
        device loop met the degenerate case it hands back (scoary_upgma).  Hamming counts,
        distances and the merge loop stay on the device; only the merge list comes back."""
        torch = _torch()
        R, N = rows01.shape
        counts = self.hamming(rows01, device=True)
        scratch = self._empty(((int(self.lib.scoary_upgma_scratch_bytes(R)) + 7) // 8,), torch.int64)
        merges = self._empty((max(R - 1, 1), 2), torch.int32)
        status = self._empty((1,), torch.int32)
        self._check(self.lib.scoary_upgma(self.h, self._ptr(counts), R, N, self._ptr(scratch),
                                          self._ptr(merges), self._ptr(status), self._stream()),
                    "scoary_upgma")
        if int(status.cpu()[0]) != 0:
            return None
        return merges.cpu().numpy()[:R - 1]

    def hamming(self, rows01, device=False):
        """rows01: (R, N) 0/1 numpy (rows = isolates, columns = variable genes)
        -> (R, R) int32 pairwise Hamming counts (numpy, or the device tensor)."""
        torch = _torch()
        rows01 = np.ascontiguousarray(rows01, dtype=np.uint8)
        R, N = rows01.shape
        bits = pack_bits_rows(rows01)
        gm = self.tile_rows(bits, N)
        vec = self.vecrows(bits, N)
        out = self._empty((R, R), torch.int32)
        self._check(self.lib.scoary_hamming(self.h, self._ptr(gm.tiled), self._ptr(vec), R, N,
                                            self._ptr(out), self._stream()), "scoary_hamming")
        return out if device else out.cpu().numpy()

    def gather_bits(self, rows, index):
        """rows: int32 device tensor [R, Wsrc] of bit rows; index: int32 device
        tensor [K] -> int32 [R, ceil(K/32)] with bit k = source bit index[k]."""
        torch = _torch()
        rows = rows.contiguous()
        R, Wsrc = rows.shape
        K = int(index.shape[0])
        out = self._empty((R, (K + 31) // 32), torch.int32)
        self._check(self.lib.scoary_gather_bits(self.h, self._ptr(rows), R, Wsrc,
                                                self._ptr(index), K, self._ptr(out),
                                                self._stream()), "scoary_gather_bits")
        return out

    def tree_pairs(self, ops, depth, gene_bits, label_bits, K):
        torch = _torch()
        G, L = gene_bits.shape[0], label_bits.shape[0]
        out = self._empty((G, L, 3), torch.int32)
        self._check(self.lib.scoary_tree_pairs(self.h, self._ptr(ops), int(ops.shape[0]),
                                               int(depth), self._ptr(gene_bits),
                                               self._ptr(label_bits), G, L, int(K),
                                               self._ptr(out), self._stream()),
                    "scoary_tree_pairs")
        return out

    def tree_permute(self, ops, depth, gene_bits, label_bits, K, obs):
        torch = _torch()
        G, L = gene_bits.shape[0], label_bits.shape[0]
        out = self._empty((G, L), torch.uint8)
        self._check(self.lib.scoary_tree_permute(self.h, self._ptr(ops), int(ops.shape[0]),
                                                 int(depth), self._ptr(gene_bits),
                                                 self._ptr(label_bits), G, L, int(K),
                                                 self._ptr(obs.contiguous()), self._ptr(out),
                                                 self._stream()), "scoary_tree_permute")
        return out

    # -- timing (bench.py) ------------------------------------------------------
    def set_timing(self, on):
        self._check(self.lib.scoary_set_timing(self.h, 1 if on else 0), "scoary_set_timing")
        self._timing = bool(on)          # per-kernel events: steps run eagerly (no graph replay)

    def kernel_ms(self, name):
        ms = ctypes.c_double()
        self._check(self.lib.scoary_last_kernel_ms(self.h, name.encode(), ctypes.byref(ms)),
                    "scoary_last_kernel_ms")
        return ms.value
