"""Host-side mirror of the reference's interface for the association path.

Same entry points, argument meaning and result shapes as scoary/methods.py of
AdmiralenOla/Scoary v1.6.16 -- but array-based inside, with every count,
Fisher test and permutation evaluated by the HIP library (scoary_amd.engine):

  Csv_to_dic_Roary   scoary/methods.py:335-508   gene presence/absence reader
  Csv_to_dic         scoary/methods.py:546-614   traits reader
  Perform_statistics scoary/methods.py:930-982   one (gene, trait) 2x2 table
  Setup_results      scoary/methods.py:757-928   counts + Fisher + B/BH per trait
  Permute            scoary/methods.py:1314-1369 empirical p (Fisher statistic,
                                                 SURVEY D1 / DESIGN.md)
  StoreResults / StoreTraitResult  :987-1197     per-trait results CSV
  ScoaryArgumentParser / main      :1551-1744, :49-330   the CLI

There is no CPU fallback: without the GPU library these functions raise.
"""
import argparse
import csv
import logging
import os
import re
import sys
import time as _time
from collections.abc import Mapping

import numpy as np

from . import SCOARY_COMPAT_VERSION, __version__
from .engine import AssociationEngine, pack_bits_rows

log = logging.getLogger("scoary_amd")
log.setLevel(logging.DEBUG)

ABSENT_CELLS = ("", "0", "-")                     # methods.py:476
MISSING_TRAIT = ("NA", "-", ".", " ", "")         # methods.py:592
ALLOWED_TRAIT = ("0", "1") + MISSING_TRAIT        # methods.py:576
ROARY_HEAD = ["Gene", "Non-unique Gene name", "Annotation"]
ROARY_COLS = set(ROARY_HEAD + [
    "No. isolates", "No. sequences", "Avg sequences per isolate", "Genome Fragment",
    "Order within Fragment", "Accessory Fragment", "Accessory Order with Fragment", "QC",
    "Min group size nuc", "Max group size nuc", "Avg group size nuc",
    "Order within fragment", "Genome fragment", "Accessory fragment"])

DEFAULT_SEED = 0x5C0A27


# ---------------------------------------------------------------------------
# Input containers
# ---------------------------------------------------------------------------
class GeneTable(Mapping):
    """The reference's ``genedic`` (gene id -> {strain: 0/1, "Non-unique Gene
    name", "Annotation", "<col>_name"}) held as arrays: ordered ids, metadata
    columns and bit-packed presence rows (rows64, spec S1).  Behaves as a
    read-only mapping with the reference's shape, so code written against
    ``genedic[gene][strain]`` keeps working."""

    def __init__(self, ids, nugn, annotation, strains, rows64, extra=None):
        self.ids = list(ids)
        self.nugn = list(nugn)
        self.annotation = list(annotation)
        self.strains = list(strains)
        self.rows64 = np.ascontiguousarray(rows64, dtype=np.uint64)
        self.extra = extra or {}                 # "<col>_name" -> list per gene
        self._index = {g: i for i, g in enumerate(self.ids)}
        self._device = None                      # (engine, GeneMatrix) cache

    @classmethod
    def from_genedic(cls, genedic, strains=None):
        """Pack a plain reference-style dict of dicts."""
        ids = list(genedic.keys())
        meta = ("Non-unique Gene name", "Annotation")
        if strains is None:
            first = genedic[ids[0]] if ids else {}
            strains = [k for k in first if k not in meta and not str(k).endswith("_name")]
        dense = np.zeros((len(ids), len(strains)), dtype=np.uint8)
        for i, g in enumerate(ids):
            row = genedic[g]
            dense[i] = [1 if row[s] else 0 for s in strains]
        extra_keys = [k for k in (genedic[ids[0]] if ids else {}) if str(k).endswith("_name")]
        extra = {k: [genedic[g].get(k, "") for g in ids] for k in extra_keys}
        return cls(ids, [genedic[g].get(meta[0], "") for g in ids],
                   [genedic[g].get(meta[1], "") for g in ids], strains,
                   pack_bits_rows(dense), extra)

    # Mapping interface --------------------------------------------------
    def __len__(self):
        return len(self.ids)

    def __iter__(self):
        return iter(self.ids)

    def __getitem__(self, gene):
        i = self._index[gene]
        bits = np.unpackbits(self.rows64[i].view(np.uint8), bitorder="little")
        row = {"Non-unique Gene name": self.nugn[i], "Annotation": self.annotation[i]}
        for j, s in enumerate(self.strains):
            row[s] = int(bits[j])
        for k, col in self.extra.items():
            row[k] = col[i]
        return row

    def index(self, gene):
        return self._index[gene]

    def dense(self):
        """(G, N) uint8 presence matrix."""
        N = len(self.strains)
        by = self.rows64.view(np.uint8).reshape(len(self.ids), -1)
        return np.unpackbits(by, axis=1, bitorder="little")[:, :N]

    def on_device(self, engine):
        if self._device is None or self._device[0] is not engine:
            self._device = (engine, engine.tile_rows(self.rows64, len(self.strains)))
        return self._device[1]


def _trait_arrays(traitsdic, strains):
    """traitsdic (trait -> {strain: "0"/"1"}, missing isolates absent) ->
    names, (T, N) uint8 with 2 = missing.  A strain named in a trait but not in
    the gene table is the reference's fatal KeyError path (methods.py:974-980)."""
    names = list(traitsdic.keys())
    col = {s: j for j, s in enumerate(strains)}
    arr = np.full((len(names), len(strains)), 2, dtype=np.uint8)
    for t, name in enumerate(names):
        for s, v in traitsdic[name].items():
            if s not in col:
                log.critical("CRITICAL: Could not find %s in the genes file." % str(s))
                sys.exit("Make sure strains are named the same in your traits file as in "
                         "your gene presence/absence file")
            if v in ("0", "1", 0, 1):
                arr[t, col[s]] = int(v)
            elif v in MISSING_TRAIT or v == "NA":
                arr[t, col[s]] = 2
            else:
                sys.exit("There was a problem with comparing your traits and gene "
                         "presence/absence files: unexpected trait value %r" % (v,))
    return names, arr


class TraitResults(Mapping):
    """``Results[trait]``: gene -> row dict, array-backed (one numpy column per
    field, rows materialised on access).  Keys are the testable genes in the
    reference's insertion order."""

    FIELDS = ("tpgp", "tngp", "tpgn", "tngn", "sens", "spes", "OR", "p_v", "B_p", "BH_p")

    def __init__(self, genes, nugn, annotation, cols, number_of_tests, members=None, table=None,
                 rows_idx=None):
        """Either explicit name lists (``genes`` / ``nugn`` / ``annotation``: the --collapse units,
        plain reference dicts) or -- the common case -- ``table`` + ``rows_idx``: row i is gene
        ``rows_idx[i]`` of the GeneTable and its names are looked up when asked for.  A result set
        of G genes x T traits then costs T index arrays, not G x T Python strings; only the rows
        that are written (or read through the mapping interface) are ever materialised."""
        self._table, self._rows_idx = table, rows_idx
        self._genes = None if table is not None else list(genes)
        self._nugn = None if table is not None else list(nugn)
        self._annotation = None if table is not None else list(annotation)
        self.cols = cols                         # field -> ndarray
        self.number_of_tests = number_of_tests
        self.members = members                   # collapse: row -> list of source gene ids
        self._index_cache = None
        self.p_order = None                      # stable argsort of cols["p_v"], if already known

    # names of one row: O(1), nothing materialised
    def gene_at(self, i):
        return self._genes[i] if self._genes is not None else self._table.ids[self._rows_idx[i]]

    def nugn_at(self, i):
        return self._nugn[i] if self._nugn is not None else self._table.nugn[self._rows_idx[i]]

    def annotation_at(self, i):
        return self._annotation[i] if self._annotation is not None \
            else self._table.annotation[self._rows_idx[i]]

    def source_index(self, i):
        """Row of the GeneTable behind result row i (the last member of a --collapse unit)."""
        if self._rows_idx is not None:
            return int(self._rows_idx[i])
        return None

    # whole name lists (the reference's dict view): built on first use
    @property
    def genes(self):
        if self._genes is None:
            ids = self._table.ids
            self._genes = [ids[i] for i in self._rows_idx]
        return self._genes

    @property
    def nugn(self):
        if self._nugn is None:
            v = self._table.nugn
            self._nugn = [v[i] for i in self._rows_idx]
        return self._nugn

    @property
    def annotation(self):
        if self._annotation is None:
            v = self._table.annotation
            self._annotation = [v[i] for i in self._rows_idx]
        return self._annotation

    @property
    def _index(self):
        if self._index_cache is None:
            self._index_cache = {g: i for i, g in enumerate(self.genes)}
        return self._index_cache

    def __len__(self):
        return len(self._rows_idx) if self._genes is None else len(self._genes)

    def __iter__(self):
        return iter(self.genes)

    def __getitem__(self, gene):
        i = self._index[gene]
        row = {"NUGN": self.nugn_at(i), "Annotation": self.annotation_at(i)}
        for k, col in self.cols.items():
            v = col[i]
            row[k] = int(v) if k in ("tpgp", "tngp", "tpgn", "tngn") else \
                (float(v) if k in ("sens", "spes") else v)
        return row

    def column(self, name):
        return self.cols[name]


class GeneTraitCombinations(Mapping):
    """``Gene_trait_combinations[trait]``: gene -> {strain: "AB"|"Ab"|"aB"|"ab"}
    over the trait's valid isolates (methods.py:953-965), computed lazily --
    only the tree stage of the reference consumes it."""

    def __init__(self, table, genes, members, trait_row, trait_index=0):
        """``genes``: a list of names or the trait's TraitResults (names looked up on demand)."""
        self.table, self._names, self.members, self.trait_row = table, genes, members, trait_row
        self.trait_index = trait_index           # enters the permutation counters (spec S4)
        self._index_cache = None

    @property
    def genes(self):
        return self._names.genes if isinstance(self._names, TraitResults) else self._names

    @property
    def _index(self):
        if self._index_cache is None:
            self._index_cache = {g: i for i, g in enumerate(self.genes)}
        return self._index_cache

    def __len__(self):
        return len(self._names)

    def __iter__(self):
        return iter(self.genes)

    def __getitem__(self, gene):
        i = self._index[gene]
        src = self.members[i][-1] if self.members is not None else gene
        gi = self.table.index(src)
        bits = np.unpackbits(self.table.rows64[gi].view(np.uint8), bitorder="little")
        out = {}
        for j, s in enumerate(self.table.strains):
            t = self.trait_row[j]
            if t == 2:
                continue
            out[s] = ("A" if bits[j] else "a") + ("B" if t == 1 else "b")
        return out


# ---------------------------------------------------------------------------
# Stage clock: seconds per named stage of a command-line run ("Stage detail" log line)
# ---------------------------------------------------------------------------
class _Stages:
    def __init__(self):
        self.seconds = {}

    def reset(self):
        self.seconds = {}

    def __call__(self, name):
        return _StageTimer(self, name)


class _StageTimer:
    def __init__(self, owner, name):
        self.owner, self.name = owner, name

    def __enter__(self):
        self.t0 = _time.perf_counter()
        return self

    def __exit__(self, *exc):
        self.owner.seconds[self.name] = self.owner.seconds.get(self.name, 0.0) + _time.perf_counter() - self.t0
        return False


_stage = _Stages()


# ---------------------------------------------------------------------------
# Engine singleton
# ---------------------------------------------------------------------------
_ENGINE = None


def _preload_scipy():
    """Default (pairwise) mode may ask SciPy for a few binomial tails (tree.binom_two_sided_many: the central pairs of
    genes with 79 or more contrasting pairs); importing scipy.stats takes 0.3-0.5 s, so it starts here, on a helper
    thread under the association stage, instead of in the middle of the first trait's pairwise stage."""
    import threading

    def load():
        try:
            import scipy.stats  # noqa: F401
        except ImportError:
            pass
    threading.Thread(target=load, name="scoary-scipy-preload", daemon=True).start()


def get_engine():
    global _ENGINE
    if _ENGINE is None:
        eng = AssociationEngine()
        try:
            _warm_up(eng)
        except Exception as e:      # only an optimisation: a tiny list budget, a failed LDS opt-in or an
            log.debug("engine warm-up skipped: %s: %s" % (type(e).__name__, e))   # OOM must not cost the run
        _ENGINE = eng
    return _ENGINE


def _warm_up(eng):
    """One tiny pass through the path when the engine is created: the first launch of the
    library loads its code object, the first copies set up torch's allocator and staging
    buffers -- tens of milliseconds that belong to process start-up, not to the association
    stage of the first (and usually only) data set."""
    import torch
    g = eng.pack_dense(np.array([[1, 0, 1, 0], [0, 1, 1, 0]], dtype=np.uint8))
    lab = pack_bits_rows(np.array([[1, 1, 0, 0]], dtype=np.uint8))
    val = pack_bits_rows(np.ones((1, 4), dtype=np.uint8))
    trv, mkv = eng.vecrows(lab, 4), eng.vecrows(val, 4)
    eng.build_lists(g)
    res = eng.associate(g, trv, mkv, permutations=32, seed=1)
    eng.pack_records(res).cpu()
    torch.cuda.synchronize(eng.device)


# ---------------------------------------------------------------------------
# Readers
# ---------------------------------------------------------------------------
def _is_roary_file(path, delimiter):
    """Does the table's header start with Roary's three columns (methods.py:373-376)?"""
    try:
        with open(path, "r") as f:
            return next(csv.reader(f, skipinitialspace=True, delimiter=delimiter))[0:3] == ROARY_HEAD
    except (OSError, StopIteration, UnicodeError):
        return True


def Csv_to_dic_Roary(genefile, delimiter, grabcols, startcol=14, allowed_isolates=None,
                     writereducedset=False, time="", outdir="./"):
    """Read a Roary-style gene presence/absence table (methods.py:335-508).

    Returns the reference's dict: "Roarydic" (a GeneTable), "Zero_ones_matrix"
    (strain-major 0/1 lists of the variable genes, for tree building),
    "Strains", "Extracols", "Firstcolnames".  Plain files are tokenised by the
    native streaming reader (scoary_amd/csrc/scoary_io.cpp, same csv dialect);
    other handles, or --include_input_columns reaching into strain columns, go
    through Python's csv module."""
    opened = None
    if writereducedset:
        # under torchrun only rank 0 writes the reduced table; the other ranks wait for it
        # (all ranks writing the same path with mode "w" raced each other's reads)
        from . import dist as _dist
        world, rank = _dist.world_rank()
        name = _reduced_name(outdir, time)
        if world > 1:
            # every rank built `time` from its own clock: ranks that straddle a minute would
            # look for a file rank 0 never wrote -- rank 0's name is the name
            name = _dist.all_gather_objects(name)[0]
        if rank == 0:
            ReduceSet(genefile, delimiter, grabcols, startcol, allowed_isolates, time, outdir)
        if world > 1:
            _dist.barrier()
        opened = open(name, "r", newline=None)
        genefile = opened
    from . import io_native
    path = getattr(genefile, "name", None)
    native = (io_native.available() and isinstance(path, str) and os.path.isfile(path)
              and len(delimiter) == 1 and delimiter not in ' "\r\n'
              and os.environ.get("SCOARY_PY_CSV") != "1"
              and startcol >= 3          # a -s inside the identifier cells (or negative: Python's slices): the csv path
              and grabcols != [-999] and all(0 <= c < startcol for c in grabcols))
    rows_iter = None
    if native:
        try:
            def wanted(hdr):     # identifier, name and annotation columns + the grabbed ones
                try:
                    first = [hdr.index(h) for h in ROARY_HEAD] if hdr[0:3] == ROARY_HEAD else [0, 1, 2]
                except ValueError:
                    first = [0, 1, 2]
                return first + list(grabcols)
            header, meta_rows, bits, kept_native = _read_gpa_ranks(
                io_native, path, delimiter, startcol, allowed_isolates, wanted)
        except io_native.GpaError as e:
            if "startcol" in str(e):
                sys.exit("The startcol (-s) you have specified does not seem to correspond to "
                         "any column in your gene presence/absence file.")
            # the reference's two messages (methods.py:447-462): a data row too short to hold its
            # identifier cells ends a Roary table with the first, a plain table with the second;
            # the reader's own description of the row goes to the log, not into the message
            log.debug("gene presence absence reader: %s" % e)
            short = re.search(r"row \d+ has (\d+) cells", str(e))
            if short and int(short.group(1)) < 3 and not _is_roary_file(path, delimiter):
                sys.exit("CRITICAL: Could not properly assign column. Please report this bug.")
            sys.exit("CRITICAL: Could not read gene presence absence file. Verify that this "
                     "file is a proper Roary file using the specified delimiter (default is "
                     "',').")
    else:
        reader = csv.reader(genefile, skipinitialspace=True, delimiter=delimiter)
        header = next(reader)
        rows_iter = reader
    if grabcols == [-999]:
        grabcols = list(range(3, len(header)))
    if startcol >= len(header):
        sys.exit("The startcol (-s) you have specified does not seem to correspond to any "
                 "column in your gene presence/absence file.")
    strains = header[startcol:]
    extracols = [header[c] for c in grabcols]
    roary = header[0:3] == ROARY_HEAD

    # -s sanity hints (methods.py:382-414): same diagnostics, same conditions
    if roary and strains[0] in ROARY_COLS:
        guess = next((startcol + c for c, s in enumerate(strains) if s not in ROARY_COLS), None)
        log.error("ERROR: Make sure you have set the -s parameter correctly. You are running "
                  "with -s %s. This correponds to the column %s. If this is not an isolate, "
                  "Scoary might crash or produce strange results. Scoary thinks you should have "
                  "run with -s %s instead" % (startcol + 1, strains[0],
                                              (guess + 1) if guess is not None else "?"))
    if roary:
        suspicious = []
        for name in reversed(header[:startcol]):
            if name not in ROARY_COLS:
                suspicious.append(name)
            else:
                if suspicious:
                    log.error("ERROR: Make sure you have set the -s parameter correctly. You "
                              "are running with -s %s. Scoary thinks you should have used %s. "
                              "This excludes the following, which Scoary thinks are isolates: %s"
                              % (startcol + 1, startcol - len(suspicious) + 1,
                                 ", ".join(suspicious)))
                break

    keep = [c for c, s in enumerate(strains)
            if allowed_isolates is None or s in allowed_isolates]
    kept_strains = [strains[c] for c in keep]
    if roary:
        try:
            genecol, nugcol, anncol = (header.index(h) for h in ROARY_HEAD)
        except ValueError:
            genecol, nugcol, anncol = 0, 1, 2
        firstcolnames = list(ROARY_HEAD)
    else:
        genecol, nugcol, anncol = 0, 1, 2
        firstcolnames = header[0:3]

    # a repeated identifier replaces the earlier row but keeps its position
    # (dict overwrite in the reference)
    index, ids, nugn, ann, source = {}, [], [], [], []
    extra = {header[c] + "_name": [] for c in grabcols}
    dense_rows = []
    all_rows = []        # presence of every FILE row (the tree stage uses these, see the return)

    def take(q, r, present):
        try:
            ident = q[genecol] if roary else "_|_".join((q[genecol], q[nugcol], q[anncol]))
            meta = (q[nugcol], q[anncol])
            grabbed = [q[c] for c in grabcols]
        except IndexError:
            if not roary:
                sys.exit("CRITICAL: Could not properly assign column. Please report this bug.")
            sys.exit("CRITICAL: Could not read gene presence absence file. Verify that this "
                     "file is a proper Roary file using the specified delimiter (default is ',').")
        if ident in index:
            i = index[ident]
            nugn[i], ann[i], source[i] = meta[0], meta[1], r
            if present is not None:
                dense_rows[i] = present
            for c, v in zip(grabcols, grabbed):
                extra[header[c] + "_name"][i] = v
        else:
            index[ident] = len(ids)
            ids.append(ident)
            nugn.append(meta[0])
            ann.append(meta[1])
            source.append(r)
            if present is not None:
                dense_rows.append(present)
            for c, v in zip(grabcols, grabbed):
                extra[header[c] + "_name"].append(v)

    if native:
        if startcol > max(genecol, nugcol, anncol, *grabcols, 2):
            # the native reader's rows all have startcol cells: the dict semantics of the reference
            # (a repeated identifier keeps its first position and takes the last row's content)
            # are those of dict(zip(...)) -- no per-row Python function
            idents = [q[genecol] for q in meta_rows] if roary else \
                ["_|_".join((q[genecol], q[nugcol], q[anncol])) for q in meta_rows]
            last = dict(zip(idents, range(len(idents))))
            ids, source = list(last.keys()), list(last.values())
            nugn = [meta_rows[r][nugcol] for r in source]
            ann = [meta_rows[r][anncol] for r in source]
            for c in grabcols:
                extra[header[c] + "_name"] = [meta_rows[r][c] for r in source]
        else:
            for r, q in enumerate(meta_rows):
                take(q, r, None)
        rows64 = bits[np.array(source, dtype=np.int64)] if ids else bits[:0]
        table = GeneTable(ids, nugn, ann, kept_strains, rows64, extra)
        file_rows64 = np.ascontiguousarray(bits, dtype=np.uint64)     # EVERY file row, see below
    else:
        for r, q in enumerate(rows_iter):
            try:
                present = [q[startcol + c] not in ABSENT_CELLS for c in keep]
            except IndexError:
                sys.exit("CRITICAL: Could not read gene presence absence file. Verify that "
                         "this file is a proper Roary file using the specified delimiter "
                         "(default is ',').")
            take(q, r, present)
            all_rows.append(present)
        dense = np.array(dense_rows, dtype=np.uint8).reshape(len(ids), len(keep))
        # A -s inside the identifier cells makes "Non-unique Gene name" / "Annotation" isolate columns, and in the
        # reference's dictionary the isolate's 0 / 1 then REPLACES the text stored under that key (:449-481): the
        # result files print it.  Same here (the last column of that name wins, as in a dict).
        for key, texts in (("Non-unique Gene name", nugn), ("Annotation", ann)):
            if key in kept_strains:
                c = len(kept_strains) - 1 - kept_strains[::-1].index(key)
                texts[:] = [str(int(v)) for v in dense[:, c]]
        table = GeneTable(ids, nugn, ann, kept_strains, pack_bits_rows(dense), extra)
        file_rows64 = pack_bits_rows(np.array(all_rows, dtype=np.uint8).reshape(len(all_rows), len(keep)))
    if opened is not None:
        opened.close()
    # Zero_ones_matrix comes from every row of the FILE, not from the de-duplicated table: the
    # reference appends each (variable) row as it reads it and only the dict entry of a
    # repeated identifier is overwritten (scoary/methods.py:445-497), so repeated identifiers
    # (VCF multi-allelic sites in non-Roary files) count once per row in the Hamming distances.
    return {"Roarydic": table, "Zero_ones_matrix": _LazyZeroOnes(file_rows64, len(kept_strains)),
            "Strains": kept_strains, "Extracols": extracols, "Firstcolnames": firstcolnames}


class _LazyZeroOnes:
    """``Zero_ones_matrix`` (methods.py:493-502): strain-major 0/1 lists of the
    variable genes.  Only the tree builder wants it, so the list-of-lists is
    built on first use."""

    def __init__(self, rows64, nstrains):
        self._rows64, self._n, self._lists = rows64, nstrains, None

    def file_rows(self):
        """(file rows, N) uint8 presence of every row of the gene file (what the
        reference's zero_ones_line loop sees, before the all-0 / all-1 filter)."""
        rows, words = self._rows64.shape     # explicit width: a table without data rows has size 0
        by = np.ascontiguousarray(self._rows64).view(np.uint8).reshape(rows, words * 8)
        return np.unpackbits(by, axis=1, bitorder="little")[:, :self._n]

    def _get(self):
        if self._lists is None:
            d = self.file_rows()
            tot = d.sum(axis=1) if d.size else np.zeros(0)
            var = (tot > 0) & (tot < d.shape[1])
            # no variable row: the reference transposes an empty list, zip(*[]) == [] (:499)
            self._lists = d[var].T.tolist() if d.shape[1] and var.any() else []
        return self._lists

    def __iter__(self):
        return iter(self._get())

    def __len__(self):
        return len(self._get())

    def __getitem__(self, k):
        return self._get()[k]


def _read_gpa_ranks(io_native, path, delimiter, startcol, allowed_isolates, wanted):
    """The native reader, shared between the ranks of a torchrun launch: every rank parses
    one byte range of the body (cut at line ends) and the bit rows / text columns are put
    back together with one all-gather each, instead of every rank tokenising the whole file
    (the reference's workers all receive the one parsed dict, scoary/methods.py:1083-1097;
    here the file is the thing that is big).  A boundary inside a quoted multi-line cell, or
    SCOARY_SHARDED_READ=0, makes every rank read the whole file."""
    from . import dist as _dist
    world, rank = _dist.world_rank()
    if world > 1 and os.environ.get("SCOARY_SHARDED_READ", "1") != "0":
        part, status = None, "ok"
        try:
            part = io_native.read_gpa(path, delimiter, startcol, allowed_isolates,
                                      need_cols=wanted, part=(rank, world))
        except io_native.GpaPartBoundary:
            status = "boundary"
        except io_native.GpaError as e:
            status = "error:" + str(e)
        except Exception as e:   # OSError, MemoryError, ctypes failure: still reach the gather,
            status = "fatal:%s: %s" % (type(e).__name__, e)     # or the other ranks block in it
        statuses = _dist.all_gather_objects(status)
        fatal = [x[6:] for x in statuses if x.startswith("fatal:")]
        if fatal:
            raise RuntimeError("reading %s failed on a rank: %s" % (path, fatal[0]))
        if "boundary" not in statuses:
            # (after a boundary inside a quoted cell the NEXT rank starts mid-cell and sees a
            # malformed row: its error means nothing, the whole-file read below decides)
            errs = [x[6:] for x in statuses if x.startswith("error:")]
            if errs:
                raise io_native.GpaError(errs[0])    # a malformed row: every rank leaves together
            header, meta_rows, bits, kept = part
            bits, counts = _dist.all_gather_host_rows(bits)
            meta_rows = [row for rows in _dist.all_gather_objects(meta_rows) for row in rows]
            log.info("Read the gene presence/absence file in %d byte ranges (rows per rank: %s)"
                     % (world, ", ".join(str(c) for c in counts)))
            return header, meta_rows, bits, kept
    return io_native.read_gpa(path, delimiter, startcol, allowed_isolates, need_cols=wanted)


def _reduced_name(outdir, time):
    return "%sgene_presence_absence_reduced%s.csv" % (outdir, time)


def ReduceSet(genefile, delimiter, grabcols, startcol=14, allowed_isolates=None, time="",
              outdir="./"):
    """-r/-w: write the column subset of the table (methods.py:510-544)."""
    reader = csv.reader(genefile, skipinitialspace=True, delimiter=delimiter)
    header = next(reader)
    cols = list(range(startcol)) + [c for c in range(len(header)) if header[c] in allowed_isolates]
    log.info("Writing gene presence absence file for the reduced set of isolates")
    name = _reduced_name(outdir, time)
    with open(name, "w") as out:
        w = csv.writer(out, delimiter=delimiter)
        w.writerow([header[c] for c in cols])
        for r in reader:
            w.writerow([r[c] for c in cols])
    log.info("Finished writing reduced gene presence absence list to file %s" % name)
    return name


def Csv_to_dic(csvfile, delimiter, allowed_isolates, strains):
    """Read the traits table (methods.py:546-614): returns (traitsdic, Prunedic);
    isolates with a missing value are dropped from that trait only."""
    cols = list(zip(*csv.reader(csvfile, delimiter=delimiter)))
    if len(cols) < 2:
        sys.exit("Please check that your traits file is formatted properly and contains at "
                 "least one trait")
    traitsdic, prunedic = {}, {}
    for tcol in cols[1:]:
        p = dict(zip(cols[0], tcol))
        if "" in p:
            name = p.pop("")
        elif "Name" in p:
            name = p.pop("Name")
        else:
            sys.exit("Make sure the top-left cell in the traits file is either empty or "
                     "'Name'. Do not include empty rows")
        if allowed_isolates is not None:
            p = {s: v for s, v in p.items() if s in allowed_isolates}
        if not all(v in ALLOWED_TRAIT for v in p.values()):
            sys.exit("Unrecognized character found in trait file. Allowed values (no "
                     "commas): %s" % ",".join(["0", "1", "NA", ".", "-", " ", ""]))
        missing = [s for s, v in p.items() if v in MISSING_TRAIT]
        if missing:
            log.warning("WARNING: Some isolates have missing values for trait %s. "
                        "Missing-value isolates will not be counted in association analysis "
                        "towards this trait." % str(name))
        kept = {s: v for s, v in p.items() if v not in MISSING_TRAIT} if missing else p
        prune = list(missing)
        if not all(s in p for s in strains):
            log.error("ERROR: Some isolates in your gene presence absence file were not "
                      "represented in your traits file. These will count as MISSING data and "
                      "will not be included.")
            prune += [s for s in strains if s not in p and s not in prune]
        traitsdic[name] = kept
        prunedic[name] = prune + [None]
    return traitsdic, prunedic


# ---------------------------------------------------------------------------
# Statistics
# ---------------------------------------------------------------------------
def _as_table(genedic):
    return genedic if isinstance(genedic, GeneTable) else GeneTable.from_genedic(genedic)


def Perform_statistics(traits, genes):
    """One (gene, trait) table (methods.py:930-982): ``traits`` {strain: "0"/"1"},
    ``genes`` {strain: 0/1, ...}.  Counted on the GPU through the same kernel as
    the batched path."""
    strains = list(traits.keys())
    for s in strains:
        if s not in genes:
            log.critical("CRITICAL: Could not find %s in the genes file." % str(s))
            sys.exit("Make sure strains are named the same in your traits file as in your "
                     "gene presence/absence file")
    g = np.array([[1 if genes[s] else 0 for s in strains]], dtype=np.uint8)
    t = np.array([[2 if traits[s] in ("NA", "-", ".") else int(traits[s]) for s in strains]],
                 dtype=np.uint8)
    eng = get_engine()
    N = len(strains)
    counts, _ = eng.counts(eng.pack_dense(g), eng.vecrows(pack_bits_rows(t == 1), N),
                           eng.vecrows(pack_bits_rows(t != 2), N))
    tpgp, tpgn, tngp, tngn = (int(x) for x in counts.cpu().numpy()[0, 0])
    gene_trait = {}
    for j, s in enumerate(strains):
        gene_trait[s] = "NA" if t[0, j] == 2 else \
            ("A" if g[0, j] else "a") + ("B" if t[0, j] == 1 else "b")
    pattern = "".join(str(int(x)) for x in g[0])
    return {"statistics": {"tpgp": tpgp, "tpgn": tpgn, "tngp": tngp, "tngn": tngn},
            "hash": int(pattern, 2) if pattern else 0, "gene_trait": gene_trait}


def bonferroni_bh(p_sorted_input, number_of_tests, order=None):
    """Bonferroni and step-up Benjamini-Hochberg exactly as methods.py:903-925.

    ``p_sorted_input``: p-values in insertion order.  The reference sorts
    stably, keeps the least significant p as is, and walks towards the most
    significant with  bh = last if tie else min(last, p*ntests/rank), where
    tie = exact equality with the next sorted p.  Vectorised without changing
    a single floating-point operation: inside a run of equal p only the run's
    last (highest-rank) element evaluates min(); every other member copies it.
    So: reversed cumulative minimum over the run ends, broadcast to the runs.
    ``order``: the stable ascending argsort of the p-values if the caller already has it."""
    p = np.asarray(p_sorted_input, dtype=np.float64)
    n = p.shape[0]
    if n == 0:
        # the reference raises IndexError at methods.py:914 here
        raise IndexError("no testable genes for this trait")
    order = np.argsort(p, kind="stable") if order is None else order
    sp = p[order]
    v = sp * number_of_tests / (np.arange(n, dtype=np.float64) + 1.0)
    v[n - 1] = sp[n - 1]
    is_end = np.ones(n, dtype=bool)
    is_end[:-1] = sp[:-1] != sp[1:]
    ends = np.nonzero(is_end)[0]
    bh_ends = np.minimum.accumulate(v[ends][::-1])[::-1]
    run = np.cumsum(is_end) - is_end
    bh = np.empty(n, dtype=np.float64)
    bh[order] = bh_ends[run]
    return np.minimum(p * number_of_tests, 1.0), np.minimum(bh, 1.0)


def _pattern_groups(table, maskrow, idx, hashes):
    """Group id per tested gene (``idx``): equal ids <=> identical presence
    pattern over the trait's valid isolates.  ``hashes``: (G, 2) uint64 from the
    device (scoary_row_hash); candidate groups are confirmed on the bit rows,
    so a hash collision only costs a second pass."""
    keyed = None
    if hashes is not None:
        h = hashes[idx]
        _, inv = np.unique(h, axis=0, return_inverse=True)
        inv = inv.reshape(-1)
        # confirm: within a hash group every row must equal the group's first row
        order = np.argsort(inv, kind="stable")
        gs = inv[order]
        first = np.nonzero(np.r_[True, gs[1:] != gs[:-1]])[0]
        rep = order[np.repeat(first, np.diff(np.r_[first, len(gs)]))]
        keyed = table.rows64[idx] & maskrow[None, :]
        if np.array_equal(keyed[order], keyed[rep]):
            return inv
    if keyed is None:
        keyed = table.rows64[idx] & maskrow[None, :]
    _, inv = np.unique(keyed, axis=0, return_inverse=True)
    return inv.reshape(-1)


def _associate(table, tarr, permutations=0, seed=DEFAULT_SEED, early_abort=False):
    """Whole hot path for all traits; under torchrun (world > 1) every rank
    takes a stride gene shard (dist.GenePartition: the reference's domains,
    scoary/methods.py:1076-1078) and the per-gene records are all-gathered
    over RCCL and woven back into file order, so every rank returns the full arrays.
    ``early_abort``: the reference's sequential estimator (scoary/methods.py:1360-1363)
    on the Fisher statistic instead of the fixed-P count: out["nstop"] then holds the
    permutation count every gene stopped at (0 = ran to the end)."""
    import torch
    from . import dist
    eng = get_engine()
    G, N, T = len(table), len(table.strains), tarr.shape[0]
    with _stage("device setup (H2D, tiling, trait plan)"):
        trv = eng.vecrows(pack_bits_rows(tarr == 1), N)
        mkv = eng.vecrows(pack_bits_rows(tarr != 2), N)
        plan = eng.trait_plan(trv, mkv, N)            # margins + mask classes: once per trait set

    def scipy_digits(res):
        # Up to 170 isolates k_fisher's p IS scipy.stats.fisher_exact's double; above, it is the exact value of SciPy's
        # rule to ~3e-15 -- inside the path's 1e-12, not the digits the reference writes.  What gets PRINTED is
        # therefore passed through scoary_fisher_scipy (Boost's prime-factorised pmf restated, bit for bit SciPy up
        # to 104 723 isolates): the result files are the reference's bytes at any realistic size.  An
        # output-fidelity pass, ~50 x k_fisher per table (9.5 ms per 500 000 tables at N = 2000) and not part of the
        # step bench.py times; SCOARY_FISHER_SCIPY=0 keeps k_fisher's value.
        if N <= 170 or os.environ.get("SCOARY_FISHER_SCIPY", "1") == "0":
            return
        with _stage("SciPy's digits for the printed p (k_fisher_scipy)"):
            skipped = eng.fisher_scipy(res["counts"], res["p"])
            torch.cuda.synchronize(eng.device)
        if skipped:
            log.info("%d tables are outside what scoary_fisher_scipy takes (more than %d isolates): their p is the exact "
                     "value of SciPy's rule (within 1e-12 of SciPy), not SciPy's last digits"
                     % (skipped, eng.fisher_scipy_max_isolates()))

    def local(sel):
        a = sel.start
        whole = sel == slice(0, G, 1)
        if len(range(*sel.indices(G))) == 0:
            return torch.zeros((T, 0, dist.REC_WORDS), dtype=torch.int32, device=eng.device)
        with _stage("device setup (H2D, tiling, trait plan)"):
            gm = table.on_device(eng) if whole else eng.tile_rows(table.rows64[sel], N)
            torch.cuda.synchronize(eng.device)
        if permutations > 0 and not early_abort and gm.lists is None and eng.lists_supported(N):
            # list-driven permutation kernel: cost follows each gene's minority count.  The
            # index array (4 bytes per padded minority entry, 2 for N > 20479) is sized by the
            # plan before it is allocated: a wide, dense matrix that would not fit keeps the
            # dense kernels (same results, no lists) instead of dying in the allocator
            from .engine import ListMemoryError
            try:
                with _stage("index lists (device)"):
                    eng.build_lists(gm)
                    torch.cuda.synchronize(eng.device)
            except (ListMemoryError, torch.cuda.OutOfMemoryError) as e:
                gm.lists = None
                log.info("index lists not built (%s); using the dense permutation kernels" % e)
        elif permutations > 0 and not early_abort and not eng.lists_supported(N) and a == 0:
            # more than 131 070 isolates: the list counts would need a 17th counter plane; the
            # dense AND + popcount kernels take over (same results, 1.5-3x the time per test)
            log.info("%d isolates: more than the list-driven permutation kernel takes (%d); "
                     "using the dense permutation kernels" % (N, eng.lib.scoary_list_max_isolates()))
        if early_abort and permutations > 0:
            from . import tree as T_
            res = eng.associate(gm, trv, mkv, permutations=0, plan=plan)
            crit = eng.fisher(res["counts"], want_crit=True)[2]
            scipy_digits(res)
            r, nstop = eng.permute_sequential(gm, mkv, res["margins"], crit, permutations, seed,
                                              T_._abort_thresholds(permutations))
            res["r"] = r
            return eng.pack_records(res, nstop=nstop)
        with _stage("kernels (counts, Fisher, permutations)"):
            res = eng.associate(gm, trv, mkv, permutations=permutations, seed=seed, plan=plan)
            torch.cuda.synchronize(eng.device)
        scipy_digits(res)
        with _stage("kernels (counts, Fisher, permutations)"):
            rec = eng.pack_records(res)
            torch.cuda.synchronize(eng.device)
        return rec

    rec = dist.associate_sharded(local, G)
    with _stage("results D2H"):
        # the stable p order for BH: numpy does 200 000 doubles in 25-40 ms, which beats the
        # first-use cost of torch's sort kernels in a fresh process (0.2-0.4 s of lazy code loading,
        # profiles/r05_e2e_cli_*); the traits' sorts also run on worker threads (Setup_results), so
        # the device sort only wins from tens of millions of (gene, trait) pairs on
        p_order = _p_order_on_device(rec) if T * G >= DEVICE_SORT_MIN_PAIRS else None
        out = dist.numpy_records(rec)
        out["p_order"] = p_order
    if permutations <= 0:
        out["r"] = None
    return out


DEVICE_SORT_MIN_PAIRS = 32_000_000
HOST_THREADS_MIN_PAIRS = 200_000        # (gene, trait) pairs from which the per-trait statistics use worker threads


def _p_order_on_device(rec):
    """Per trait: the testable genes (skip rule, methods.py:804-814) in ascending order of their
    Fisher p, ties in file order -- the stable sort Benjamini-Hochberg (methods.py:903-925) and
    the results writer (:1124) both start from.  The records are still on the device here, so the
    one O(G log G) step of the host statistics is one batched torch.argsort there (a stable sort
    of 200 000 doubles is 25-40 ms per trait in numpy, 50 traits x 1 M genes would be a minute).
    Returns a list of int64 arrays of positions into the trait's testable genes."""
    torch = _torch_mod()
    T, G, _ = rec.shape
    if G == 0:
        return [np.zeros(0, dtype=np.int64) for _ in range(T)]
    r = rec.contiguous()
    c = r[:, :, 0:4]
    testable = ((c[:, :, 0] + c[:, :, 2]) != 0) & ((c[:, :, 1] + c[:, :, 3]) != 0)
    pv = r[:, :, 4:6].reshape(-1).clone().view(torch.float64).view(T, G)
    key = torch.where(testable, pv, torch.full_like(pv, float("inf")))
    order = torch.argsort(key, dim=1, stable=True).to(torch.int32).cpu().numpy()
    tst = testable.cpu().numpy()
    out = []
    for t in range(T):
        n = int(tst[t].sum())
        pos = np.cumsum(tst[t]) - 1                       # gene index -> position among the testable
        out.append(pos[order[t, :n]].astype(np.int64))
    return out


def _torch_mod():
    import torch
    return torch


def _usable_cpus():
    """Logical CPUs this process may use: the affinity mask and the cgroup quota applied to
    os.cpu_count() (a container is often granted far fewer than the host has)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def Setup_results(genedic, traitsdic, collapse, permutations=0, seed=DEFAULT_SEED,
                  early_abort=False):
    """Counts, Fisher's exact test and B/BH correction for every trait x gene
    (methods.py:757-928).  ``permutations`` >= 10 additionally attaches the
    Fisher-statistic ``Empirical_p`` (= (r+1)/(P+1), methods.py:1365) to every
    row -- the north_star's replacement for the tree-statistic Permute loop."""
    table = _as_table(genedic)
    names, tarr = _trait_arrays(traitsdic, table.strains)
    dev = _associate(table, tarr, permutations if permutations >= 10 else 0, seed, early_abort)
    collapse_hashes = None
    if collapse:
        eng = get_engine()
        N = len(table.strains)
        collapse_hashes = eng.row_hash(table.on_device(eng),
                                       eng.vecrows(pack_bits_rows(tarr != 2), N))
    all_traits, combos = {}, {}
    G = len(table)
    t_host = _time.perf_counter()

    def one_trait(t):
        trait = names[t]
        c = dev["counts"][t]                                   # tpgp, tpgn, tngp, tngn
        testable = ((c[:, 0] + c[:, 2]) != 0) & ((c[:, 1] + c[:, 3]) != 0)
        number_of_tests = G - int((~testable).sum())
        idx = np.nonzero(testable)[0]
        if len(idx) == 0:
            # (the reference falls over here too: IndexError from its empty sort list, scoary/methods.py:903-925)
            raise IndexError("Trait %s has no testable genes" % trait)
        p_all, or_all = dev["p"][t], dev["odds"][t]
        emp = None
        if dev["r"] is not None:
            # (r+1)/(P+1), :1365; with --permute-early-abort a gene that stopped after n
            # permutations gets (r+1)/(n+1) = the reference's (r+1.0)/(i+2.0), :1362
            n_used = np.where(dev["nstop"][t] > 0, dev["nstop"][t], permutations).astype(np.float64)
            emp = (dev["r"][t].astype(np.float64) + 1.0) / (n_used + 1.0)

        if not collapse:
            # names stay in the GeneTable: the result rows are (table, idx) -- nothing per
            # (gene, trait) is materialised in Python (a cfg5-sized run has 50 M such pairs)
            rows_idx, members, names_out = idx, None, None
            nugn = ann = None
            plist = p_all[idx]
            bh_rank_p = plist
        else:
            # identical presence pattern over the trait's valid isolates => one
            # merged unit (methods.py:816-840).  In the reference's dict
            # bookkeeping a unit is re-inserted (at the end) whenever it absorbs a
            # gene and takes that gene's statistics, so: unit order = ascending
            # index of its LAST member; name / NUGN / annotation = members joined
            # by "--" in file order; statistics of the last member.
            maskrow = pack_bits_rows((tarr[t:t + 1] != 2))[0]
            group = _pattern_groups(table, maskrow, idx, collapse_hashes[t] if
                                    collapse_hashes is not None else None)
            order_g = np.argsort(group, kind="stable")           # members stay in file order
            gs = group[order_g]
            starts = np.nonzero(np.r_[True, gs[1:] != gs[:-1]])[0]
            ends = np.r_[starts[1:], len(gs)]
            last_pos = order_g[ends - 1]                         # position (in idx) of last member
            unit_order = np.argsort(last_pos, kind="stable")
            members_idx = [idx[order_g[starts[u]:ends[u]]].tolist() for u in unit_order]
            number_of_tests -= int(len(idx) - len(starts))
            rows_idx = np.array([m[-1] for m in members_idx], dtype=np.int64)
            members = [[table.ids[i] for i in m] for m in members_idx]
            names_out = ["--".join(m) for m in members]
            nugn = ["--".join(table.nugn[i] for i in m) for m in members_idx]
            ann = ["--".join(table.annotation[i] for i in m) for m in members_idx]
            plist = p_all[rows_idx]
            # the reference's p_value_list keeps one entry per tested gene,
            # superseded names included (methods.py:873/892), so BH ranks count
            # them; a unit's BH is that of its last member's entry
            bh_rank_p = p_all[idx]
            bh_entry = last_pos[unit_order]

        cc = c[rows_idx]
        num_pos = (cc[:, 0] + cc[:, 1]).astype(np.float64)
        num_neg = (cc[:, 2] + cc[:, 3]).astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            sens = np.where(num_pos > 0, cc[:, 0].astype(np.float64) / num_pos * 100, 0.0)
            spes = np.where(num_neg > 0, cc[:, 3].astype(np.float64) / num_neg * 100, 0.0)
        # stable ascending order of the testable genes' p: from the device if it came along
        p_order = dev["p_order"][t] if dev.get("p_order") is not None else None
        if not collapse:
            if p_order is None:
                p_order = np.argsort(plist, kind="stable")
            B, BH = bonferroni_bh(plist, number_of_tests, order=p_order)   # order shared with the writer
        else:
            _, bh_all = bonferroni_bh(bh_rank_p, number_of_tests, order=p_order)
            p_order = None                                   # the units' own p order is the writer's business
            BH = bh_all[bh_entry]
            B = np.minimum(plist * number_of_tests, 1.0)
        cols = {"tpgp": cc[:, 0], "tngp": cc[:, 2], "tpgn": cc[:, 1], "tngn": cc[:, 3],
                "sens": sens, "spes": spes, "OR": or_all[rows_idx], "p_v": plist,
                "B_p": B, "BH_p": BH}
        if emp is not None:
            cols["Empirical_p"] = emp[rows_idx]
        if collapse:
            tr = TraitResults(names_out, nugn, ann, cols, number_of_tests, members)
        else:
            tr = TraitResults(None, None, None, cols, number_of_tests, None, table=table, rows_idx=rows_idx)
            tr.p_order = p_order
        return tr, GeneTraitCombinations(table, tr, members, tarr[t], t)

    for trait in names:
        log.info("Gene-wise counting and Fisher's exact tests for trait: %s" % str(trait))
    # the traits are independent and their statistics are numpy sorts / gathers that release the
    # interpreter lock: one worker thread per trait, as many as the process may use (50 traits x
    # 125 000 genes: 1.5 s one after the other)
    nthreads = min(len(names), _usable_cpus())
    if nthreads > 1 and G * len(names) >= HOST_THREADS_MIN_PAIRS:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=nthreads) as pool:
            done = list(pool.map(one_trait, range(len(names))))      # in trait order; the first error is re-raised
    else:
        done = [one_trait(t) for t in range(len(names))]
    for trait, (tr, gtc) in zip(names, done):
        all_traits[trait], combos[trait] = tr, gtc
    _stage.seconds["host statistics (skip rule, B / BH, columns)"] = \
        _stage.seconds.get("host statistics (skip rule, B / BH, columns)", 0.0) + _time.perf_counter() - t_host
    return {"Results": all_traits, "Gene_trait_combinations": combos}


def Permute(tree, GTC, permutations, cutoffs, seed=DEFAULT_SEED, trait_index=0):
    """Empirical p of ONE gene by label permutation (methods.py:1314-1369).

    With a ``tree`` (nested lists of the GTC's strains): the reference's
    statistic -- supporting (or opposing) pairs / max contrasting pairs of the
    PhyloTree, r = #permutations reaching the observed value, sequential early
    abort after 30 permutations, (r+1)/(P+1) otherwise.  With ``tree=None``: the
    Fisher statistic the north_star prescribes for the ``--no_pairwise`` path,
    (r+1)/(P+1) with r = #label shuffles whose 2x2 table is as or less probable
    than the observed one.  ``GTC`` is {strain: "AB"|"Ab"|"aB"|"ab"}."""
    if permutations < 10:
        sys.stdout.write("Number of permutations too few. The absolute minimum is 10.")
        return None
    strains = list(GTC.keys())
    g = np.array([[1 if GTC[s][0] == "A" else 0 for s in strains]], dtype=np.uint8)
    t = np.array([[1 if GTC[s][-1] == "B" else 0 for s in strains]], dtype=np.uint8)
    if tree is None:
        table = GeneTable(["gene"], [""], [""], strains, pack_bits_rows(g))
        dev = _associate(table, t, permutations, seed)
        return (float(dev["r"][0, 0]) + 1.0) / (permutations + 1.0)
    stage = _TreeStage(get_engine(), tree, strains, t[0], trait_index, seed)
    obs = stage.observed(pack_bits_rows(g))
    exceed = stage.permute(pack_bits_rows(g), obs, permutations)
    from .tree import empirical_p_sequential
    return empirical_p_sequential(exceed[0])


class _TreeStage:
    """Device state of the pairwise-comparison stage for ONE trait: the pruned
    tree as a stack program, label bits in tip order, and the kernels that
    evaluate genes (observed) and gene x permutation (scoary_tree_permute)."""

    def __init__(self, engine, tree, strains, trait_row, trait_index, seed):
        import torch
        from .tree import TreeProgram
        self.eng, self.N = engine, len(strains)
        self.trait_index, self.seed = int(trait_index), seed
        index_of = {s: i for i, s in enumerate(strains)}
        self.prog = TreeProgram(tree, index_of)
        dev = engine.device
        self.ops = torch.from_numpy(self.prog.ops).to(dev)
        self.tips = torch.from_numpy(self.prog.tips).to(dev)
        trait_row = np.asarray(trait_row, dtype=np.uint8)
        self.label_rows = engine.vecrows(pack_bits_rows((trait_row == 1)[None]), self.N)
        self.mask_rows = engine.vecrows(pack_bits_rows((trait_row != 2)[None]), self.N)
        self.labels_tip = engine.gather_bits(self.label_rows, self.tips)
        self.margins = torch.tensor([[int((trait_row == 1).sum()), int((trait_row != 2).sum())]],
                                    dtype=torch.int32, device=dev)

    def _gene_tip_bits(self, rows64):
        return self.eng.gather_bits(self.eng.vecrows(rows64, self.N), self.tips)

    def observed(self, rows64):
        """(G, 3) int32 numpy: max contrasting / supporting / opposing pairs."""
        gt = self._gene_tip_bits(rows64)
        out = self.eng.tree_pairs(self.ops, self.prog.depth, gt, self.labels_tip, self.prog.ntips)
        return out[:, 0, :].cpu().numpy()

    def permute(self, rows64, obs, permutations, batch_threads=1 << 26):
        """(G, P) uint8 numpy exceedance flags (methods.py:1353-1355) under the
        spec-S4 label permutations 0..P-1 of this trait."""
        import torch
        gt = self._gene_tip_bits(rows64)
        obs_d = torch.from_numpy(np.ascontiguousarray(obs, dtype=np.int32)).to(self.eng.device)
        G = gt.shape[0]
        out = np.empty((G, permutations), dtype=np.uint8)
        pb = max(1, min(permutations, batch_threads // max(G, 1)))
        for p0 in range(0, permutations, pb):
            nb = min(pb, permutations - p0)
            perms = self.eng.perm_generate(self.mask_rows, self.margins, self.N, nb, p0, self.seed,
                                           trait_base=self.trait_index)
            ptip = self.eng.gather_bits(perms[0], self.tips)
            ex = self.eng.tree_permute(self.ops, self.prog.depth, gt, ptip, self.prog.ntips, obs_d)
            out[:, p0:p0 + nb] = ex.cpu().numpy()
        return out


# ---------------------------------------------------------------------------
# Output
# ---------------------------------------------------------------------------
def _fmt(x):
    """str() of the reference's cells: ints as decimals, floats as Python's
    shortest round-trip repr."""
    if isinstance(x, (int, np.integer)):
        return str(int(x))
    return repr(float(x))


def SortResultsAndSetKey(genedic, key="p_v"):
    """{rank: gene} ascending by ``key``, stable (methods.py:1448-1454)."""
    if isinstance(genedic, TraitResults):
        order = np.argsort(np.asarray(genedic.column(key), dtype=np.float64), kind="stable")
        return {i: genedic.genes[j] for i, j in enumerate(order)}
    return {i: g for i, g in enumerate(sorted(genedic, key=lambda x: genedic[x][key]))}


CUT_FIELD = {"I": "p_v", "B": "B_p", "BH": "BH_p", "PW": "Plowest", "EPW": "Pboth",
             "P": "Empirical_p"}


def StoreResults(Results, max_hits, cutoffs, upgmatree, GTC, Prunedic, outdir, permutations,
                 num_threads, no_pairwise, genedic, extracolstoprint, firstcolnames, time="",
                 delimiter=",", seed=DEFAULT_SEED):
    def one(Trait, writer_threads=0):
        StoreTraitResult(Results[Trait], Trait, max_hits, cutoffs, upgmatree, GTC, Prunedic,
                         outdir, permutations, num_threads, no_pairwise, genedic,
                         extracolstoprint, firstcolnames, time, delimiter, seed=seed,
                         writer_threads=writer_threads)
    traits = list(Results)
    pairs = sum(len(Results[t]) for t in traits)
    nthreads = min(len(traits), _usable_cpus())
    if no_pairwise and nthreads > 1 and pairs >= HOST_THREADS_MIN_PAIRS:
        # --no_pairwise: a trait's file is sorting, filtering and formatting its own columns (numpy and
        # the native writer, both outside the interpreter lock) -- one worker thread per trait.  The
        # pairwise stage keeps the loop: it drives the GPU and logs its progress trait by trait.
        for Trait in traits:
            sys.stdout.write("\n")
            log.info("Storing results: " + Trait)
        from concurrent.futures import ThreadPoolExecutor
        with _stage("result files (sort, filter, pairwise stage, write)"):
            for Trait in traits:                     # the gene table's string columns: once, before the workers
                tr = Results[Trait]
                if isinstance(tr, TraitResults) and tr._table is not None:
                    _csv_text_columns(tr._table)
            with ThreadPoolExecutor(max_workers=nthreads) as pool:
                list(pool.map(lambda t: one(t, writer_threads=1), traits))   # the files are the parallel axis
        return
    for Trait in traits:
        sys.stdout.write("\n")
        log.info("Storing results: " + Trait)
        with _stage("result files (sort, filter, pairwise stage, write)"):
            one(Trait)


def decideifbreak(cutoffs, currentgene):
    """True when a gene violates a naive / Bonferroni / BH cutoff, which ends
    the pairwise stage for everything less significant (methods.py:1489-1508)."""
    for m in ("I", "B", "BH"):
        if m in cutoffs and currentgene[CUT_FIELD[m]] > cutoffs[m]:
            return True
    return False


def _kept_ranks(order, cols, cutoffs, num_threads=None):
    """Ranks (positions in ``order``) of the genes the reference's pairwise stage keeps.
    Worker k of n walks the ranks k, k+n, k+2n, ... and stops at ITS OWN first gene that fails
    an I/B/BH cutoff (scoary/methods.py:1076-1078, :1290-1294): the kept set is the union of the
    workers' prefixes.  With one worker, or with cutoff columns that are monotone in the rank,
    that is a prefix of ``order``; a non-monotone column (the stale-rank BH of --collapse) lets
    a later worker keep ranks past another worker's stop.  The weave's modulo (StoreTraitResult)
    is taken of these ranks."""
    nworkers = max(1, int(num_threads or 1))
    ranks = []
    for k in range(nworkers):
        for rank in range(k, len(order), nworkers):
            i = order[rank]
            if any(m in cutoffs and cols[CUT_FIELD[m]][i] > cutoffs[m] for m in ("I", "B", "BH")):
                break
            ranks.append(rank)
    return np.array(sorted(ranks), dtype=np.int64)


def _pairwise_stage(Trait, Traitname, order, cutoffs, upgmatree, GTC, Prunedic, permutations,
                    genedic, seed, num_threads=None):
    """PairWiseComparisons (methods.py:1208-1312) for the genes ``order`` (row
    indices, ascending naive p) of one trait, on the GPU: the prefix that passes
    the I/B/BH cutoffs gets max contrasting / supporting / opposing pairs, the
    two binomial p-values and, with permutations, the tree-statistic empirical p.
    Returns (row indices kept, their ranks in ``order``, dict of extra columns over those
    rows)."""
    from . import tree as T
    cols = {k: np.asarray(Trait.column(k)) for k in ("p_v", "B_p", "BH_p")}
    ranks = _kept_ranks(order, cols, cutoffs, num_threads)
    keep = np.asarray(order, dtype=np.int64)[ranks] if len(ranks) else np.zeros(0, dtype=np.int64)
    extra = {k: np.zeros(len(keep)) for k in ("Pbest", "Pworst", "Plowest", "Pboth")}
    for k in ("max_total_pairs", "max_propairs", "max_antipairs"):
        extra[k] = np.zeros(len(keep), dtype=np.int64)
    if permutations >= 10:
        extra["Empirical_p"] = np.zeros(len(keep))
    if len(keep) == 0:
        return keep, ranks, extra
    gtc = GTC[Traitname]
    table = gtc.table
    tree = upgmatree
    if len(Prunedic[Traitname]) > 0:
        tree = T.prune_missing(upgmatree, Prunedic[Traitname])
    stage = _TreeStage(get_engine(), tree, table.strains, gtc.trait_row, gtc.trait_index, seed)
    src = [Trait.source_index(i) if Trait.source_index(i) is not None else
           table.index(Trait.members[i][-1] if Trait.members is not None else Trait.gene_at(i))
           for i in keep]
    rows64 = table.rows64[src]
    obs = stage.observed(rows64)
    extra["max_total_pairs"] = obs[:, 0].astype(np.int64)
    extra["max_propairs"] = obs[:, 1].astype(np.int64)
    extra["max_antipairs"] = obs[:, 2].astype(np.int64)
    # the two binomial p-values of every gene (methods.py:1259-1275), SciPy's doubles (tree.binom_two_sided_many)
    tot, pro, anti = (obs[:, j].astype(np.int64) for j in range(3))
    both = T.binom_two_sided_many(np.concatenate([pro, tot - anti]), np.concatenate([tot, tot]))
    a, b = both[:len(keep)], both[len(keep):]
    best_is_a = pro >= anti
    extra["Pbest"], extra["Pworst"] = np.where(best_is_a, a, b), np.where(best_is_a, b, a)
    extra["Plowest"] = np.minimum(extra["Pbest"], extra["Pworst"])
    extra["Pboth"] = np.maximum(extra["Pbest"], extra["Pworst"])
    if permutations >= 10:
        log.info("Performing %d label permutations of the pairwise-comparison statistic for "
                 "%d genes on the GPU" % (permutations, len(keep)))
        exceed = stage.permute(rows64, obs, permutations)
        extra["Empirical_p"] = T.empirical_p_sequential_many(exceed)
    return keep, ranks, extra


def _csv_text_columns(table):
    """The three leading text cells of every gene of a GeneTable as string tables for the native
    writer, built once per table: Roary identifiers give (gene, non-unique name, annotation),
    identifiers of the form c0_|_c1_|_c2 (non-Roary files, scoary/methods.py:458-460) their three
    parts (methods.py:1154-1158).  None if some identifier splits into another number of cells
    (the general writer handles those)."""
    cached = getattr(table, "_csv_text", None)
    if cached is not None:
        return cached or None
    from . import io_native
    c0, c1, c2 = [], [], []
    for g, n, a in zip(table.ids, table.nugn, table.annotation):
        if "_|_" in g:
            parts = g.split("_|_")
            if len(parts) != 3:
                table._csv_text = False
                return None
            c0.append(parts[0]); c1.append(parts[1]); c2.append(parts[2])
        else:
            c0.append(g); c1.append(str(n)); c2.append(str(a))
    table._csv_text = tuple(io_native.TextColumn(c) for c in (c0, c1, c2))
    return table._csv_text


def _write_rows_native(fname, delimiter, header_line, Trait, table, sel_i, sel_k, colget, fields, extra, with_emp,
                       extracolstoprint, threads=0):
    """The rows of one results file through the native writer (scoary_results_write: string tables +
    index arrays + numeric columns, formatted by all cores) -- a -p 1.0 run of 50 traits x 125 000
    genes writes six million rows, 15-30 us each through the per-cell Python loop.  Returns False
    when the case is one the general writer keeps (--collapse units, exotic identifiers, no native
    library)."""
    from . import io_native
    if (not io_native.available() or os.environ.get("SCOARY_PY_WRITER") == "1" or len(delimiter) != 1
            or not delimiter.isascii() or Trait.members is not None or Trait._table is None or Trait._rows_idx is None):
        return False
    src = Trait._table
    text = _csv_text_columns(src)
    if text is None:
        return False
    tab_idx = np.asarray(Trait._rows_idx, dtype=np.int64)[sel_i]      # gene-table row of every written row
    text_cols, text_rows = list(text), [tab_idx, tab_idx, tab_idx]
    num = [np.asarray(colget[f])[sel_i] for f in fields]
    if extra is not None:
        num += [np.asarray(extra[k])[sel_k] for k in ("max_total_pairs", "max_propairs", "max_antipairs",
                                                      "Pbest", "Pworst")]
        if with_emp:
            num.append(np.asarray(extra["Empirical_p"])[sel_k])
    # grabbed input columns come after the numeric cells (methods.py:1190-1194): written as a second
    # block of text would break the column order, so those runs keep the general writer
    if extracolstoprint:
        return False
    io_native.results_write(fname, delimiter, header_line, text_cols, text_rows, num,
                            np.arange(len(sel_i), dtype=np.int64), threads=threads or _usable_cpus())
    return True


def _write_rows_python(fname, delimiter, header_line, Trait, table, sel_i, sel_k, colget, fields, extra, with_emp,
                       extracolstoprint):
    """The general writer: one Python string per cell (--collapse units, grabbed input columns)."""
    members = Trait.members
    with open(fname, "w") as out:
        out.write(header_line)
        for n, i in enumerate(sel_i):
            i = int(i)
            k = None if sel_k is None else int(sel_k[n])
            gene = Trait.gene_at(i)                     # only the rows that are written get names
            if "_|_" in gene:
                cells = gene.split("_|_")
            else:
                cells = [gene, str(Trait.nugn_at(i)), str(Trait.annotation_at(i))]
            cells += [_fmt(colget[f][i]) for f in fields]
            if extra is not None:
                cells += [_fmt(extra["max_total_pairs"][k]), _fmt(extra["max_propairs"][k]),
                          _fmt(extra["max_antipairs"][k]), _fmt(extra["Pbest"][k]),
                          _fmt(extra["Pworst"][k])]
                if with_emp:
                    cells.append(_fmt(extra["Empirical_p"][k]))
            for colname in extracolstoprint:
                key = colname + "_name"
                if "--" in gene:
                    parts = members[i] if members is not None else gene.split("--")
                    cells.append("--".join(str(table.extra[key][table.index(g)]) for g in parts))
                else:
                    cells.append(str(table.extra[key][table.index(gene)]))
            out.write(delimiter.join('"' + c + '"' for c in cells) + "\n")


def StoreTraitResult(Trait, Traitname, max_hits, cutoffs, upgmatree, GTC, Prunedic, outdir,
                     permutations, num_threads, no_pairwise, genedic, extracolstoprint,
                     firstcolnames, time="", delimiter=",", seed=DEFAULT_SEED, writer_threads=0):
    """Write ``<outdir><Trait><time>.results.csv`` (methods.py:1003-1197):
    header, rows sorted (stable) by the reference's key, every active cutoff
    applied, every cell double-quoted.  Without ``no_pairwise`` the
    pairwise-comparison stage runs first (see _pairwise_stage)."""
    permutations = int(permutations)
    fname = outdir + Traitname + time + ".results.csv"
    columns = list(firstcolnames) + [
        "Number_pos_present_in", "Number_neg_present_in", "Number_pos_not_present_in",
        "Number_neg_not_present_in", "Sensitivity", "Specificity", "Odds_ratio", "Naive_p",
        "Bonferroni_p", "Benjamini_H_p"]
    if not no_pairwise:
        columns += ["Max_Pairwise_comparisons", "Max_supporting_pairs", "Max_opposing_pairs",
                    "Best_pairwise_comp_p", "Worst_pairwise_comp_p"]
    with_emp = permutations >= 10
    if with_emp:
        columns.append("Empirical_p")
    columns += list(extracolstoprint)
    table = _as_table(genedic) if extracolstoprint else None
    if not isinstance(Trait, TraitResults):
        Trait = _trait_results_from_dict(Trait)

    n = len(Trait)
    # (-m 0 or a negative -m: no rows, as the reference's xrange(num_results), :1024 / :1143)
    num_results = n if max_hits is None else max(0, min(max_hits, n))
    pcol = np.asarray(Trait.column("p_v"), dtype=np.float64)
    order = getattr(Trait, "p_order", None)
    if order is not None and len(order) == n and n > 1:
        # the order Setup_results already had -- valid only while the p column is the one it was
        # taken from (a caller may have replaced it): ascending, ties in row order, O(n) to check
        sp = pcol[order]
        if not np.all((sp[:-1] < sp[1:]) | ((sp[:-1] == sp[1:]) & (order[:-1] < order[1:]))):
            order = None
    if order is None or len(order) != n:
        order = np.argsort(pcol, kind="stable")
    order = order[:num_results]
    fields = ["tpgp", "tngp", "tpgn", "tngn", "sens", "spes", "OR", "p_v", "B_p", "BH_p"]
    if no_pairwise:
        log.info("Skipping population structure-aware analyses." if not with_emp else
                 "Performing %s label permutations per gene on the GPU (Fisher statistic)"
                 % permutations)
        cand = order
        colget = {k: np.asarray(Trait.column(k)) for k in fields}
        if with_emp:
            colget["Empirical_p"] = np.asarray(Trait.column("Empirical_p"))
            fields.append("Empirical_p")
        keyed = {CUT_FIELD[m]: colget[CUT_FIELD[m]] for m in cutoffs}
        sel = cand[np.all([keyed[CUT_FIELD[m]][cand] <= c for m, c in cutoffs.items()], axis=0)] \
            if cutoffs else cand
        rows = np.asarray(sel, dtype=np.int64)          # result rows to write, in order (no per-row objects)
        extra = None
    else:
        log.info("Calculating max number of contrasting pairs for each %s gene%s"
                 % ("significant" if with_emp else "nominally significant",
                    " and performing %d permutations" % permutations if with_emp else ""))
        keep, ranks, extra = _pairwise_stage(Trait, Traitname, order, cutoffs, upgmatree, GTC,
                                             Prunedic, permutations, genedic, seed, num_threads)
        colget = {k: np.asarray(Trait.column(k)) for k in fields}
        # sort key of the filtered set (methods.py:1124-1135); ``keep`` is already
        # in ascending-p order, and every sort is stable
        pos = np.arange(len(keep))
        if num_threads is not None and int(num_threads) > 1 and len(keep):
            # --threads n: the reference hands worker k the ranks k, k+n, k+2n, ... and weaves the
            # workers' results back thread by thread (scoary/methods.py:1076-1078, 1115-1122), so
            # rows with EQUAL sort keys come out in (rank mod n, rank) order
            # -- the modulo is of the gene's rank in the sorted results, not of its position in
            # ``keep`` (the two differ once a worker stopped before another)
            pos = np.lexsort((ranks, ranks % int(num_threads)))
        if any(m in cutoffs for m in ("I", "B", "BH")):
            pos = pos[np.argsort(colget["p_v"][keep][pos], kind="stable")]
        elif "EPW" in cutoffs:
            pos = pos[np.argsort(np.asarray(extra["Pboth"])[pos], kind="stable")]
        elif "PW" in cutoffs:
            pos = pos[np.argsort(np.asarray(extra["Plowest"])[pos], kind="stable")]
        elif "P" in cutoffs:
            pos = pos[np.argsort(np.asarray(extra["Empirical_p"])[pos], kind="stable")]
        else:
            log.info("No filtration applied")
        pos = pos[:min(num_results, len(keep))]
        rows = []
        for k in pos:
            i = int(keep[k])
            ok = True
            for m, c in cutoffs.items():
                f = CUT_FIELD[m]
                v = extra[f][k] if f in extra else colget[f][i]
                ok = ok and (v <= c)
            if ok:
                rows.append((i, int(k)))
    log.info("Storing results to file")
    sel_i = np.fromiter((r[0] for r in rows), dtype=np.int64, count=len(rows)) if not isinstance(rows, np.ndarray) \
        else rows
    sel_k = None if extra is None else np.fromiter((r[1] for r in rows), dtype=np.int64, count=len(rows))
    header_line = delimiter.join('"' + c + '"' for c in columns) + "\n"
    if not _write_rows_native(fname, delimiter, header_line, Trait, table, sel_i, sel_k, colget, fields, extra,
                              with_emp, extracolstoprint, threads=writer_threads):
        _write_rows_python(fname, delimiter, header_line, Trait, table, sel_i, sel_k, colget, fields, extra,
                           with_emp, extracolstoprint)
    return fname


def _trait_results_from_dict(rows):
    """Plain {gene: row dict} (the reference's Results[trait]) -> TraitResults."""
    genes = list(rows.keys())
    cols = {}
    for k in TraitResults.FIELDS + ("Empirical_p",):
        if genes and k in rows[genes[0]]:
            cols[k] = np.array([rows[g][k] for g in genes])
    return TraitResults(genes, [rows[g]["NUGN"] for g in genes],
                        [rows[g]["Annotation"] for g in genes], cols, len(genes))


def StoreUPGMAtreeToFile(upgmatree, outdir, time=""):
    """Write the tree as ``Tree<time>.nwk`` (methods.py:741-752)."""
    from .tree import newick_text
    name = str(outdir + ("Tree%s.nwk" % time))
    with open(name, "w") as f:
        f.write(newick_text(upgmatree) if upgmatree is not None else "None;")
    log.info("Wrote the UPGMA tree to file: %s" % name)
    return name


def filtrationoptions(cutoffs, collapse):
    long_names = {"I": "Individual (Naive)", "B": "Bonferroni", "BH": "Benjamini-Hochberg",
                  "PW": "Pairwise comparison (Best)", "EPW": "Pairwise comparison (Entire range)",
                  "P": "Empirical p-value (permutation-based)"}
    lines = ["-- Filtration options --"]
    lines += ["%s:    %s" % (long_names[k], v) for k, v in cutoffs.items()]
    lines.append("Collapse genes:    %s\n\n" % collapse)
    return lines


# ---------------------------------------------------------------------------
# CLI
# ---------------------------------------------------------------------------
def grabcoltype(string):
    """--include_input_columns: "4,6,8,16-23" / "ALL" -> 0-based column list
    (methods.py:1510-1549; ranges are half-open there, kept as is)."""
    if string == "ALL":
        return [-999]
    if string == "":
        return []
    cols = []
    for part in string.split(","):
        if "-" in part:
            try:
                ab = part.split("-")
                if not int(ab[1]) > int(ab[0]):
                    raise ValueError(part)
                cols += list(range(int(ab[0]), int(ab[1])))
            except (ValueError, TypeError, IndexError):
                sys.exit("Could not understand --include_input_columns argument %s" % part)
        else:
            # a word where a number belongs is a ValueError here as in the reference (which only catches TypeError,
            # :1537-1541): argparse turns it into its own "invalid grabcoltype value" error, exit status 2
            cols.append(int(part))
    cols = [c - 1 for c in cols if c - 1 not in (0, 1, 2)]
    if not all(c > 1 for c in cols):
        sys.exit("Could not understand --include_input_columns argument. Make sure all "
                 "numbers are positive and real.")
    return list(set(cols))


def ScoaryArgumentParser(argv=None):
    """Same flags, dests, types and defaults as methods.py:1551-1744, plus
    --seed for reproducible permutations."""
    ap = argparse.ArgumentParser(
        description="scoary_amd %s - MI355X-native pan-genome association "
                    "(Scoary %s compatible command line)" % (__version__, SCOARY_COMPAT_VERSION))
    g = ap.add_argument_group("Input options")
    g.add_argument("-t", "--traits", help="Trait table (csv): strains in rows, traits in "
                   "columns, 1 = trait present, 0 = absent, NA/./- = missing")
    g.add_argument("-g", "--genes", help="Gene presence/absence table (csv, Roary format); "
                   "strain names must match the trait table")
    g.add_argument("-n", "--newicktree", default=None,
                   help="Custom Newick tree for the pairwise-comparison stage")
    g.add_argument("-s", "--start_col", type=int, default=15,
                   help="1-based column where isolates start in the gene table (default 15)")
    g.add_argument("--delimiter", type=str, default=",",
                   help="Single-character cell delimiter of input and output files")
    g.add_argument("-r", "--restrict_to", help="File with a comma-separated list of isolates "
                   "to restrict the analysis to")
    o = ap.add_argument_group("Output options")
    o.add_argument("-o", "--outdir", default="./", help="Output directory (default .)")
    o.add_argument("-u", "--upgma_tree", action="store_true", default=False,
                   help="Write the calculated UPGMA tree to a newick file")
    o.add_argument("-p", "--p_value_cutoff", nargs="+", type=float, default=[0.05],
                   help="P-value cut-off(s): one for all correction methods or one per "
                   "method in -c order (default 0.05; 1.0 reports every gene)")
    o.add_argument("-c", "--correction", nargs="*", default=["I"],
                   choices=["I", "B", "BH", "PW", "EPW", "P"],
                   help="Filtration measures: I naive, B Bonferroni, BH Benjamini-Hochberg, "
                   "PW best pairwise, EPW entire pairwise range, P empirical (permutation)")
    o.add_argument("-m", "--max_hits", type=int, help="Report at most this many hits per trait")
    o.add_argument("--include_input_columns", dest="grabcols", type=grabcoltype, default=[],
                   help="Columns of the gene table to copy to the output, e.g. 4,6,8,16-23 or ALL")
    o.add_argument("-w", "--write_reduced", action="store_true", default=False,
                   help="With -r: also write the reduced gene presence/absence table")
    o.add_argument("--no-time", dest="no_time", action="store_true", default=False,
                   help="No timestamp in output file names")
    a = ap.add_argument_group("Analysis options")
    a.add_argument("-e", "--permute", type=int, default=0,
                   help="Number of trait-label permutations per gene for empirical p-values "
                   "(0 = off, minimum 10)")
    a.add_argument("--permute-early-abort", dest="permute_early_abort", action="store_true",
                   default=False,
                   help="With --no_pairwise --permute: use the reference's sequential estimator "
                   "with early abort for the Fisher-statistic permutations -- a gene stops after "
                   "i >= 30 permutations once 1 - binom.cdf(r, i, 0.1) < 0.05 and gets "
                   "(r+1)/(i+2) (large empirical p-values become coarse, like the reference's)")
    a.add_argument("--no_pairwise", action="store_true", default=False,
                   help="Population-structure-naive analysis only (Fisher's test, odds ratios)")
    a.add_argument("--collapse", action="store_true", default=False,
                   help="Merge genes with identical distribution patterns into one unit")
    a.add_argument("--seed", type=int, default=DEFAULT_SEED,
                   help="Seed of the counter-based permutation generator (scoary_amd extension)")
    m = ap.add_argument_group("Misc options")
    m.add_argument("--threads", type=int, default=1,
                   help="Accepted for compatibility; the GPU path does not use host threads")
    m.add_argument("--test", action="store_true", default=False,
                   help="Run on the bundled example data (needs the path in SCOARY_EXAMPLEDATA)")
    m.add_argument("--citation", action="store_true", default=False,
                   help="Show citation information and exit")
    m.add_argument("--version", action="version", version=SCOARY_COMPAT_VERSION)
    args = ap.parse_args(argv)
    if len(args.p_value_cutoff) == 1:
        cutoffs = {c: args.p_value_cutoff[0] for c in args.correction}
    else:
        cutoffs = dict(zip(args.correction, args.p_value_cutoff))
    return args, cutoffs


CITATION = ("If you use Scoary, please cite: Brynildsrud O, Bohlin J, Scheffer L, Eldholm V. "
            "Rapid scoring of genes in microbial pan-genome-wide association studies with "
            "Scoary. Genome Biol. 2016;17:238.  (scoary_amd re-implements Scoary's Fisher / "
            "permutation path for AMD MI355X.)")


def main(**kwargs):
    """The command-line flow of methods.py:49-330 for the association path."""
    if len(kwargs) == 0:
        args, cutoffs = ScoaryArgumentParser()
    else:
        args, cutoffs = kwargs["args"], kwargs["cutoffs"]
        if "statusbar" in kwargs:
            sys.stdout = kwargs["statusbar"]
    if args.citation:
        sys.exit(CITATION)
    if args.test:
        ex = os.environ.get("SCOARY_EXAMPLEDATA")
        if not ex:
            sys.exit("--test needs SCOARY_EXAMPLEDATA=<dir with Gene_presence_absence.csv and "
                     "Tetracycline_resistance.csv>")
        args.correction, args.delimiter, args.grabcols = ["I", "EPW"], ",", []
        args.genes = os.path.join(ex, "Gene_presence_absence.csv")
        args.traits = os.path.join(ex, "Tetracycline_resistance.csv")
        args.max_hits = args.newicktree = args.restrict_to = None
        args.no_pairwise, args.outdir, args.permute = False, "./", 0
        args.p_value_cutoff, args.start_col, args.threads = [0.05, 0.05], 15, 4
        args.upgma_tree, args.write_reduced, args.no_time, args.collapse = True, False, False, False
        cutoffs = {"I": 0.05, "EPW": 0.05}

    from . import dist
    world, rank, _local = dist.init_from_env()
    start = _time.time()
    stamp = "" if args.no_time else _time.strftime("_%d_%m_%Y_%H%M")
    if not args.outdir.endswith("/"):
        args.outdir += "/"
    os.makedirs(args.outdir, exist_ok=True)
    console = logging.StreamHandler(sys.stdout)
    console.setFormatter(logging.Formatter("%(message)s"))
    console.setLevel(logging.INFO)
    logfile = logging.FileHandler(os.path.join(
        args.outdir, "scoary%s%s.log" % (stamp, "" if rank == 0 else ".rank%d" % rank)), mode="w")
    logfile.setFormatter(logging.Formatter("%(asctime)s    %(message)s", "%m/%d/%Y %I:%M:%S %p"))
    log.addHandler(console)
    log.addHandler(logfile)
    counts = _LevelCounter()
    log.addHandler(counts)
    log.info("==== Scoary started ====")
    log.info("Command: " + " ".join(sys.argv))
    starter = None
    try:
        _validate(args, cutoffs)
        seed = getattr(args, "seed", DEFAULT_SEED)
        allowed = None
        if args.restrict_to is not None:
            with open(args.restrict_to, "r") as f:
                allowed = {iso: "all" for line in f for iso in line.rstrip().split(",")}
        elif args.write_reduced:
            sys.exit("You cannot use the -w argument without specifying a subset (-r)")
        _stage.reset()
        # The first GPU call of the process -- torch import, library load, HIP context, warm-up: about a
        # second -- and the read of the gene table need nothing from each other, and the native reader
        # runs outside the interpreter lock: a single process starts the engine on a helper thread and
        # joins it after the read (a 2.5 GB table reads in 0.6 s: the second shrinks to what the read
        # does not cover).  Under torchrun the process group already imported torch and picked the device.
        if world == 1 and not dist._initialized_safe() and os.environ.get("SCOARY_OVERLAP_STARTUP", "1") == "1":
            import threading
            started = {}

            def _start_engine():
                t0 = _time.perf_counter()
                try:
                    get_engine()
                except BaseException as e:          # re-raised on the main thread at the join
                    started["error"] = e
                started["seconds"] = _time.perf_counter() - t0
            starter = threading.Thread(target=_start_engine, name="scoary-engine-start", daemon=True)
            starter.start()
        else:
            with _stage("import torch, HIP library, engine"):
                get_engine()
        with open(args.genes, "r", newline=None) as genes, \
                open(args.traits, "r", newline=None) as traits:
            log.info("Reading gene presence absence file")
            grab = [-999] if args.grabcols == "ALL" else args.grabcols
            with _stage("read gene table"):
                gd = Csv_to_dic_Roary(genes, args.delimiter, grab, startcol=int(args.start_col) - 1,
                                      allowed_isolates=allowed, writereducedset=args.write_reduced,
                                      time=stamp, outdir=args.outdir)
            if starter is not None:
                with _stage("engine start not covered by the read"):
                    starter.join()
                _stage.seconds["import torch, HIP library, engine (helper thread, under the read)"] = \
                    started.get("seconds", 0.0)
                if "error" in started:
                    raise started["error"]
            genedic, strains = gd["Roarydic"], gd["Strains"]
            upgmatree = None
            if args.newicktree is None and not args.no_pairwise:
                log.info("Creating Hamming distance matrix based on gene presence/absence")
                from . import tree as T
                log.info("Building UPGMA tree from distance matrix")
                upgmatree = T.upgma(get_engine(), gd["Zero_ones_matrix"].file_rows(), strains)
                _preload_scipy()
            elif args.no_pairwise:
                log.info("Ignoring relatedness among input sample and performing only "
                         "population structure-naive analysis.")
            else:
                log.info("Reading custom tree file")
                from . import tree as T
                upgmatree, members = T.read_newick(args.newicktree)
                if sorted(strains) != sorted(members):
                    if args.restrict_to is None:
                        sys.exit("CRITICAL: Please make sure that isolates in your custom tree "
                                 "match those in your gene presence absence file.")
                    if not all(i in members for i in strains):
                        sys.exit("CRITICAL: Your provided tree file did not contain all the "
                                 "isolates in your gene presence absence file.")
                    log.info("Pruning phylogenetic tree to correspond to set of included isolates")
                    keepset = set(strains)
                    upgmatree = T.prune_missing(upgmatree, [i for i in members if i not in keepset])
            log.info("Reading traits file")
            with _stage("read traits"):
                traitsdic, prunedic = Csv_to_dic(traits, args.delimiter, allowed, strains)
        t_loaded = _time.time()
        log.info("Finished loading files into memory.\n\n")
        log.info("==== Performing statistics ====")
        for line in filtrationoptions(cutoffs, args.collapse):
            log.info(line)
        log.info("Tallying genes and performing statistical analyses")
        # --no_pairwise: Fisher-statistic permutations for every gene (north_star);
        # default mode: tree-statistic permutations of the surviving genes, done
        # in the pairwise stage like the reference does.
        res = Setup_results(genedic, traitsdic, args.collapse,
                            permutations=args.permute if args.no_pairwise else 0, seed=seed,
                            early_abort=getattr(args, "permute_early_abort", False))
        t_stats = _time.time()
        if args.upgma_tree and rank == 0:
            # (with --no_pairwise there is no tree and the reference writes str(None) + ";", :277-280, :741-751)
            StoreUPGMAtreeToFile(upgmatree, args.outdir, time=stamp)
        if rank == 0:           # every rank holds the gathered results; one writes
            StoreResults(res["Results"], args.max_hits, cutoffs, upgmatree,
                         res["Gene_trait_combinations"], prunedic, args.outdir, args.permute,
                         args.threads, args.no_pairwise, genedic, gd["Extracols"],
                         gd["Firstcolnames"], time=stamp, delimiter=args.delimiter, seed=seed)
        log.info("Stage seconds: load (+tree) %.2f, association (+permutations) %.2f, pairwise stage and "
                 "output %.2f" % (t_loaded - start, t_stats - t_loaded, _time.time() - t_stats))
        gene_bytes = os.path.getsize(args.genes) if os.path.isfile(args.genes) else 0
        log.info("Stage detail: " + "; ".join("%s %.3f s" % kv for kv in _stage.seconds.items())
                 + ("; gene table %.0f MB" % (gene_bytes / 1e6) if gene_bytes else ""))
        log.info("\n")
        log.info("==== Finished ====")
        log.info("Checked a total of %d genes for associations to %d trait(s). Total time "
                 "used: %d seconds." % (len(genedic), len(traitsdic), int(_time.time() - start)))
    except SystemExit as e:
        if starter is not None:
            starter.join(120)          # never tear the interpreter down under a thread that is inside HIP start-up
        log.exception("CRITICAL:")
        for hnd in (logfile, console, counts):
            log.removeHandler(hnd)
        dist.shutdown()
        sys.exit(e.code)
    except BaseException:
        if starter is not None:
            starter.join(120)
        raise
    if counts.n.get("CRITICAL", 0):
        log.info("Scoary finished successfully, but with CRITICAL ERRORS. Please check your log file.")
    elif counts.n.get("ERROR", 0):
        log.info("Scoary finished successfully, but with ERRORS. Please check your log file.")
    elif counts.n.get("WARNING", 0):
        log.info("Scoary finished successfully, but with WARNINGS. Please check your log file.")
    else:
        log.info("No warnings were recorded.")
    for hnd in (logfile, console, counts):
        log.removeHandler(hnd)
    logfile.close()
    dist.shutdown()
    sys.exit(0)


class _LevelCounter(logging.Handler):
    def __init__(self):
        super().__init__(logging.WARNING)
        self.n = {}

    def emit(self, record):
        self.n[record.levelname] = self.n.get(record.levelname, 0) + 1


def _validate(args, cutoffs):
    """Argument checks of methods.py:126-181 (same conditions, same exits)."""
    if args.traits is None or args.genes is None:
        sys.exit("The following arguments are required: -t/--traits, -g/--genes")
    if args.threads <= 0:
        sys.exit("Number of threads must be positive")
    if not os.path.isfile(args.traits):
        sys.exit("Could not find the traits file: %s" % args.traits)
    if not os.path.isfile(args.genes):
        sys.exit("Could not find the gene presence absence file: %s" % args.genes)
    if args.newicktree is not None and not os.path.isfile(args.newicktree):
        sys.exit("Could not find the custom tree file: %s" % args.newicktree)
    if not all(0.0 < p <= 1.0 for p in args.p_value_cutoff):
        sys.exit("P must be between 0.0 and 1.0 or exactly 1.0")
    if len(args.delimiter) > 1:
        sys.exit("Delimiter must be a single character string. There is no support for tab.")
    if len(args.p_value_cutoff) != len(args.correction) and len(args.p_value_cutoff) != 1:
        sys.exit("You can not use more p-value cutoffs than correction methods. Either provide "
                 "a single p-value that will be applied to all correction methods, or provide "
                 "exactly as many as the number of correction methods and in corresponding "
                 "sequence. e.g. -c I EPW -p 0.1 0.05 will apply an individual p-value cutoff "
                 "of 0.1 AND a pairwise comparisons p-value cutoff of 0.05.")
    if "P" in cutoffs and args.permute == 0:
        sys.exit("Cannot use empirical p-values in filtration without performing "
                 "permutations. Use '--permute X' where X is a number equal to or larger than 10")
    if args.permute < 10 and args.permute != 0:
        sys.exit("The absolute minimum number of permutations is 10 (or 0 to deactivate)")
    if "P" in cutoffs and cutoffs["P"] < (1.0 / args.permute):
        sys.exit("Permutation cutoff too low for this number of permutations")
    if args.permute > 10000:
        log.info("Note: You have set Scoary to do a high number of permutations. "
                 "This may take a while.")
    if args.no_pairwise:
        # The reference also zeroes --permute here (methods.py:174-181) because
        # its permutations need the tree.  The Fisher-statistic permutations of
        # this build do not, so --permute stays active (documented extension).
        log.info("Performing no pairwise comparisons. Ignoring all tree related options "
                 "(user tree, population aware-correction).")
        args.newicktree = None
        for m in ("PW", "EPW"):
            cutoffs.pop(m, None)


if __name__ == "__main__":
    main()
