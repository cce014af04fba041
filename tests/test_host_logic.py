"""CPU-side tests: host logic of the path (readers, B/BH, argument parser, CSV
formatting) and the shape of the C-ABI -- no GPU compute here."""
import ctypes
import io
import json
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, golden_text, read_dense


# ---------------------------------------------------------------- C-ABI ------
def _header_functions():
    with open(os.path.join(ROOT, "include", "scoary_hip.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(scoary_[a-z0-9_]+)\s*\(", src)))


def test_abi_library_exports_every_declared_symbol():
    """The C-ABI library loads without a GPU and exports exactly what
    include/scoary_hip.h declares; the ctypes binding covers all of it."""
    from scoary_amd import _abi
    if not os.path.exists(_abi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(_abi.LIB_PATH)
    declared = _header_functions()
    assert len(declared) >= 15
    for name in declared:
        assert hasattr(lib, name), "library does not export " + name
    assert sorted(_abi.SIGNATURES) == declared
    lib.scoary_abi_version.restype = ctypes.c_int
    assert lib.scoary_abi_version() == _abi.ABI_VERSION


def test_abi_layout_arithmetic():
    from scoary_amd import _abi
    lib = _abi.load()
    for N, quads in [(1, 1), (32, 1), (128, 1), (129, 2), (500, 4), (2000, 16), (2048, 16),
                     (2049, 20), (5000, 40), (6144, 48), (6145, 56), (10000, 80)]:
        assert lib.scoary_tiled_quads(N) == quads, N
        assert lib.scoary_row_words(N) == 4 * quads
        assert 4 * quads * 32 >= N
    for G, gp in [(1, 256), (256, 256), (257, 512), (50000, 50176)]:
        assert lib.scoary_tiled_genes(G) == gp
    assert lib.scoary_tiled_bytes(50000, 2000) == 16 * 16 * 50176


def test_no_gpu_means_loud_failure():
    """The product path never falls back to the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from scoary_amd import _abi
    from scoary_amd.engine import AssociationEngine
    with pytest.raises(_abi.ScoaryHipError):
        AssociationEngine()


def test_command_line_without_a_gpu_fails_loudly_from_the_helper_thread(exampledir, tmp_path):
    """The command line starts the engine (torch import, HIP context) on a helper thread while the main
    thread reads the gene table.  Without a GPU that start fails: the error must come back on the main
    thread as a clean non-zero exit naming the cause -- no CPU fallback, and no crash of the interpreter
    (round 6: an `import torch.distributed` on the main thread deadlocked against the helper's
    `import torch`; the importer's _DeadlockError surfaced inside torch's C++ start-up as
    `terminate called after throwing an instance of 'python_error'`)."""
    import subprocess
    import sys
    if __import__("torch").cuda.is_available():
        pytest.skip("a GPU is visible: the start succeeds")
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = subprocess.run([sys.executable, "-m", "scoary_amd", "-g", os.path.join(exampledir, "Gene_presence_absence.csv"),
                          "-t", os.path.join(exampledir, "Tetracycline_resistance.csv"), "--no_pairwise",
                          "-o", str(tmp_path / "out")], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    text = out.stdout + out.stderr
    assert out.returncode not in (0, -6, 134), text[-2000:]
    assert "no GPU visible" in text and "terminate called" not in text
    assert not [f for f in os.listdir(tmp_path / "out") if f.endswith(".results.csv")]


def test_product_package_never_imports_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "scoary_amd")):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                with open(os.path.join(dirpath, fn)) as f:
                    src = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), fn
                assert "liboracle" not in src and "orc_" not in src, fn
                assert not re.search(r"#include\s+[\"<][^\n]*oracle", src), fn


# -------------------------------------------------------------- readers ------
def test_gpa_reader_matches_reference_shapes(exampledir, manifest):
    from scoary_amd import methods as m
    with open(os.path.join(exampledir, "Gene_presence_absence.csv")) as f:
        gd = m.Csv_to_dic_Roary(f, ",", [], startcol=14)
    assert gd["Strains"] == manifest["strains"]
    assert gd["Firstcolnames"] == ["Gene", "Non-unique Gene name", "Annotation"]
    table = gd["Roarydic"]
    z = np.load(os.path.join(GOLDEN, "setup_results_exampledata.npz"))
    assert list(table) == json.loads(str(z["all_genes"]))
    ids, strains, genes, names, traits = read_dense(
        golden_text("exampledata/Gene_presence_absence.csv.gz"),
        golden_text("exampledata/Tetracycline_resistance.csv.gz"))
    assert np.array_equal(table.dense(), genes)
    # mapping view has the reference's genedic shape
    row = table["TetRCG"]
    assert row["Annotation"].startswith("A fictitious gene") and row["Isolate_1"] in (0, 1)
    assert sum(row[s] for s in strains) == int(genes[ids.index("TetRCG")].sum())
    # Zero_ones_matrix: strain-major, variable genes only (methods.py:496-502)
    zm = np.array(gd["Zero_ones_matrix"])
    tot = genes.sum(1)
    assert zm.shape == (len(strains), int(((tot > 0) & (tot < len(strains))).sum()))


def test_traits_reader_missing_values_and_prunedic(exampledir, manifest):
    from scoary_amd import methods as m
    with open(os.path.join(exampledir, "Tetracycline_resistance.csv")) as f:
        td, prune = m.Csv_to_dic(f, ",", None, manifest["strains"])
    assert list(td) == ["Tetracycline_resistance", "Bogus_trait"]
    assert len(td["Tetracycline_resistance"]) == 100 and len(td["Bogus_trait"]) == 97
    for k, v in manifest["prune"].items():
        assert prune[k] == v + [None]
    assert set(td["Bogus_trait"].values()) == {"0", "1"}


def test_restrict_to_reader(exampledir, manifest):
    from scoary_amd import methods as m
    with open(os.path.join(exampledir, "Restrict_to.csv")) as f:
        allowed = {iso: "all" for line in f for iso in line.rstrip().split(",")}
    with open(os.path.join(exampledir, "Gene_presence_absence.csv")) as f:
        gd = m.Csv_to_dic_Roary(f, ",", [], startcol=14, allowed_isolates=allowed)
    assert gd["Strains"] == manifest["restrict_strains"]


def test_non_roary_reader_vcf(exampledir, manifest):
    from scoary_amd import methods as m
    with open(os.path.join(exampledir, "mutations_presence_absence.csv")) as f:
        gd = m.Csv_to_dic_Roary(f, ",", [], startcol=manifest["vcf_startcol_1based"] - 1)
    z = np.load(os.path.join(GOLDEN, "setup_results_vcf.npz"))
    assert list(gd["Roarydic"]) == json.loads(str(z["all_genes"]))   # duplicate ids collapse
    assert gd["Firstcolnames"] == manifest["vcf_firstcolnames"]
    assert gd["Strains"] == manifest["vcf_strains"]
    # the tree-building matrix holds every FILE row (8), not the 5 de-duplicated identifiers
    # (the reference appends each row as it reads it, scoary/methods.py:445-497) -- both readers
    assert len(gd["Roarydic"]) == 5 and manifest["vcf_file_rows"] == 8
    assert [list(r) for r in gd["Zero_ones_matrix"]] == manifest["vcf_zero_ones_matrix"]
    assert gd["Zero_ones_matrix"].file_rows().shape == (8, len(gd["Strains"]))
    os.environ["SCOARY_PY_CSV"] = "1"
    try:
        with open(os.path.join(exampledir, "mutations_presence_absence.csv")) as f:
            gd2 = m.Csv_to_dic_Roary(f, ",", [], startcol=manifest["vcf_startcol_1based"] - 1)
    finally:
        del os.environ["SCOARY_PY_CSV"]
    assert [list(r) for r in gd2["Zero_ones_matrix"]] == manifest["vcf_zero_ones_matrix"]


def test_traits_reader_rejects_bad_input():
    from scoary_amd import methods as m
    with pytest.raises(SystemExit):
        m.Csv_to_dic(io.StringIO(",T\nA,2\nB,0\n"), ",", None, ["A", "B"])
    with pytest.raises(SystemExit):
        m.Csv_to_dic(io.StringIO("X,T\nA,1\nB,0\n"), ",", None, ["A", "B"])
    with pytest.raises(SystemExit):
        m.Csv_to_dic(io.StringIO("Name\nA\n"), ",", None, ["A"])


# ---------------------------------------------------------------- B / BH -----
def test_bonferroni_bh_bit_exact_vs_reference():
    """Vectorised B/BH == the reference's sequential loop, bit for bit, on the
    reference's own p-values (both traits of exampledata)."""
    from scoary_amd.methods import bonferroni_bh
    z = np.load(os.path.join(GOLDEN, "setup_results_exampledata.npz"))
    for t in range(2):
        p = z["t%d_p_v" % t]
        B, BH = bonferroni_bh(p, len(p))
        assert np.array_equal(B, z["t%d_B_p" % t])
        assert np.array_equal(BH, z["t%d_BH_p" % t])


def test_bonferroni_bh_ties_and_loop_equivalence():
    from oracle import oracle as orc
    from scoary_amd.methods import bonferroni_bh
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 17, 400):
        p = rng.choice(np.concatenate([rng.random(max(1, n // 3)), [1.0, 1e-300, 0.5]]), size=n)
        for ntests in (n, n + 5, max(1, n - 3)):
            B, BH = bonferroni_bh(p, ntests)
            B2, BH2 = orc.bonferroni_bh([float(x) for x in p], ntests)
            assert np.array_equal(B, np.array(B2)) and np.array_equal(BH, np.array(BH2))
    with pytest.raises(IndexError):
        bonferroni_bh([], 0)


# ------------------------------------------------------------------- CLI -----
def test_argument_parser_defaults_and_cutoffs():
    from scoary_amd.methods import ScoaryArgumentParser
    args, cut = ScoaryArgumentParser(["-g", "g.csv", "-t", "t.csv"])
    assert (args.start_col, args.delimiter, args.permute, args.threads) == (15, ",", 0, 1)
    assert args.no_pairwise is False and args.collapse is False and args.outdir == "./"
    assert cut == {"I": 0.05}
    args, cut = ScoaryArgumentParser(["-g", "g", "-t", "t", "-c", "I", "BH", "-p", "0.1", "0.01",
                                      "-e", "100", "--no_pairwise", "--no-time", "-m", "5"])
    assert cut == {"I": 0.1, "BH": 0.01} and args.permute == 100 and args.max_hits == 5
    args, cut = ScoaryArgumentParser(["-g", "g", "-t", "t", "-c", "B", "P", "-p", "0.2"])
    assert cut == {"B": 0.2, "P": 0.2}
    with pytest.raises(SystemExit):
        ScoaryArgumentParser(["-c", "XYZ"])


def test_grabcoltype():
    from scoary_amd.methods import grabcoltype
    assert grabcoltype("ALL") == [-999] and grabcoltype("") == []
    assert sorted(grabcoltype("4,6,8,16-19")) == [3, 5, 7, 15, 16, 17]
    assert sorted(grabcoltype("1,2,3,4")) == [3]
    with pytest.raises(SystemExit):
        grabcoltype("9-4")


def test_validation_exits(tmp_path):
    from scoary_amd import methods as m
    g = tmp_path / "g.csv"; g.write_text("x")
    t = tmp_path / "t.csv"; t.write_text("x")
    base = ["-g", str(g), "-t", str(t)]
    for extra in (["-e", "5"], ["-p", "1.5"], ["--delimiter", ";;"], ["--threads", "0"],
                  ["-c", "P"], ["-c", "I", "B", "-p", "0.1", "0.2", "0.3"],
                  ["-c", "P", "-e", "10", "-p", "0.01"]):
        args, cut = m.ScoaryArgumentParser(base + extra)
        with pytest.raises(SystemExit):
            m._validate(args, cut)
    args, cut = m.ScoaryArgumentParser(["-t", str(t)])
    with pytest.raises(SystemExit):
        m._validate(args, cut)
    args, cut = m.ScoaryArgumentParser(base + ["--no_pairwise", "-c", "I", "EPW", "-e", "50"])
    m._validate(args, cut)
    assert "EPW" not in cut and args.permute == 50     # documented extension: permute survives


def test_cell_formatting_is_shortest_roundtrip():
    from scoary_amd.methods import _fmt
    assert _fmt(np.int32(29)) == "29"
    assert _fmt(90.625) == "90.625" and _fmt(np.float64(88.23529411764706)) == "88.23529411764706"
    assert _fmt(np.float64(1.0862106610751687e-14)) == "1.0862106610751687e-14"
    assert _fmt(float("inf")) == "inf" and _fmt(float("nan")) == "nan" and _fmt(0.0) == "0.0"
    assert _fmt(np.float64(1.0)) == "1.0"


def test_native_float_cells_equal_python_repr():
    """The native results writer prints float cells as Python's repr(float) (= str() of the
    reference's numpy.float64 cells): checked value by value on random bit patterns,
    the decades around the fixed / exponent switch (1e-4, 1e16), subnormals, and the specials."""
    from scoary_amd import io_native
    rng = np.random.default_rng(7)
    vals = [0.0, -0.0, 1.0, 72.5, 1e-4, 9.999e-5, 1e-5, 1e15, 1e16, 9999999999999998.0, 1e17, 0.1, 1 / 3,
            5e-324, 2.2250738585072014e-308, 1.7976931348623157e308, float("inf"), float("-inf"), float("nan"),
            100.0, 88.23529411764706, 1.0862106610751687e-14, 2.5700921296104394e-219, 1e22, 1e23,
            9007199254740993.0, 0.30000000000000004, 6.45209132679e-11]
    vals += list(rng.integers(0, 2**64, 20000, dtype=np.uint64).view(np.float64))
    vals += list(rng.random(5000)) + list(rng.random(5000) ** 30) + list(np.round(rng.random(5000) * 100, 3))
    vals += list(10.0 ** rng.integers(-320, 308, 3000)) + [float(i) for i in range(300)]
    for v in vals:
        assert io_native.format_float_repr(float(v)) == repr(float(v)), repr(float(v))


@pytest.mark.parametrize("roary", [True, False])
def test_native_results_writer_equals_the_python_writer(tmp_path, roary):
    """StoreTraitResult's native row writer (scoary_results_write: string tables, index arrays and
    numeric columns, formatted on all cores) against the general per-cell Python writer, byte for
    byte: Roary identifiers and c0_|_c1_|_c2 identifiers, a filtered and re-ordered row selection,
    inf / nan / 0.0 / tiny cells, with and without the pairwise columns; an empty selection writes
    the header alone."""
    from scoary_amd import methods as m
    rng = np.random.default_rng(3)
    G, N = 5000, 70
    ids = ["gene_%d" % i for i in range(G)] if roary else ["chr%d_|_%d_|_id%d" % (i % 3, 1000 + i, i) for i in range(G)]
    table = m.GeneTable(ids, ["nu%d" % (i % 7) if i % 5 else "" for i in range(G)],
                        ['ann "%d", x' % i if i % 11 == 0 else "ann%d" % i for i in range(G)],
                        ["s%d" % j for j in range(N)], rng.integers(0, 2**62, (G, 2), dtype=np.uint64), {})
    rows_idx = np.sort(rng.choice(G, 4000, replace=False))
    n = len(rows_idx)
    odds = rng.random(n) * 50
    odds[::17] = np.inf
    odds[5::29] = np.nan
    odds[3::31] = 0.0
    cols = {"tpgp": rng.integers(0, 70, n).astype(np.int32), "tngp": rng.integers(0, 70, n).astype(np.int32),
            "tpgn": rng.integers(0, 70, n).astype(np.int32), "tngn": rng.integers(0, 70, n).astype(np.int32),
            "sens": np.round(rng.random(n) * 100, 2), "spes": rng.random(n) * 100, "OR": odds,
            "p_v": rng.random(n) ** 40, "B_p": np.minimum(rng.random(n) ** 20 * 10, 1.0), "BH_p": rng.random(n),
            "Empirical_p": (rng.integers(0, 1000, n) + 1.0) / 1001.0}
    tr = m.TraitResults(None, None, None, cols, n, None, table=table, rows_idx=rows_idx)
    fields = ["tpgp", "tngp", "tpgn", "tngn", "sens", "spes", "OR", "p_v", "B_p", "BH_p"]
    header = '"a","b"\n'
    for with_extra in (False, True):
        sel_i = rng.permutation(n)[:2500]
        sel_k = extra = None
        f2 = fields + ["Empirical_p"]
        if with_extra:
            sel_k = np.arange(len(sel_i))[::-1].copy()
            extra = {"max_total_pairs": rng.integers(0, 30, len(sel_i)), "max_propairs": rng.integers(0, 30, len(sel_i)),
                     "max_antipairs": rng.integers(0, 30, len(sel_i)), "Pbest": rng.random(len(sel_i)) ** 9,
                     "Pworst": rng.random(len(sel_i)), "Empirical_p": rng.random(len(sel_i))}
            f2 = fields
        a, b = str(tmp_path / "native.csv"), str(tmp_path / "python.csv")
        assert m._write_rows_native(a, ",", header, tr, None, sel_i, sel_k, tr.cols, f2, extra, True, [])
        m._write_rows_python(b, ",", header, tr, None, sel_i, sel_k, tr.cols, f2, extra, True, [])
        assert open(a, "rb").read() == open(b, "rb").read()
        assert open(a).read().count("\n") == 2501
    assert m._write_rows_native(a, ";", header, tr, None, np.zeros(0, dtype=np.int64), None, tr.cols, fields, None,
                                False, [])
    assert open(a).read() == header
    # --collapse units and grabbed input columns stay with the general writer
    assert not m._write_rows_native(a, ",", header, tr, table, sel_i, None, tr.cols, fields, None, False, ["QC"])


# -------------------------------------------------------------- vcf2scoary ---
def test_vcf2scoary_matches_reference_output(exampledir, tmp_path):
    """tests/test_scoary_output.py:16-17,123-136 of the reference pins the first
    row; the whole converted file is compared here (captured from the reference)."""
    from scoary_amd import vcf2scoary as v
    out = tmp_path / "mpa.csv"
    with pytest.raises(SystemExit) as e:
        v.main(["--force", "--out", str(out), os.path.join(exampledir, "Example.vcf")])
    assert e.value.code == 0
    got = out.read_text()
    assert got == golden_text("exampledata/mutations_presence_absence.csv.gz")
    first = got.splitlines()[1].replace('"', "").split(",")
    assert first == ["NC_000962", "4013", "0", "T", "C", "9999", "0", "TYPE=snp", "GT", "False",
                     "0", "1", "1", "1"]
    with pytest.raises(SystemExit):                       # refuses to overwrite
        v.main(["--out", str(out), os.path.join(exampledir, "Example.vcf")])
    io_out = io.StringIO()
    with open(os.path.join(exampledir, "Example.vcf")) as f:
        n = v.convert(f, io_out, types=["ins"], log=lambda *a: None)
    assert n == 0 and io_out.getvalue().count("\n") == 1


def _vcf_text(rng, V, S, crlf=False, extra=()):
    nl = "\r\n" if crlf else "\n"
    lines = ["##fileformat=VCFv4.2",
             '##FORMAT=<ID=GT,Number=1,Type=String,Description="Genotype">',
             '##INFO=<ID=TYPE,Number=A,Type=String,Description="type, of variant">',
             "\t".join(["#CHROM", "POS", "ID", "REF", "ALT", "QUAL", "FILTER", "INFO", "FORMAT"] +
                       ["s%d" % i for i in range(S)])]
    kinds = ["snp", "ins", "del", "mnp"]
    for v in range(V):
        nalt = 1 + (v % 5 == 0) + (v % 10 == 0)
        alt = ",".join("ACGT"[a] for a in range(nalt)) if v % 20 else "A,,T"[:2 * nalt - 1]
        g = rng.integers(0, nalt + 1, S).astype(str)
        g[rng.random(S) < 0.05] = "."
        fmt = "GT:DP" if v % 3 else "GT"
        cells = [x + ":12" if v % 3 else x for x in g]
        info = ("DP=5;TYPE=%s" % kinds[v % 4]) if v % 7 else ("TYPE=%s;AF=0.5" % kinds[v % 4])
        lines.append("\t".join(["chr1", str(100 + v), ".", "G", alt, "50", "PASS", info, fmt] + cells))
    lines.extend(extra)
    return nl.join(lines) + nl


@pytest.mark.parametrize("crlf", [False, True])
def test_vcf2scoary_native_loop_equals_python_loop(tmp_path, crlf):
    """scoary_vcf_convert (native record loop) writes exactly what the Python loop
    (the line-by-line mirror of the reference) writes: bi- and multi-allelic
    sites, empty ALT alleles, missing genotypes, GT with and without sub-fields,
    --types filters, \\r\\n files."""
    from scoary_amd import io_native, vcf2scoary as v
    rng = np.random.default_rng(4)
    vcf = tmp_path / "in.vcf"
    vcf.write_text(_vcf_text(rng, 300, 37, crlf), newline="")
    assert v._records_offset(str(vcf)) is not None and io_native.available()
    for types in ("ALL", ["snp"], ["ins", "mnp"], ["nothing"]):
        py = io.StringIO()
        with open(vcf, "r", newline=None) as f:
            n_py = v.convert(f, py, types, log=lambda *a: None)
        out = tmp_path / "native.csv"
        n_nat = v.convert_file(str(vcf), str(out), types, log=lambda *a: None)
        assert n_nat == n_py and out.read_text() == py.getvalue(), types
    assert n_py == 0


@pytest.mark.parametrize("bad", ['chr1\t9\t.\tG\tA\t50\tPASS\tTYPE=snp\tGT\t"0"\t1',   # quoted cell
                                 "chr1\t9\t.\tG\tA,T\t50\tPASS\tTYPE=snp\tGT\t+1\t2",  # int() accepts "+1"
                                 "chr1\t9\t.\tG\tA\t50\tPASS\tTYPE=snp\tGT"])           # no sample cells
def test_vcf2scoary_native_loop_hands_odd_files_to_python(tmp_path, bad):
    from scoary_amd import io_native, vcf2scoary as v
    rng = np.random.default_rng(5)
    vcf = tmp_path / "in.vcf"
    vcf.write_text(_vcf_text(rng, 12, 2, extra=[bad]), newline="")
    off = v._records_offset(str(vcf))
    probe = tmp_path / "probe.csv"
    probe.write_text("")
    assert io_native.vcf_convert(str(vcf), off, str(probe)) == -2
    py = io.StringIO()
    with open(vcf, "r", newline=None) as f:
        n_py = v.convert(f, py, "ALL", log=lambda *a: None)
    out = tmp_path / "out.csv"
    assert v.convert_file(str(vcf), str(out), "ALL", log=lambda *a: None) == n_py > 12
    assert out.read_text() == py.getvalue()


# ------------------------------------------------------- native GPA reader ---
def _py_cells(text, delimiter=","):
    import csv
    return list(csv.reader(io.StringIO(text, newline=None) if False else io.StringIO(text),
                           skipinitialspace=True, delimiter=delimiter))


TRICKY = [
    'Gene,x,y,S1,S2,S3\ng1,a,b,1,0,\ng2,"q,1","say ""hi""",, - ,2\n',
    'Gene,x,y,S1,S2\r\ng1,a,b,grp_1,\r\n"g 2", a ,b,"0","-"\r\n',
    'Gene,x,y,S1,S2\ng1,"multi\nline",b,1,1\ng2,a,"b"tail,0,1',          # no trailing newline
    'Gene;x;y;S1;S2\ng1;a;b;1;0\ng2;;;;\n',
    'Gene,x,y,S1,S2\ng1,a,b,   1,   0\ng2,a,b,"",  "-"\n',
    'Gene,x,y,S1,S2\ng1,a,b,00,--\ng2,a,b,0 ,-x\n',
]


@pytest.mark.parametrize("k", range(len(TRICKY)))
def test_native_reader_tokenises_like_python_csv(tmp_path, k):
    from scoary_amd import io_native
    if not io_native.available():
        import __graft_entry__
        __graft_entry__.build()
    text = TRICKY[k]
    delim = ";" if k == 3 else ","
    path = tmp_path / "g.csv"
    path.write_bytes(text.encode())
    with open(path, "r", newline=None) as f:                 # universal newlines like "rU"
        import csv
        rows = list(csv.reader(f, skipinitialspace=True, delimiter=delim))
    header, meta, bits, kept = io_native.read_gpa(str(path), delim, 3)
    assert header == rows[0]
    assert meta == [r[:3] for r in rows[1:]]
    want = np.array([[c not in ("", "0", "-") for c in r[3:]] for r in rows[1:]], dtype=np.uint8)
    got = np.unpackbits(bits.view(np.uint8), axis=1, bitorder="little")[:, :want.shape[1]]
    assert np.array_equal(got, want)
    assert kept == rows[0][3:]


@pytest.mark.parametrize("threads,min_chunk", [(4, 1), (7, 50), (3, 1000), (64, 1)])
def test_native_reader_parallel_body_parse(tmp_path, threads, min_chunk):
    """scoary_gpa_parse_mt cuts the body at line ends and parses the ranges in
    parallel; cuts that land inside quoted multi-line cells are detected and the
    body is then parsed in one piece.  Same result as one thread, every time."""
    from conftest import golden_text
    from scoary_amd import io_native
    rng = np.random.default_rng(8)
    texts = [golden_text("synth2/gpa.csv.gz"),
             golden_text("exampledata/Gene_presence_absence.csv.gz")[:200000].rsplit("\n", 1)[0] + "\n"]
    # a table whose quoted cells contain line breaks (every few rows), CRLF line ends
    rows = ['"Gene","b","c",' + ",".join('"s%d"' % i for i in range(9))]
    for r in range(120):
        cells = ["x" if rng.random() < 0.5 else "" for _ in range(9)]
        note = '"multi\nline\r\ncell %d"' % r if r % 3 == 0 else '"plain %d"' % r
        rows.append('"g%d",%s,"c",%s' % (r, note, ",".join(cells)))
    texts.append("\r\n".join(rows) + "\r\n")
    for k, text in enumerate(texts):
        path = tmp_path / ("t%d.csv" % k)
        path.write_text(text, newline="")
        startcol = 3 if k == 2 else 14
        one = io_native.read_gpa(str(path), ",", startcol, threads=1)
        many = io_native.read_gpa(str(path), ",", startcol, threads=threads, min_chunk=min_chunk)
        assert one[0] == many[0] and one[1] == many[1] and one[3] == many[3]
        assert np.array_equal(one[2], many[2]) and one[2].shape[0] > 100
    # a short row is reported with the same row number whatever the cut positions
    short = list(rows)
    short[78] = '"g77","c"'                                   # record 79 of the file: two cells only
    path = tmp_path / "bad.csv"
    path.write_text("\r\n".join(short) + "\r\n", newline="")
    msgs = []
    for th in (1, threads):
        with pytest.raises(io_native.GpaError) as e:
            io_native.read_gpa(str(path), ",", 3, threads=th, min_chunk=min_chunk)
        msgs.append(str(e.value))
    assert msgs[0] == msgs[1] and msgs[0].startswith("row 79 has 2 cells")


def test_native_reader_errors_and_restriction(tmp_path):
    from scoary_amd import io_native
    p = tmp_path / "g.csv"
    p.write_text("Gene,x,y,S1,S2,S3\ng1,a,b,1,0,1\ng2,a,b,1\n")
    with pytest.raises(io_native.GpaError):
        io_native.read_gpa(str(p), ",", 3)                   # short row
    p.write_text("Gene,x,y,S1,S2,S3\ng1,a,b,1,0,1\n\ng2,a,b,1,1,1\n")
    with pytest.raises(io_native.GpaError):
        io_native.read_gpa(str(p), ",", 3)                   # blank line = empty record
    with pytest.raises(io_native.GpaError):
        io_native.read_gpa(str(p), ",", 6)                   # startcol beyond header
    p.write_text("Gene,x,y,S1,S2,S3\ng1,a,b,1,0,1\ng2,a,b,0,1,1\n")
    header, meta, bits, kept = io_native.read_gpa(str(p), ",", 3, allowed={"S3", "S1"})
    assert kept == ["S1", "S3"] and bits[:, 0].tolist() == [3, 2]


def test_native_and_python_readers_build_the_same_table(exampledir, monkeypatch):
    from scoary_amd import methods as m
    path = os.path.join(exampledir, "Gene_presence_absence.csv")
    with open(path) as f:
        a = m.Csv_to_dic_Roary(f, ",", [3, 5], startcol=14)
    monkeypatch.setenv("SCOARY_PY_CSV", "1")
    with open(path) as f:
        b = m.Csv_to_dic_Roary(f, ",", [3, 5], startcol=14)
    ta, tb = a["Roarydic"], b["Roarydic"]
    assert ta.ids == tb.ids and ta.nugn == tb.nugn and ta.annotation == tb.annotation
    assert np.array_equal(ta.rows64, tb.rows64) and ta.extra == tb.extra
    assert a["Extracols"] == b["Extracols"] == ["No. isolates", "Avg sequences per isolate"]
    assert list(a["Zero_ones_matrix"]) == list(b["Zero_ones_matrix"])
    # duplicate identifiers (non-Roary files): later row wins, earlier position kept
    vcf = os.path.join(exampledir, "mutations_presence_absence.csv")
    monkeypatch.delenv("SCOARY_PY_CSV")
    with open(vcf) as f:
        c = m.Csv_to_dic_Roary(f, ",", [], startcol=10)
    monkeypatch.setenv("SCOARY_PY_CSV", "1")
    with open(vcf) as f:
        d = m.Csv_to_dic_Roary(f, ",", [], startcol=10)
    assert c["Roarydic"].ids == d["Roarydic"].ids
    assert np.array_equal(c["Roarydic"].rows64, d["Roarydic"].rows64)


def test_readers_on_synth2_quoted_cells(tmp_path, monkeypatch):
    """Second data set: quoted headers, quoted cells with commas, `""`, "0" and "-"
    absent spellings -- native and Python readers agree with the test suite's own
    minimal reader."""
    from conftest import golden_text, read_dense
    from scoary_amd import methods as m
    from scoary_amd.engine import pack_bits_rows
    path = tmp_path / "gpa.csv"
    path.write_text(golden_text("synth2/gpa.csv.gz"))
    ids, strains, genes, _names, _traits = read_dense(golden_text("synth2/gpa.csv.gz"),
                                                      golden_text("synth2/traits.csv.gz"))
    tables = []
    for py in (False, True):
        if py:
            monkeypatch.setenv("SCOARY_PY_CSV", "1")
        with open(path) as f:
            tables.append(m.Csv_to_dic_Roary(f, ",", [], startcol=14))
    for gd in tables:
        t = gd["Roarydic"]
        assert t.ids == ids and gd["Strains"] == strains
        assert np.array_equal(t.rows64, pack_bits_rows(genes))
        assert t.annotation[7] == "hypothetical, protein 7" and t.nugn[3] == "" and t.nugn[4] == "nm4"
    with open(tmp_path / "traits.csv", "w") as f:
        f.write(golden_text("synth2/traits.csv.gz"))
    with open(tmp_path / "traits.csv") as f:
        td, prune = m.Csv_to_dic(f, ",", None, strains)
    assert sorted(x for x in prune["lineage"] if x) == ["iso_04", "iso_09", "iso_17", "iso_33", "iso_40"]
    assert [x for x in prune["driven"] if x] == ["iso_17"]


def test_io_library_exports_every_declared_symbol():
    from scoary_amd import io_native
    with open(os.path.join(ROOT, "include", "scoary_io.h")) as f:
        src = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(scoary_(?:gpa|lists|vcf|upgma|results|format)_[a-z_]+)\s*\(", src)))
    lib = ctypes.CDLL(io_native.LIB_PATH)
    assert len(names) == 21 and "scoary_vcf_convert" in names and "scoary_upgma_merges" in names
    assert "scoary_results_write" in names and "scoary_format_float_repr" in names
    for n in names:
        assert hasattr(lib, n), n


# ------------------------------------------------- minority index lists ------
@pytest.mark.parametrize("gpw,classes,stride,piece", [(4, 2, 64, 0), (8, 4, 32, 0), (16, 8, 16, 0),
                                                       (16, 4, 64, 16), (32, 8, 32, 8),
                                                       (64, 16, 16, 4), (64, 32, 8, 4), (64, 64, 4, 4)])
def test_minority_lists_builder(gpw, classes, stride, piece):
    """scoary_lists_build (host native): per gene the positions of its minority
    value, padded with N to a multiple of 16 (half a kernel step; ngroups counts 16s, start of
    the interleaved layout 32s) and to the wave group's longest list,
    genes ordered by descending length, entries in the bank-rotation order of spec S6
    (entry e of slot k from residue class (k + e) mod classes while every class has
    positions left), entries premultiplied by the row stride;
    piece > 0: the group's lists interleaved in pieces, last group stored in full."""
    from scoary_amd import io_native
    from scoary_amd.engine import pack_bits_rows
    rng = np.random.default_rng(9)
    G, N = 203, 333
    dense = (rng.random((G, N)) < rng.uniform(0.0, 1.0, (G, 1))).astype(np.uint8)
    dense[0] = 0
    dense[1] = 1
    L = io_native.build_lists(pack_bits_rows(dense), N, stride, gpw, classes, piece)
    order, start, ng, flipped = L["order"], L["start"], L["ngroups"], L["flipped"]
    assert sorted(order.tolist()) == list(range(G))
    n1 = dense.sum(1).astype(np.int64)
    length = np.where(n1 * 2 <= N, n1, N - n1)
    assert np.array_equal(flipped, (n1 * 2 > N).astype(np.uint8))
    assert np.all(np.diff(length[order]) <= 0)                      # descending
    total = 0
    for k in range(G):
        g = order[k]
        if piece == 0:
            ent = L["idx"][start[k] * 16:(start[k] + ng[k]) * 16]    # contiguous mode: start in 16s
            assert start[k] * 16 == total
            total += ng[k] * 16
        else:
            j, base = k % gpw, start[k] * 32
            assert start[k] == start[(k // gpw) * gpw]
            if j == 0:
                assert base == total
                total += gpw * ng[k] * 16                           # full groups, also the last
            e = np.arange(ng[k] * 16)
            ent = L["idx"][base + ((e // piece) * gpw + j) * piece + e % piece]
        assert ng[k] == ng[(k // gpw) * gpw]                        # equal within a wave group
        assert ng[k] * 16 >= length[g] and \
            (ng[(k // gpw) * gpw] * 16 - length[order[(k // gpw) * gpw]]) < 16
        assert np.all(ent % stride == 0)
        pos = (ent // stride).astype(np.int64)
        real = pos[:length[g]]
        assert np.all(pos[length[g]:] == N)                         # padding -> zero row
        want = np.nonzero(dense[g] == (0 if flipped[g] else 1))[0]
        assert sorted(real.tolist()) == want.tolist()
        cls = real % classes                                        # round-robin over the classes
        full = classes * min(np.bincount(cls, minlength=classes)) if len(real) else 0
        assert np.array_equal(cls[:full], (k + np.arange(full)) % classes)
        # spec S6 in full: position (class, rank) sits on grid slot rank * classes + ((class - k)
        # mod classes) if that is below the list length; the rest fill the holes, both in order
        wc = want % classes
        rho = np.zeros(len(want), dtype=np.int64)
        for c in range(classes):
            rho[wc == c] = np.arange(int((wc == c).sum()))
        slot = rho * classes + ((wc - k) % classes)
        assert len(set(slot.tolist())) == len(slot)
        expect = np.full(len(want), -1, dtype=np.int64)
        inside = slot < len(want)
        expect[slot[inside]] = want[inside]
        over = want[~inside][np.argsort(slot[~inside])]
        expect[np.nonzero(expect < 0)[0]] = over
        assert np.array_equal(real, expect)
        aligned = (real % classes) == ((k + np.arange(len(real))) % classes)
        assert aligned.sum() >= len(real) - len(over)                # only hole entries are misaligned
    if piece:                                                       # missing genes of the last group
        k0 = (G - 1) // gpw * gpw
        e = np.arange(ng[k0] * 16)
        for j in range(G - k0, gpw):
            ent = L["idx"][start[k0] * 32 + ((e // piece) * gpw + j) * piece + e % piece]
            assert np.all(ent == N * stride)
    assert L["entries"] == total and len(L["idx"]) >= total + 32


# ------------------------------------------------------------ UPGMA, host ----
def test_native_upgma_equals_numpy_loop_and_reference_goldens():
    """scoary_upgma_merges (host library) == the numpy quad-tree loop == the trees the
    reference built (tests/golden/upgma_cases.json, random matrices with many exact ties),
    plus tie-heavy larger cases (duplicated strains) native vs numpy."""
    from scoary_amd import io_native, tree as T
    assert io_native.available()
    with open(os.path.join(GOLDEN, "upgma_cases.json")) as f:
        cases = json.load(f)["cases"]

    def counts_of(X):                                   # strains x variable genes, 0/1
        Xi = X.astype(np.int64)
        return Xi @ (1 - Xi).T + (1 - Xi) @ Xi.T

    for c in cases:
        X = np.array(c["matrix"], dtype=np.uint8)
        tot = X.sum(axis=0)
        var = X[:, (tot > 0) & (tot < X.shape[0])]
        cnt = counts_of(var)
        for native in (True, False):
            assert str(T.upgma_from_counts(cnt, var.shape[1], c["names"], native=native)) == c["newick"]
    for seed in range(40):
        rng = np.random.default_rng(seed)
        n = int(rng.choice([2, 3, 5, 8, 9, 17, 33, 64, 65, 130]))
        X = (rng.random((n, int(rng.choice([3, 8, 40])))) < 0.3).astype(np.uint8)
        if seed % 2:
            X[n // 2:] = X[:n - n // 2]                 # duplicated strains: distance-0 and other ties
        cnt, names = counts_of(X), ["s%d" % i for i in range(n)]
        assert T.upgma_from_counts(cnt, X.shape[1], names, native=True) == \
            T.upgma_from_counts(cnt, X.shape[1], names, native=False)


# ------------------------------------------------------------- custom trees ----
def test_read_newick_resolves_polytomies_like_ete3(tmp_path):
    """-n trees with more than two children per node: the reference calls ete3's
    resolve_polytomy(recursive=True) (scoary/nwkhandler.py:19), which keeps the first child
    at the top and nests the rest as a chain of new FIRST children, pairing the last two:
    (a,b,c) -> [[b, c], a]; also below the root, where the rooted topology matters.  Branch
    lengths, support values, internal labels and quotes are dropped."""
    from scoary_amd import tree as T
    cases = {
        "(a,b,c);": [["b", "c"], "a"],
        "(a,b,c,d);": [[["c", "d"], "b"], "a"],
        "((a,b,c)X:0.1,d);": [[["b", "c"], "a"], "d"],
        "((a:1,'b b':2,c:3,d)90:0.5,(e,f),g);": [[["e", "f"], "g"], [[["c", "d"], "b b"], "a"]],
        "(a,(b,c));": ["a", ["b", "c"]],
    }
    for text, want in cases.items():
        p = tmp_path / "t.nwk"
        p.write_text(text + "\n")
        got, members = T.read_newick(str(p))
        assert got == want, text
        assert sorted(members) == sorted(T.tips_of(got))


# ------------------------------------------- build-time kernel checks, counters ----
_REMARKS = """\
x.hip:1:1: remark: Function Name: _ZN1A15k_permute_listsILi4ELi4ELi11EEEvPKj [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     TotalSGPRs: 66 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     VGPRs: %(vgprs)d [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     ScratchSize [bytes/lane]: %(scratch)d [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     Occupancy [waves/SIMD]: 4 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     SGPRs Spill: 0 [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     VGPRs Spill: %(spill)d [-Rpass-analysis=kernel-resource-usage]
x.hip:1:1: remark:     LDS Size [bytes/block]: %(lds)d [-Rpass-analysis=kernel-resource-usage]
"""


def test_build_refuses_a_list_kernel_that_spills_or_grew_static_lds():
    """the register-pinned kernel must fail the BUILD, not the GPU run, when
    the compiler did not produce what the source assumes (no scratch, <= 128 VGPRs, the label
    tile as the only LDS object).  The parser reads hipcc's own resource-usage remarks."""
    import __graft_entry__ as ge

    def report(**kw):
        vals = dict(vgprs=127, scratch=0, spill=0, lds=0)
        vals.update(kw)
        one = _REMARKS % vals
        seg = (_REMARKS % dict(vals, **(seg_kw or {}))).replace("15k_permute_listsILi4ELi4ELi11E",
                                                              "18k_permute_seglistsILi16E")
        return ge.parse_resource_usage("".join(one.replace("Li11E", "Li%dE" % (11 + i)) for i in range(4)) + seg)

    seg_kw = None
    ok = report()
    assert len(ok) == 5 and all(v["VGPRs"] == 127 and v["ScratchSize"] == 0 for v in ok.values())
    ge.check_kernel_resources(ok)
    for bad in (dict(scratch=16), dict(spill=3), dict(vgprs=130), dict(lds=64)):
        with pytest.raises(RuntimeError):
            ge.check_kernel_resources(report(**bad))
    for seg_kw in (dict(scratch=8), dict(vgprs=129), dict(lds=16)):      # the segmented kernel (N > 40959) alike
        with pytest.raises(RuntimeError):
            ge.check_kernel_resources(report())
    seg_kw = None
    with pytest.raises(RuntimeError):                     # an instance went missing
        ge.check_kernel_resources({k: v for k, v in list(ok.items())[:3] + list(ok.items())[4:]})
    with pytest.raises(RuntimeError):                     # the segmented kernel went missing
        ge.check_kernel_resources({k: v for k, v in ok.items() if "seglists" not in k})
    ge.check_ctr_banks()                                  # the committed register table is consistent
    # the library that is actually in the tree passed the same check when it was built
    res_path = ge.HIP_RESOURCES
    if os.path.exists(res_path):
        with open(res_path) as f:
            ge.check_kernel_resources(json.load(f))


def test_bench_refuses_counters_of_other_kernel_sources(tmp_path, monkeypatch):
    """roofline counters come from a profiles/ file that records the hash
    of the kernel sources it was collected from; a summary of other sources is refused
    (null + reason), never reported."""
    import bench
    sha = bench.kernel_source_sha()
    assert len(sha) == 64 and sha == bench.kernel_source_sha()
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    monkeypatch.setattr(bench, "kernel_source_sha", lambda: sha)
    body = {"k_permute_lists<4, 4, 11>": {"SQ_INSTS_VALU": 3.0e9, "hbm_traffic_bytes_per_launch": 2.4e9},
            "k_permute_reg<16, 1>": {"SQ_INSTS_VALU": 1.0e10}}
    (prof / "r09_pmc.json").write_text(json.dumps(dict(body, _meta={"workload": "cfg3", "kernel_source_sha256": "0" * 64})))
    ctr, why = bench.load_counters("cfg3", True, "k_permute_lists")
    assert ctr is None and "sha256 mismatch" in why
    (prof / "r10_pmc.json").write_text(json.dumps(dict(body, _meta={"workload": "cfg3", "kernel_source_sha256": sha})))
    ctr, why = bench.load_counters("cfg3", True, "k_permute_lists")
    assert why is None and ctr["SQ_INSTS_VALU"] == 3.0e9 and ctr["source"].endswith("r10_pmc.json")
    ctr, _ = bench.load_counters("cfg3", True, "k_permute")            # the dense kernel's entry, not the list one
    assert ctr["SQ_INSTS_VALU"] == 1.0e10
    assert bench.load_counters("cfg4", True, "k_permute_lists")[0] is None        # other workload
    ctr, why = bench.load_counters("cfg3", False, "k_permute_lists")              # shape overridden
    assert ctr is None and "overridden" in why


def test_kept_ranks_are_the_union_of_the_workers_prefixes():
    """the reference's worker k of n stops at its own first failing rank
    (scoary/methods.py:1076-1078, :1290-1294); the weave's modulo is of the rank in the
    sorted results.  Monotone cutoff columns: a plain prefix, whatever n.  A non-monotone
    column (stale-rank BH under --collapse): worker 1 keeps rank 3 although rank 2 failed."""
    from scoary_amd import methods as M
    order = np.array([4, 2, 0, 3, 1])
    mono = {"p_v": np.array([.03, .5, .02, .04, .01]), "B_p": np.ones(5), "BH_p": np.ones(5)}
    for n in (None, 1, 2, 3, 7):
        assert M._kept_ranks(order, mono, {"I": 0.035}, n).tolist() == [0, 1, 2]
    bh = np.ones(5)
    bh[order] = [.01, .01, .9, .02, .9]           # by rank: pass pass FAIL pass FAIL
    cols = {"p_v": mono["p_v"], "B_p": np.ones(5), "BH_p": bh}
    assert M._kept_ranks(order, cols, {"BH": 0.05}, 1).tolist() == [0, 1]
    assert M._kept_ranks(order, cols, {"BH": 0.05}, 2).tolist() == [0, 1, 3]   # worker 1: ranks 1, 3
    assert M._kept_ranks(order, cols, {"BH": 0.05}, 3).tolist() == [0, 1, 3]   # worker 0: ranks 0, 3
    assert M._kept_ranks(order, cols, {"P": 0.05}, 2).tolist() == [0, 1, 2, 3, 4]   # no I/B/BH cutoff


def test_zero_ones_matrix_of_a_table_without_data_rows():
    """a header-only (or fully filtered) gene table has a (0, W) bit matrix; the
    lazy Zero_ones_matrix must be empty, not a numpy reshape error."""
    from scoary_amd import methods as M
    z = M._LazyZeroOnes(np.zeros((0, 3), dtype=np.uint64), 130)
    assert z.file_rows().shape == (0, 130)
    assert len(z) == 0 and list(z) == []


def test_bench_telemetry_summarises_the_samples_of_the_timed_region():
    """bench.py samples shader clock and socket power from a side thread during
    the timed region.  With an injected reader: only samples inside [t_begin, t_end] count, missing
    values are skipped, and a box without any source yields nulls -- never an exception."""
    import time
    import bench
    seq = iter([(2100.0, 1300.0), (2200.0, None), (None, 1350.0)] + [(2150.0, 1320.0)] * 10000)
    tel = bench.Telemetry(0, period_s=0.001, reader=lambda: next(seq)).start()
    t0 = time.perf_counter()
    time.sleep(0.05)
    t1 = time.perf_counter()
    out = tel.stop(t0, t1)
    assert out["source"] == "injected" and out["samples"] >= 10
    assert 2100.0 <= out["sclk_mhz_min"] <= out["sclk_mhz_mean"] <= out["sclk_mhz_max"] <= 2200.0
    assert 1300.0 <= out["socket_power_w_mean"] <= 1350.0
    assert bench.Telemetry(0, reader=lambda: (1.0, 1.0)).start().stop(t1 + 100, t1 + 200)["samples"] == 0

    def broken():
        raise OSError("sensor gone")
    out = bench.Telemetry(0, period_s=0.001, reader=broken).start()
    time.sleep(0.01)
    out = out.stop()
    assert out["sclk_mhz_mean"] is None and out["socket_power_w_mean"] is None
    none = bench.Telemetry(0)                                # this container: no amdgpu device at all
    if none.source is None:
        assert none.start().stop()["samples"] == 0


def test_roofline_ops_per_clock_follows_the_run_s_clock(monkeypatch):
    """ops_per_clock = SQ_INSTS_VALU x 64 / (kernel time x sclk_mhz_mean): the same counters and
    kernel time at two clocks give the same frac (taken against 2.4 GHz) and different ops per clock."""
    import types
    import bench
    monkeypatch.setattr(bench, "load_counters", lambda *a: ({"SQ_INSTS_VALU": 3.0e9, "source": "x"}, None))
    monkeypatch.setattr(bench, "measured_copy_peak", lambda dev: 5000.0)
    args = types.SimpleNamespace(config="cfg3", genes=None, permutations=None, isolates=None, traits=None)
    eng = types.SimpleNamespace(device="cpu")
    a = bench.roofline_report(args, eng, True, 50000, 2000, 10, 10000, 10240, "k_permute_lists", 4.8,
                              telemetry={"sclk_mhz_mean": 2400.0, "socket_power_w_mean": 1000.0, "samples": 12})
    b = bench.roofline_report(args, eng, True, 50000, 2000, 10, 10000, 10240, "k_permute_lists", 4.8,
                              telemetry={"sclk_mhz_mean": 2000.0, "socket_power_w_mean": 1350.0, "samples": 12})
    assert a["frac"] == b["frac"] and abs(a["ops_per_clock_frac"] - a["frac"]) < 1e-12
    assert abs(b["ops_per_clock"] / a["ops_per_clock"] - 1.2) < 1e-12
    assert b["sclk_mhz_mean"] == 2000.0 and b["socket_power_w_mean"] == 1350.0 and b["telemetry_samples"] == 12
    c = bench.roofline_report(args, eng, True, 50000, 2000, 10, 10000, 10240, "k_permute_lists", 4.8)
    assert c["ops_per_clock"] is None and c["sclk_mhz_mean"] is None


def test_usable_cpus_honours_affinity_and_cgroup_quota(monkeypatch):
    """The CPU baselines start as many workers as the process may really use (a
    Pool(256) on a container granted 16 CPUs reported 55 tests/s per worker): os.cpu_count(), the
    affinity mask and the cgroup v2 / v1 quota, whichever is smallest."""
    import builtins
    import io
    from oracle import scipy_baseline as sb
    monkeypatch.setattr(sb.os, "cpu_count", lambda: 256)
    monkeypatch.setattr(sb.os, "sched_getaffinity", lambda pid: set(range(64)), raising=False)
    real_open = builtins.open

    def fake(files):
        def _open(path, *a, **k):
            if path in files:
                if files[path] is None:
                    raise OSError("no such file")
                return io.StringIO(files[path])
            if str(path).startswith("/sys/fs/cgroup"):
                raise OSError("no such file")
            return real_open(path, *a, **k)
        return _open
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": "1600000 100000\n"}))
    assert sb.usable_cpus() == 16                              # cgroup v2: quota / period
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": "max 100000\n"}))
    assert sb.usable_cpus() == 64                              # no quota: the affinity mask
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": None,
                                                "/sys/fs/cgroup/cpu/cpu.cfs_quota_us": "800000\n",
                                                "/sys/fs/cgroup/cpu/cpu.cfs_period_us": "100000\n"}))
    assert sb.usable_cpus() == 8                               # cgroup v1
    monkeypatch.setattr(builtins, "open", fake({}))
    assert sb.usable_cpus() == 64
    monkeypatch.setattr(builtins, "open", fake({"/sys/fs/cgroup/cpu.max": "50000 100000\n"}))
    assert sb.usable_cpus() == 1                               # half a CPU still runs one worker


# ------------------------------------------------- binomial test of the pairwise stage ---
def test_binomial_p_is_scipys_double():
    """The two binomial p columns of the pairwise stage (ss.binom_test(x, n, 0.5), scoary/methods.py:1267-1275):
    tree._boost_binom_tail_half restates Boost's finite sum behind scipy.stats.binom.cdf / .sf -- equal to SciPy
    bit for bit wherever it applies (every n <= 78, the tails beyond) --, and tree.binom_two_sided is
    scipy.stats.binomtest(x, n, 0.5).pvalue to the last bit for every n <= 85 (tables of at most 170 isolates, whose
    result files are held to the reference's bytes), within 1e-12 beyond (the exact dyadic tail there; SciPy's own
    value is 1e-13 off it in the far tail of n = 1500)."""
    import scipy.stats as ss
    from scoary_amd import tree as T
    covered = 0
    for n in list(range(1, 131)) + [170, 233, 300]:
        ks = np.arange(-1, n + 2)
        cdf, sf = ss.binom.cdf(ks, n, 0.5), ss.binom.sf(ks, n, 0.5)
        for j, k in enumerate(ks):
            for upper, want in ((False, cdf[j]), (True, sf[j])):
                got = T._boost_binom_tail_half(int(k), n, upper)
                if got is not None:
                    covered += 1
                    assert got == want, (k, n, upper, repr(got), repr(float(want)))
        if n <= 78:
            assert all(T._boost_binom_tail_half(k, n, u) is not None for k in range(n + 1) for u in (False, True))
    assert covered > 15000
    assert T.binom_two_sided(5, 15) == 0.30175781249999994 and T.binom_two_sided(25, 25) == 5.960464477539063e-08
    for n in range(1, 86):
        for x in range(n + 1):
            assert T.binom_two_sided(x, n) == ss.binomtest(x, n, 0.5).pvalue, (x, n)
    for n in (86, 99, 128, 257, 1000, 1500):
        for x in range(0, n + 1, 1 if n < 130 else 37):
            want = ss.binomtest(x, n, 0.5).pvalue
            assert abs(T.binom_two_sided(x, n) - want) <= 1e-12 * want, (x, n)


def test_abort_thresholds_without_scipy_equal_scipys():
    """The early-abort thresholds of the sequential permutation estimator (smallest r with 1 - binom.cdf(r, i, 0.1) <
    0.05, scoary/methods.py:1360-1361): the SciPy-free evaluation (tree._abort_thresholds_direct) gives SciPy's answer
    for every i in [30, 6000) and calls nothing undecided; the cached table is what empirical_p_sequential reads."""
    from scoary_amd import tree as T
    i = np.arange(30, 6000)
    thr, undecided = T._abort_thresholds_direct(i)
    assert not undecided.any()
    assert np.array_equal(thr, T._abort_thresholds_scipy(i))
    T._ABORT_CACHE.clear()
    table = T._abort_thresholds(700)
    assert np.array_equal(table[30:], thr[:670]) and (table[:30] == np.iinfo(np.int64).max).all()
    x = np.arange(2000, dtype=np.float64)
    import math
    assert np.allclose(T._lgamma(x + 1.0), [math.lgamma(v + 1.0) for v in x], rtol=1e-14, atol=1e-13)
