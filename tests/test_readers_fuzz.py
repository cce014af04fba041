"""The two input readers against the REAL reference's on the 200 tables of the differential corpus, each written in
five ways (as it is, Windows line ends, byte-order mark, blank last line, no final newline): the gene table
(identifiers in dictionary order, every gene's row, strains, extra columns, the tree stage's 0/1 matrix) and the
traits (values per trait, isolates to prune) must come out as the reference's do -- compared through digests written
by tests/golden/make_fuzz.py -- and an input the reference refuses must be refused with the same message.  Host code:
runs without a GPU."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def test_readers_equal_the_references_on_the_corpus(monkeypatch):
    # make_fuzz imports the reference when it is there (build container); only its helpers are used here
    monkeypatch.setitem(sys.modules, "scoary", type(sys)("scoary"))
    stub = type(sys)("scoary.methods")
    from scoary_amd import methods as ours
    stub.grabcoltype = ours.grabcoltype
    monkeypatch.setitem(sys.modules, "scoary.methods", stub)
    sys.modules["scoary"].methods = stub
    sys.modules.pop("make_fuzz", None)
    import make_fuzz as mf
    from test_gpu_fuzz import CORPUS
    differing, compared = [], 0
    for case in CORPUS["cases"]:
        got = mf.reader_records(case, ours)
        for v in mf.READER_VARIANTS:
            for part in ("genes", "traits"):
                want = case["readers"][v][part]
                if want is None:             # the reference crashed (or never got that far): nothing to match
                    continue
                compared += 1
                if got[v][part] != want:
                    differing.append((case["id"], v, part, got[v][part], want))
    assert not differing, differing[:5]
    assert compared > 1500
