import csv
import gzip
import io
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden_text(rel):
    """Text of a (possibly .gz) fixture under tests/golden/."""
    path = os.path.join(GOLDEN, rel)
    if path.endswith(".gz"):
        with gzip.open(path, "rt", newline="") as f:
            return f.read()
    with open(path, newline="") as f:
        return f.read()


@pytest.fixture(scope="session")
def exampledir(tmp_path_factory):
    """The reference's exampledata, unpacked into a temp dir."""
    d = tmp_path_factory.mktemp("exampledata")
    src = os.path.join(GOLDEN, "exampledata")
    for fn in os.listdir(src):
        with open(os.path.join(d, fn[:-3]), "w", newline="") as out:
            out.write(golden_text(os.path.join("exampledata", fn)))
    return str(d)


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLDEN, "manifest.json")) as f:
        return json.load(f)


def read_dense(gpa_text, traits_text, startcol=14, allowed=None):
    """Independent, minimal reader used only to feed the oracle in tests:
    returns (gene_ids, strains, genes[G,N] uint8, trait_names, traits[T,N] uint8
    with 2 = missing).  Presence rule methods.py:476-485; missing-value rule
    methods.py:576-598, 602-610."""
    rows = list(csv.reader(io.StringIO(gpa_text), skipinitialspace=True))
    header = rows[0]
    cols = [c for c in range(startcol, len(header))
            if allowed is None or header[c] in allowed]
    strains = [header[c] for c in cols]
    roary = header[:3] == ["Gene", "Non-unique Gene name", "Annotation"]
    ids, dense = {}, []
    for q in rows[1:]:
        ident = q[0] if roary else "_|_".join(q[:3])
        line = [0 if q[c] in ("", "0", "-") else 1 for c in cols]
        if ident in ids:                       # duplicate ids overwrite
            dense[ids[ident]] = line
        else:
            ids[ident] = len(dense)
            dense.append(line)
    genes = np.array(dense, dtype=np.uint8).reshape(len(dense), len(cols))
    trows = list(csv.reader(io.StringIO(traits_text)))
    names = trows[0][1:]
    tmap = {r[0]: r[1:] for r in trows[1:] if r}
    traits = np.full((len(names), len(strains)), 2, dtype=np.uint8)
    for j, s in enumerate(strains):
        if s in tmap and (allowed is None or s in allowed):
            for t, v in enumerate(tmap[s]):
                if v in ("0", "1"):
                    traits[t, j] = int(v)
    return list(ids.keys()), strains, genes, names, traits
