"""INTEGRATION.md section 2 is executable: the raw ctypes stub a reference maintainer would
write (no scoary_amd import) is pulled out of the document and run on a small case; counts,
Fisher p and the exceedance counts of BOTH permutation flows must equal the oracle's, and so must
the counts of the planned form (scoary_trait_plan + scoary_counts_planned, ABI 7)."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _blocks():
    with open(os.path.join(ROOT, "INTEGRATION.md")) as f:
        text = f.read()
    sec2 = text[text.index("## 2. Binding the C-ABI directly"):text.index("### Launch-bound workloads")]
    return re.findall(r"```python\n(.*?)```", sec2, flags=re.S)


def test_integration_md_ctypes_stub_runs_and_matches_the_oracle():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X")
    from oracle import oracle as orc
    blocks = _blocks()
    assert len(blocks) == 3                     # create / counts / Fisher; permutations; the trait plan
    rng = np.random.default_rng(11)
    G, N, T, P, seed = 300, 700, 2, 600, 424
    dense01 = (rng.random((G, N)) < rng.uniform(0.05, 0.95, (G, 1))).astype(np.uint8)
    labels = (rng.random((T, N)) < 0.4).astype(np.uint8)
    labels[1, ::17] = 2
    env = {"dense01": dense01, "labels": labels, "P": P, "seed": seed}
    cwd = os.getcwd()
    os.chdir(ROOT)                              # the stub loads the library by its repo-relative path
    try:
        exec(blocks[0], env)                    # create, pack, counts, Fisher
        tb = orc.pack_rows((labels == 1).astype(np.uint8))
        mb = orc.pack_rows((labels != 2).astype(np.uint8))
        gb = orc.pack_rows(dense01)
        want_c = orc.counts_packed(gb, tb, mb).transpose(1, 0, 2)
        assert np.array_equal(env["d_counts"].cpu().numpy(), want_c)
        _, want_p = orc.fisher_many(np.ascontiguousarray(want_c).reshape(-1, 4))
        assert np.max(np.abs(env["d_p"].cpu().numpy().ravel() - want_p)) < 1e-12
        exec(blocks[1], env)                    # dense flow, then the list-driven flow (adds into d_r)
        torch.cuda.synchronize()
    finally:
        os.chdir(cwd)
    want_r = orc.permute_r(gb, tb, mb, N, P, seed).T
    got = env["d_r"].cpu().numpy().view(np.uint32)
    assert np.array_equal(got, 2 * want_r)      # both flows accumulated the same counts into d_r
    assert np.array_equal(env["d_r2"].cpu().numpy().view(np.uint32), want_r)   # fused regions, overwrite
    # the planned form of the counts (ABI 7): same tables, margins and mask classes from the plan
    env["d_counts"].fill_(-1)
    env["d_margins"].fill_(-1)
    os.chdir(ROOT)
    try:
        exec(blocks[2], env)
        torch.cuda.synchronize()
    finally:
        os.chdir(cwd)
    assert np.array_equal(env["d_counts"].cpu().numpy(), want_c)
    assert np.array_equal(env["d_margins"].cpu().numpy()[:, 0], (labels == 1).sum(1))
    assert np.array_equal(env["d_margins"].cpu().numpy()[:, 1], (labels != 2).sum(1))
    assert env["d_class"].cpu().numpy().tolist() == [0, 1]      # trait 1 has missing isolates: its own class
    env["lib"].scoary_destroy(env["h"])
