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


_GATHER_SCRIPT = r'''
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.environ["SCOARY_ROOT"])
from scoary_amd import _abi
lib = _abi.load()
# the RCCL instance torch ships (the one a torch process already holds), else the ROCm one
cand = [os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1"]
path = next(p for p in cand if os.path.exists(p))
rccl = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]
uid = UniqueId()
assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
comm = ctypes.c_void_p()
rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
torch.cuda.set_device(0)
assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
h = ctypes.c_void_p()
assert lib.scoary_create(0, ctypes.byref(h)) == 0
send = torch.arange(3 * 7 * 10, dtype=torch.int32, device="cuda")          # [T=3][G=7][10] records
recv = torch.full_like(send, -1)
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
rc = lib.scoary_gather(h, ctypes.c_void_p(rccl._handle), comm, ctypes.c_void_p(send.data_ptr()),
                       ctypes.c_void_p(recv.data_ptr()), send.numel() * 4, 0, 1, 0, stream)
assert rc == 0, lib.scoary_last_error(h)
torch.cuda.synchronize()
assert torch.equal(send, recv)
# bad arguments come back as a status, not as a crash
assert lib.scoary_gather(h, None, None, ctypes.c_void_p(send.data_ptr()), None, 4, 0, 1, 0, stream) == -1
rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
rccl.ncclCommDestroy(comm)
print("GATHER_OK", path)
'''


def test_gather_entry_point_through_rccl_one_rank(tmp_path):
    """scoary_gather (SURVEY 8b's C-level exchange entry): a host that binds the C-ABI without
    torch.distributed hands over its own ncclComm_t and the dlopen handle of its RCCL; the records of
    every rank arrive on the root through one group of ncclSend / ncclRecv.  One rank here (the only
    device): RCCL communicator from ctypes, a self send / receive, block compared word for word.
    Run in a child process with a timeout -- a collective that does not complete must fail the test,
    not hang the suite.  Reference analogue: the result weave of scoary/methods.py:1115-1122."""
    import subprocess
    import sys
    from conftest import ROOT
    script = tmp_path / "gather.py"
    script.write_text(_GATHER_SCRIPT)
    env = dict(os.environ, SCOARY_ROOT=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=240, env=env)
    assert out.returncode == 0 and "GATHER_OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])
