"""scoary_fisher_scipy on a GPU box: 20 000 random tables up to 104 723 isolates against the oracle (bitwise) and the
box's SciPy, the tables it must leave alone, and its time on 500 000 tables at N = 2000."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from scoary_amd.engine import AssociationEngine
from oracle import oracle as orc
from scipy.stats import fisher_exact
eng = AssociationEngine(0)
rng = np.random.default_rng(11)
tabs = []
for k in range(20000):
    N = int(rng.choice([171, 200, 997, 2000, 5000, 10007, 40000, 104723])) if k % 3 == 0 else int(rng.integers(2, 12000))
    npos, g = int(rng.integers(1, N)), int(rng.integers(1, N))
    lo, hi = max(0, g + npos - N), min(g, npos)
    mu = g * npos / N; sd = max(1.0, (mu * (1 - npos / N)) ** 0.5)
    a = int(min(max(round(mu + rng.normal() * sd * rng.choice([0.3, 1, 3, 10])), lo), hi))
    tabs.append((a, npos - a, g - a, N - npos - g + a))
tabs.append((60000, 20000, 20000, 20000)); tabs.append((0, 0, 100, 200))
tabs = np.array(tabs, dtype=np.int32)
dt = torch.from_numpy(tabs).to(eng.device)
p0, odds, crit = eng.fisher(dt)
p = p0.clone()
torch.cuda.synchronize(); t0 = time.time()
skipped = eng.fisher_scipy(dt, p)
torch.cuda.synchronize(); t1 = time.time()
got = p.cpu().numpy(); walk = p0.cpu().numpy()
want = orc.fisher_scipy_many(tabs)
N = tabs.sum(1)
inrange = (N >= 171) & (N <= 104723) & (tabs[:, 0] + tabs[:, 1] > 0) & (tabs[:, 2] + tabs[:, 3] > 0) & (tabs[:, 0] + tabs[:, 2] > 0) & (tabs[:, 1] + tabs[:, 3] > 0)
print("skipped", skipped, "kernel %.1f ms" % ((t1 - t0) * 1e3), "in range", int(inrange.sum()))
print("kernel == oracle bitwise on in-range:", np.array_equal(got[inrange].view(np.uint64), want[inrange].view(np.uint64)),
      " mismatches", int((got[inrange] != want[inrange]).sum()))
print("out-of-range untouched:", np.array_equal(got[~inrange].view(np.uint64), walk[~inrange].view(np.uint64)))
print("max rel |walk - scipy|", np.max(np.abs(walk[inrange] - want[inrange]) / np.maximum(want[inrange], 1e-300)))
bad = 0
for k in np.nonzero(inrange)[0][:1500]:
    a, b, c, d = tabs[k].tolist()
    if float(fisher_exact([[a, b], [c, d]])[1]) != got[k]: bad += 1
print("vs the SciPy of this box, 1500 tables: mismatches", bad)
# timing at cfg3-like shape
G = 500000
t = np.empty((G, 4), dtype=np.int32)
npos = 700; Nn = 2000
g = rng.integers(40, 1960, G); a = rng.hypergeometric(npos, Nn - npos, g)
t[:, 0] = a; t[:, 1] = npos - a; t[:, 2] = g - a; t[:, 3] = Nn - npos - g + a
dt = torch.from_numpy(t).to(eng.device); p0, _, _ = eng.fisher(dt); p = p0.clone()
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.time(); eng.fisher_scipy(dt, p); torch.cuda.synchronize(); print("500k tables at N=2000: %.1f ms" % ((time.time() - t0) * 1e3))
