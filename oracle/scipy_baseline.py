"""CPU baseline with the REFERENCE'S STRUCTURE (test / bench infrastructure, never the
product path; only tests and bench.py's cpu_baseline leg import this).

The reference cannot travel to the GPU box, so the timed CPU baseline is a restatement
of how the reference does this path:

* per (gene, trait[, permutation]) a Python-level count over the isolates with
  dictionary look-ups per strain -- ``Perform_statistics``, scoary/methods.py:930-982
  (loop :950-965);
* ``scipy.stats.fisher_exact`` on ``[[tpgp, tpgn], [tngp, tngn]]`` memoised on the four
  counts -- scoary/methods.py:842-857 (memo dict :770, :850-857);
* parallelised the way the reference parallelises: a ``multiprocessing.Pool(n)`` whose
  worker k takes the stride domain ``range(k, G, n)`` -- scoary/methods.py:1076-1078,
  1083-1097; n = the logical CPUs this process may use (``usable_cpus``: affinity mask and
  cgroup quota applied to os.cpu_count());
* the permutation loop shuffles the trait labels over the valid isolates and
  re-evaluates the statistic, ``r += (stat_perm at least as extreme)`` --
  scoary/methods.py:1348-1355 -- with the Fisher p as the statistic (divergence D1,
  DESIGN.md section 7) and the labels of spec S4 (the oracle's ``perm_labels``), so that
  the result can be compared with the GPU's on the same sample.
"""
import multiprocessing
import os
import time

import numpy as np

TIE = 1e-7          # relative window of "p_perm <= p_obs" (same decision as spec S5, see below)

_SHARED = {}        # inherited by the forked workers


def usable_cpus():
    """Logical CPUs this process may actually run on: the smallest of os.cpu_count(), the
    scheduler affinity mask and the cgroup CPU quota (cpu.max / cfs_quota).  A container on a
    256-thread host is often granted far fewer; a Pool of os.cpu_count() workers then
    time-shares them and the per-worker rate collapses."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2
            q, per = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(per)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:     # cgroup v1
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                per = float(f.read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    if quota:
        n = min(n, max(1, int(quota + 0.5)))
    return max(1, n)


def _count(gene_row, labels, strains):
    """Perform_statistics (scoary/methods.py:950-965): one pass over the isolates of
    the trait (missing ones were removed from the dict, :591-598)."""
    tpgp = tpgn = tngp = tngn = 0
    for s in strains:
        if int(labels[s]) == 1 and gene_row[s] == 1:
            tpgp += 1
        elif int(labels[s]) == 1 and gene_row[s] == 0:
            tpgn += 1
        elif int(labels[s]) == 0 and gene_row[s] == 1:
            tngp += 1
        elif int(labels[s]) == 0 and gene_row[s] == 0:
            tngn += 1
    return tpgp, tpgn, tngp, tngn


def _warm(_k):
    import scipy.stats  # noqa: F401
    return 0


def _worker(k):
    import scipy.stats as ss
    sh = _SHARED
    n, G = sh["n"], sh["G"]
    strains, genedic, obs_labels, perm_labels = (sh["strains"], sh["genedic"], sh["obs"],
                                                 sh["perms"])
    memo = {}

    def fisher(c):
        if c not in memo:                                   # scoary/methods.py:850-857
            memo[c] = ss.fisher_exact([[c[0], c[1]], [c[2], c[3]]])
        return memo[c]

    out = []
    for g in range(k, G, n):                                # stride domain, :1076-1078
        row = genedic[g]
        c = _count(row, obs_labels, strains)
        if c[0] + c[2] == 0 or c[1] + c[3] == 0:            # skip rule, :804-814
            out.append((g, c, 1.0, len(perm_labels)))
            continue
        odds, p_obs = fisher(c)
        r = 0
        for lab in perm_labels:                             # :1348-1355
            cp = _count(row, lab, strains)
            if fisher(cp)[1] <= p_obs * (1.0 + TIE):
                r += 1
        out.append((g, c, float(p_obs), r))
    return out


def run(genes, trait, label_rows, processes=None):
    """genes (G, N) 0/1; trait (N,) with 2 = missing; label_rows (P, N) 0/1 permuted label
    vectors over the valid isolates.  Returns (counts [G,4] as tpgp,tpgn,tngp,tngn,
    p [G], r [G], seconds, processes)."""
    G, N = genes.shape
    n = processes or usable_cpus()
    n = max(1, min(n, G))
    valid = np.nonzero(trait != 2)[0]
    strains = ["iso%05d" % i for i in valid]
    _SHARED.clear()
    _SHARED.update(
        n=n, G=G, strains=strains,
        genedic=[{s: int(genes[g, i]) for s, i in zip(strains, valid)} for g in range(G)],
        obs={s: str(int(trait[i])) for s, i in zip(strains, valid)},
        perms=[{s: str(int(row[i])) for s, i in zip(strains, valid)} for row in label_rows])
    ctx = multiprocessing.get_context("fork")
    with ctx.Pool(n) as pool:
        pool.map(_warm, range(n), chunksize=1)              # workers up, scipy imported:
        t0 = time.perf_counter()                            # steady-state rate, not start-up
        parts = pool.map(_worker, range(n), chunksize=1)
        dt = time.perf_counter() - t0
    counts = np.zeros((G, 4), dtype=np.int32)
    p = np.zeros(G)
    r = np.zeros(G, dtype=np.int64)
    for part in parts:                                      # result weave, :1115-1122
        for g, c, pv, rv in part:
            counts[g] = c
            p[g] = pv
            r[g] = rv
    _SHARED.clear()
    return counts, p, r, dt, n


def sample_and_run(genes, trait, label_rows, target_s, processes=None):
    """Sizes the sample to about ``target_s`` seconds of wall time: a 4-gene probe in one
    worker, then the first Gs genes of a fixed pseudo-random order of ``genes`` (so the
    sample stays spread over the matrix) on all ``processes`` workers.  Returns
    (indices used, counts, p, r, seconds, processes)."""
    G = genes.shape[0]
    n = processes or usable_cpus()
    probe_g = min(G, 4)
    _, _, _, dt, _ = run(genes[:probe_g], trait, label_rows, processes=1)
    per_gene = max(dt / probe_g, 1e-4)
    Gs = int(min(G, max(n, n * round(target_s / per_gene))))
    order = np.random.default_rng(12345).permutation(G)[:Gs]
    order.sort()
    counts, p, r, dt, n = run(genes[order], trait, label_rows, processes=n)
    return order, counts, p, r, dt, n


def main(argv=None):
    """python -m oracle.scipy_baseline IN.npz OUT.npz TARGET_SECONDS -- run from a fresh
    interpreter (bench.py does: a Pool must not be forked from a process that holds a HIP
    context and runtime threads)."""
    import sys
    argv = argv or sys.argv[1:]
    d = np.load(argv[0])
    order, counts, p, r, dt, n = sample_and_run(d["genes"], d["trait"], d["labels"], float(argv[2]))
    np.savez(argv[1], order=order, counts=counts, p=p, r=r, dt=dt, n=n, cpu_count=os.cpu_count() or 0,
             usable=usable_cpus())


if __name__ == "__main__":
    main()
