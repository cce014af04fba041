"""Synthetic inputs for the BASELINE.json configs.

Deterministic given the config's seed (numpy default_rng), generated on the
host in gene blocks so the 50k x 2000 headline matrix never needs more than a
few hundred MB of temporaries.
"""
import numpy as np

CONFIGS = {
    # name: G, N, T, P, seed
    "cfg2": dict(G=10_000, N=500, T=1, P=1_000, seed=20260901),
    "cfg3": dict(G=50_000, N=2_000, T=10, P=10_000, seed=20260902),
    "cfg4": dict(G=200_000, N=5_000, T=1, P=10_000, seed=20260903),
    "cfg5": dict(G=1_000_000, N=10_000, T=50, P=100_000, seed=20260904),
}


def make_genes(G, N, rng, kind="uniform", core_frac=0.0, block=4096):
    """(G, N) uint8 presence matrix.  kind="uniform": gene frequency
    f_g ~ U(0.02, 0.98); kind="rare": minor-allele frequency ~ Beta(0.3, 3)
    (VCF-like); kind="ushaped": f_g ~ Beta(0.15, 0.15) (U-shaped pan-genome spectrum);
    kind="balanced": f_g ~ U(0.4, 0.6) (the list-driven kernel's worst case).  core_frac of the genes are forced all-present / all-absent
    (alternating) to exercise the skip rule (methods.py:804-814)."""
    out = np.empty((G, N), dtype=np.uint8)
    for g0 in range(0, G, block):
        g1 = min(G, g0 + block)
        if kind == "rare":
            f = rng.beta(0.3, 3.0, size=(g1 - g0, 1)).astype(np.float32)
        elif kind == "ushaped":
            # pan-genome-like gene-frequency spectrum: most genes rare (cloud / shell) or near-core,
            # few at intermediate frequency -- Beta(0.15, 0.15); not a BASELINE config, an evidence
            # line next to cfg3 (bench.py --gene-kind ushaped)
            f = rng.beta(0.15, 0.15, size=(g1 - g0, 1)).astype(np.float32)
        elif kind == "balanced":
            # the list kernel's floor: every gene near 50 %, the minority lists as long as they get
            # (bench.py --gene-kind balanced)
            f = rng.uniform(0.4, 0.6, size=(g1 - g0, 1)).astype(np.float32)
        else:
            f = rng.uniform(0.02, 0.98, size=(g1 - g0, 1)).astype(np.float32)
        out[g0:g1] = rng.random((g1 - g0, N), dtype=np.float32) < f
    if core_frac > 0:
        idx = rng.choice(G, size=int(G * core_frac), replace=False)
        out[idx[0::2]] = 1
        out[idx[1::2]] = 0
    return out


def make_traits(T, N, rng, prevalence=None, missing_traits=(), missing_frac=0.01):
    """(T, N) uint8 with 0/1 and 2 = missing."""
    if prevalence is None:
        prevalence = rng.uniform(0.2, 0.8, size=T)
    prevalence = np.broadcast_to(np.asarray(prevalence, dtype=np.float64), (T,))
    tr = (rng.random((T, N)) < prevalence[:, None]).astype(np.uint8)
    for t in missing_traits:
        if t < T:
            tr[t, rng.random(N) < missing_frac] = 2
    return tr


def make_config(name, G=None, N=None, T=None, gene_kind=None):
    """Returns (genes[G,N] uint8, traits[T,N] uint8 (2 = missing), P, seed).
    G/N/T override the config's sizes (for small parity cases of the same
    distribution)."""
    c = dict(CONFIGS[name])
    if G is not None:
        c["G"] = G
    if N is not None:
        c["N"] = N
    if T is not None:
        c["T"] = T
    rng = np.random.default_rng(c["seed"])
    if gene_kind is not None:       # the config's shape and traits over another gene-frequency spectrum
        genes = make_genes(c["G"], c["N"], rng, kind=gene_kind,
                           core_frac=0.05 if name in ("cfg3", "cfg5") else 0.0)
        traits = make_traits(c["T"], c["N"], rng, prevalence={"cfg2": 0.35, "cfg4": 0.3}.get(name),
                             missing_traits=(8, 9) if name in ("cfg3", "cfg5") else ())
        return genes, traits, c["P"], c["seed"]
    if name == "cfg2":
        genes = make_genes(c["G"], c["N"], rng)
        traits = make_traits(c["T"], c["N"], rng, prevalence=0.35)
    elif name == "cfg4":
        genes = make_genes(c["G"], c["N"], rng, kind="rare")
        traits = make_traits(c["T"], c["N"], rng, prevalence=0.3)
    else:  # cfg3 / cfg5
        genes = make_genes(c["G"], c["N"], rng, core_frac=0.05)
        traits = make_traits(c["T"], c["N"], rng, missing_traits=(8, 9))
    return genes, traits, c["P"], c["seed"]
