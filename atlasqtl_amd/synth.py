"""Synthetic hotspot-QTL data in the shape of the reference's documented example
(R/atlasqtl.R:125-157) and of its test generator (tests/testthat/main.R:9-22):
SNP dosages X_ij ~ Binomial(2, maf_j), a few "hotspot" SNPs acting on a subset
of the traits with N(0,1) effects, Gaussian noise.  Seeded with NumPy's PCG64
(R's Mersenne stream is not reproducible without R), so "the same synthetic
(X, Y)" means the arrays produced here, handed to both sides.
"""
from __future__ import annotations

import numpy as np


def simulate(n, p, q, p_act=10, q_act=None, seed=123, maf=(0.05, 0.5), prob_assoc=0.2, na_frac=0.0,
             dtype=np.float64):
    """Returns dict(X raw dosages n x p, Y n x q (NaN where missing), pat p x q bool, beta).

    maf: scalar (tests use 0.2 / 0.25 fixed) or (lo, hi) for U(lo, hi) per SNP."""
    rng = np.random.default_rng(seed)
    q_act = q if q_act is None else q_act
    if np.isscalar(maf):
        mafs = np.full(p, float(maf))
    else:
        mafs = rng.uniform(maf[0], maf[1], size=p)
    X = rng.binomial(2, mafs[None, :], size=(n, p)).astype(np.int8)
    act_x = np.sort(rng.choice(p, size=p_act, replace=False))
    act_y = np.sort(rng.choice(q, size=q_act, replace=False))
    pat = np.zeros((p, q), dtype=bool)
    sub = rng.random((p_act, q_act)) < prob_assoc
    pat[np.ix_(act_x, act_y)] = sub
    beta = np.zeros((p_act, q_act))
    beta[sub] = rng.normal(size=int(sub.sum()))
    Y = rng.normal(size=(n, q))
    Y[:, act_y] += X[:, act_x].astype(np.float64) @ beta
    if na_frac > 0:
        Y[rng.random((n, q)) < na_frac] = np.nan
    return dict(X=X.astype(dtype), Y=Y, pat=pat, act_x=act_x, act_y=act_y, beta=beta)
