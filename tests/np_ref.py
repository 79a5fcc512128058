"""Independent numpy statement of the grid operators (definitions only, no
schedules): a second opinion on the C oracle, written from the mathematical
definitions in SURVEY.md §8(a) rather than from the oracle's loops.
Interior-only n x n arrays, zero Dirichlet ring implied."""
import numpy as np


def nbr_sum(v):
    p = np.pad(v, 1)
    return ((p[:-2, 1:-1] + p[1:-1, :-2]) + p[1:-1, 2:]) + p[2:, 1:-1]


def apply_A(v):
    return 4.0 * v - nbr_sum(v)


def jacobi(v, f, mu, omega=2.0 / 3.0):
    dt = v.dtype.type
    om = dt(omega)
    c0 = dt(1.0 - float(om))
    c1 = dt(float(om) / 4.0)
    v = v.copy()
    for _ in range(mu):
        v = (c0 * v + c1 * f) + c1 * nbr_sum(v)
    return v


def rbgs(v, f, mu):
    v = v.copy()
    n = v.shape[0]
    ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
    for _ in range(mu):
        for colour in (0, 1):
            m = ((ii + jj) & 1) == colour
            new = v.dtype.type(0.25) * (f + nbr_sum(v))
            v[m] = new[m]
    return v


def residual(v, f):
    return f - apply_A(v)


def prolong(c):
    nc = c.shape[0]
    nf = 2 * nc + 1
    # 1-D bilinear interpolation matrix with zero boundary: (nf x nc)
    P = np.zeros((nf, nc), dtype=np.float64)
    for J in range(nc):
        i = 2 * J + 1            # 0-based fine index of coarse node J
        P[i, J] = 1.0
        P[i - 1, J] = 0.5
        P[i + 1, J] = 0.5
    return (P @ c.astype(np.float64) @ P.T).astype(c.dtype)


def restrict(fine, weight=0.25):
    """weight 0.25: R = P^T (consistent, D4); weight 1/16: full weighting."""
    nf = fine.shape[0]
    nc = (nf - 1) // 2
    P = np.zeros((nf, nc), dtype=np.float64)
    for J in range(nc):
        i = 2 * J + 1
        P[i, J] = 1.0
        P[i - 1, J] = 0.5
        P[i + 1, J] = 0.5
    # P^T F P has stencil [1 2 1; 2 4 2; 1 2 1]/4
    return (weight * 4.0 * (P.T @ fine.astype(np.float64) @ P)).astype(fine.dtype)
