"""ctypes view of oracle/build/libmg_oracle.so.

TEST INFRASTRUCTURE ONLY (parity unpinned, see mg_oracle.h).  Importers allowed:
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "build", "libmg_oracle.so")

SMOOTHER_JACOBI, SMOOTHER_RBGS = 0, 1
DTYPE_F32, DTYPE_F64, DTYPE_MIXED = 0, 1, 2
SCHEDULE_V, SCHEDULE_FMG = 0, 1
RESTRICT_CONSISTENT, RESTRICT_FW16, RESTRICT_INJECT, RESTRICT_INJECT4 = 0, 1, 2, 3
OP_POISSON, OP_STENCIL5 = 0, 1
BOTTOM_EXACT, BOTTOM_SMOOTH, BOTTOM_DST = 0, 1, 2
ARITH_SEPARATE, ARITH_FMA = 0, 1


class Config(C.Structure):
    _fields_ = [
        ("finest_level", C.c_int), ("coarsest_level", C.c_int),
        ("mu0", C.c_int), ("mu1", C.c_int), ("mu2", C.c_int),
        ("omega", C.c_double),
        ("smoother", C.c_int), ("dtype", C.c_int), ("schedule", C.c_int),
        ("restrict_mode", C.c_int), ("bottom", C.c_int), ("arith", C.c_int), ("op", C.c_int),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (make -C oracle)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("mg_oracle.c", "mg_oracle_impl.inc", "mg_oracle.h")
    ):
        subprocess.run(["make", "-C", _HERE], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        dp, fp, vp = C.POINTER(C.c_double), C.POINTER(C.c_float), C.c_void_p
        for suf, p in (("f64", dp), ("f32", fp)):
            getattr(L, f"orc_jacobi_{suf}").argtypes = [p, p, C.c_int, C.c_int, C.c_double]
            getattr(L, f"orc_jacobi_arith_{suf}").argtypes = [p, p, C.c_int, C.c_int, C.c_double, C.c_int]
            getattr(L, f"orc_rbgs_{suf}").argtypes = [p, p, C.c_int, C.c_int]
            getattr(L, f"orc_residual_{suf}").argtypes = [p, p, p, C.c_int]
            getattr(L, f"orc_restrict_{suf}").argtypes = [p, p, C.c_int, C.c_int]
            getattr(L, f"orc_prolong_{suf}").argtypes = [p, p, C.c_int]
            getattr(L, f"orc_prolong_add_{suf}").argtypes = [p, p, C.c_int]
            getattr(L, f"orc_var_build_jacobi_{suf}").argtypes = [p] * 5 + [C.c_int, C.c_double] + [p] * 5
            getattr(L, f"orc_var_jacobi_{suf}").argtypes = [p, p, C.c_int, C.c_int, C.c_double] + [p] * 5
            getattr(L, f"orc_var_residual_{suf}").argtypes = [p, p, p, C.c_int] + [p] * 5
            getattr(L, f"orc_restrict_inject_{suf}").argtypes = [p, p, C.c_int, C.c_double]
            i32 = C.POINTER(C.c_int32)
            getattr(L, f"orc_csr_gemv_{suf}").argtypes = [i32, i32, p, p, p, C.c_int, C.c_double]
            getattr(L, f"orc_csr_jacobi_{suf}").argtypes = [p, p, C.c_int, C.c_int, C.c_double, i32, i32, p, p]
            getattr(L, f"orc_var_set_stencil_{suf}").argtypes = [vp, C.c_int] + [dp] * 5
            getattr(L, f"orc_norm2_{suf}").argtypes = [p, C.c_size_t]
            getattr(L, f"orc_norm2_{suf}").restype = C.c_double
            getattr(L, f"orc_bottom_solve_{suf}").argtypes = [vp, p, p]
            getattr(L, f"orc_vcycle_{suf}").argtypes = [vp, C.c_int, p, p]
            getattr(L, f"orc_fmg_{suf}").argtypes = [vp, C.c_int, p, p]
            for fl in ("csr", "omp"):
                fn = getattr(L, f"orc_baseline_{fl}_jacobi_{suf}")
                fn.restype = C.c_double
                fn.argtypes = [p, p, C.c_int, C.c_int, C.c_double] + ([C.c_int] if fl == "omp" else [])
        L.orc_config_default.argtypes = [C.POINTER(Config)]
        L.orc_create.argtypes = [C.POINTER(Config)]
        L.orc_create.restype = vp
        L.orc_destroy.argtypes = [vp]
        L.orc_solve.argtypes = [vp, dp, dp, C.c_double, C.c_int, dp]
        L.orc_solve.restype = C.c_int
        L.orc_rhs_constant.argtypes = [dp, C.c_int, C.c_double]
        L.orc_rhs_sine.argtypes = [dp, C.c_int]
        L.orc_fill_uniform.argtypes = [dp, C.c_size_t, C.c_uint64]
        L.orc_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _suf(a: np.ndarray) -> str:
    if a.dtype == np.float64:
        return "f64"
    if a.dtype == np.float32:
        return "f32"
    raise TypeError(a.dtype)


def _ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.POINTER(C.c_double if a.dtype == np.float64 else C.c_float))


def default_config(**kw) -> Config:
    c = Config()
    lib().orc_config_default(C.byref(c))
    for k, v in kw.items():
        setattr(c, k, v)
    return c


# -- operators on interior n x n arrays (2-D numpy, C order) -------------------
def jacobi(v, f, mu, omega=2.0 / 3.0, arith=ARITH_SEPARATE):
    v = np.array(v, copy=True, order="C")
    f = np.ascontiguousarray(f, dtype=v.dtype)
    getattr(lib(), f"orc_jacobi_arith_{_suf(v)}")(_ptr(v), _ptr(f), v.shape[0], mu, omega, arith)
    return v


def rbgs(v, f, mu):
    v = np.array(v, copy=True, order="C")
    f = np.ascontiguousarray(f, dtype=v.dtype)
    getattr(lib(), f"orc_rbgs_{_suf(v)}")(_ptr(v), _ptr(f), v.shape[0], mu)
    return v


def residual(v, f):
    v = np.ascontiguousarray(v)
    f = np.ascontiguousarray(f, dtype=v.dtype)
    r = np.empty_like(v)
    getattr(lib(), f"orc_residual_{_suf(v)}")(_ptr(r), _ptr(v), _ptr(f), v.shape[0])
    return r


def restrict(fine, mode=RESTRICT_CONSISTENT):
    fine = np.ascontiguousarray(fine)
    nc = (fine.shape[0] - 1) // 2
    c = np.empty((nc, nc), dtype=fine.dtype)
    getattr(lib(), f"orc_restrict_{_suf(fine)}")(_ptr(c), _ptr(fine), fine.shape[0], mode)
    return c


def prolong(coarse):
    coarse = np.ascontiguousarray(coarse)
    nf = 2 * coarse.shape[0] + 1
    f = np.empty((nf, nf), dtype=coarse.dtype)
    getattr(lib(), f"orc_prolong_{_suf(coarse)}")(_ptr(f), _ptr(coarse), coarse.shape[0])
    return f


def prolong_add(v, coarse):
    v = np.array(v, copy=True, order="C")
    coarse = np.ascontiguousarray(coarse, dtype=v.dtype)
    getattr(lib(), f"orc_prolong_add_{_suf(v)}")(_ptr(v), _ptr(coarse), coarse.shape[0])
    return v


# -- general per-level operators (MF's draft): five-coefficient stencils and CSR ----------------
def var_build_jacobi(c, an, as_, aw, ae, omega=2.0 / 3.0):
    """A_jacobi_sp_dict from A_sp_dict (MF:28-32): (dinv, rn, rs, rw, re)"""
    a = [np.ascontiguousarray(x, dtype=c.dtype) for x in (c, an, as_, aw, ae)]
    out = [np.empty_like(a[0]) for _ in range(5)]
    getattr(lib(), f"orc_var_build_jacobi_{_suf(a[0])}")(*[_ptr(x) for x in a], a[0].shape[0], omega, *[_ptr(x) for x in out])
    return out


def var_jacobi(v, b, mu, omega, jac):
    """MF:75-96: mu sweeps of v <- R_omega v + omega D^-1 b; jac = var_build_jacobi(...)"""
    v = np.array(v, copy=True, order="C")
    b = np.ascontiguousarray(b, dtype=v.dtype)
    j = [np.ascontiguousarray(x, dtype=v.dtype) for x in jac]
    getattr(lib(), f"orc_var_jacobi_{_suf(v)}")(_ptr(v), _ptr(b), v.shape[0], mu, omega, *[_ptr(x) for x in j])
    return v


def var_residual(v, b, coef):
    """MF:150-153: b - A v, coef = (c, n, s, w, e)"""
    v = np.ascontiguousarray(v)
    b = np.ascontiguousarray(b, dtype=v.dtype)
    a = [np.ascontiguousarray(x, dtype=v.dtype) for x in coef]
    r = np.empty_like(v)
    getattr(lib(), f"orc_var_residual_{_suf(v)}")(_ptr(r), _ptr(v), _ptr(b), v.shape[0], *[_ptr(x) for x in a])
    return r


def restrict_inject(fine, weight=1.0):
    """MF:122-130"""
    fine = np.ascontiguousarray(fine)
    nc = (fine.shape[0] - 1) // 2
    c = np.empty((nc, nc), dtype=fine.dtype)
    getattr(lib(), f"orc_restrict_inject_{_suf(fine)}")(_ptr(c), _ptr(fine), fine.shape[0], weight)
    return c


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(C.POINTER(C.c_int32))


def csr_gemv(indptr, indices, values, x, alpha=1.0):
    """y = alpha (A x) on the reference's data layout (MF:33-41; oneMKL sparse::gemv, beta = 0)"""
    x = np.ascontiguousarray(x)
    values = np.ascontiguousarray(values, dtype=x.dtype)
    ip, ipp = _i32(indptr)
    ix, ixp = _i32(indices)
    y = np.empty_like(x)
    getattr(lib(), f"orc_csr_gemv_{_suf(x)}")(ipp, ixp, _ptr(values), _ptr(x), _ptr(y), len(ip) - 1, alpha)
    return y


def csr_jacobi(v, b, mu, omega, r_indptr, r_indices, r_values, dinv):
    """MF:75-96 on a CSR R_omega and the diagonal of D_inv"""
    v = np.array(v, copy=True, order="C")
    b = np.ascontiguousarray(b, dtype=v.dtype)
    r_values = np.ascontiguousarray(r_values, dtype=v.dtype)
    dinv = np.ascontiguousarray(dinv, dtype=v.dtype)
    ip, ipp = _i32(r_indptr)
    ix, ixp = _i32(r_indices)
    getattr(lib(), f"orc_csr_jacobi_{_suf(v)}")(_ptr(v), _ptr(b), v.size, mu, omega, ipp, ixp, _ptr(r_values), _ptr(dinv))
    return v


def stencil_from_nodes(a_nodes, level, finest):
    """five-point discretisation of -div(a grad u) on `level` from the nodal coefficient of the finest grid
    ((2^finest + 1)^2 values, boundary nodes included), sampled at the level's nodes; face coefficient = mean of
    its two nodes.  Returns interior n x n arrays (c, n, s, w, e) in double - the statement the device's
    mgx_set_coefficient follows operation by operation."""
    q = 1 << (finest - level)
    a = np.ascontiguousarray(a_nodes, dtype=np.float64)[::q, ::q]
    N = 1 << level
    assert a.shape == (N + 1, N + 1)
    ctr = a[1:N, 1:N]
    fn = 0.5 * (ctr + a[0:N - 1, 1:N])
    fs = 0.5 * (ctr + a[2:N + 1, 1:N])
    fw = 0.5 * (ctr + a[1:N, 0:N - 1])
    fe = 0.5 * (ctr + a[1:N, 2:N + 1])
    c = ((fn + fw) + fe) + fs
    return c, -fn, -fs, -fw, -fe


def norm2(x):
    x = np.ascontiguousarray(x)
    return getattr(lib(), f"orc_norm2_{_suf(x)}")(_ptr(x), x.size)


def rhs_constant(level, f=4.0):
    n = (1 << level) - 1
    b = np.empty((n, n), dtype=np.float64)
    lib().orc_rhs_constant(_ptr(b), level, f)
    return b


def rhs_sine(level):
    n = (1 << level) - 1
    b = np.empty((n, n), dtype=np.float64)
    lib().orc_rhs_sine(_ptr(b), level)
    return b


def fill_uniform(shape, seed=12345):
    u = np.empty(shape, dtype=np.float64)
    lib().orc_fill_uniform(_ptr(u), u.size, seed)
    return u


class Solver:
    """orc_solver handle (PS:575-650 schedules on the CPU)."""

    def __init__(self, **cfg):
        self.cfg = default_config(**cfg)
        self._h = lib().orc_create(C.byref(self.cfg))
        if not self._h:
            raise ValueError("invalid oracle config")

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def n(self, level=None):
        return (1 << (self.cfg.finest_level if level is None else level)) - 1

    def bottom_solve(self, rhs):
        rhs = np.ascontiguousarray(rhs)
        x = np.empty_like(rhs)
        getattr(lib(), f"orc_bottom_solve_{_suf(rhs)}")(self._h, _ptr(x), _ptr(rhs))
        return x

    def set_stencil(self, level, c, an, as_, aw, ae):
        """ProblemVar::A_sp_dict[level] (MF:19): the level's operator as five interior coefficient arrays"""
        a = [np.ascontiguousarray(x, dtype=np.float64) for x in (c, an, as_, aw, ae)]
        suf = "f64" if self.cfg.dtype == DTYPE_F64 else "f32"
        getattr(lib(), f"orc_var_set_stencil_{suf}")(self._h, level, *[_ptr(x) for x in a])

    def set_coefficient(self, a_nodes):
        """every level's operator from the nodal coefficient of the finest grid (stencil_from_nodes)"""
        for lvl in range(self.cfg.coarsest_level, self.cfg.finest_level + 1):
            self.set_stencil(lvl, *stencil_from_nodes(a_nodes, lvl, self.cfg.finest_level))

    def vcycle(self, level, v, f):
        v = np.array(v, copy=True, order="C")
        f = np.ascontiguousarray(f, dtype=v.dtype)
        getattr(lib(), f"orc_vcycle_{_suf(v)}")(self._h, level, _ptr(v), _ptr(f))
        return v

    def fmg(self, level, f):
        f = np.ascontiguousarray(f)
        v = np.zeros_like(f)
        getattr(lib(), f"orc_fmg_{_suf(f)}")(self._h, level, _ptr(v), _ptr(f))
        return v

    def solve(self, b, u0=None, tol=1e-8, max_cycles=50):
        b = np.ascontiguousarray(b, dtype=np.float64)
        u = np.zeros_like(b) if u0 is None else np.array(u0, dtype=np.float64, order="C")
        hist = np.zeros(max_cycles + 1, dtype=np.float64)
        k = lib().orc_solve(self._h, _ptr(b), _ptr(u), tol, max_cycles, _ptr(hist))
        return u, hist[: k + 1].copy()


def baseline_jacobi(kind, v, f, mu, omega=2.0 / 3.0, threads=1):
    """Time `mu` Jacobi sweeps on the CPU; kind = 'csr' (reference-shaped,
    PS:137-145, one thread) or 'omp' (matrix-free, OpenMP).  Returns seconds."""
    v = np.array(v, copy=True, order="C")
    f = np.ascontiguousarray(f, dtype=v.dtype)
    fn = getattr(lib(), f"orc_baseline_{kind}_jacobi_{_suf(v)}")
    if kind == "omp":
        return fn(_ptr(v), _ptr(f), v.shape[0], mu, omega, threads), v
    return fn(_ptr(v), _ptr(f), v.shape[0], mu, omega), v
