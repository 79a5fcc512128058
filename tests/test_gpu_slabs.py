"""GPU checks of the slab-level C-ABI (what the multi-GPU driver composes):
operators applied slab by slab, with explicit halo rows, must reproduce the
whole-grid result bit for bit (same kernels, same arithmetic)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


import hipmem as hm      # device buffers on the HIP runtime libmgx is linked against (see hipmem.py)


def grid_from_interior(pkg, a, level, dt):
    """padded device grid (rows 0..N, level pitch) from an interior array"""
    N = 1 << level
    pitch = pkg.lib().mgx_level_pitch(level, pkg.DTYPE_F64 if dt == np.float64 else pkg.DTYPE_F32)
    g = np.zeros((N + 1, pitch), dtype=dt)
    g[1:N, 1:N] = a
    return hm.from_numpy(g)


def interior(t, level):
    N = 1 << level
    return t.cpu().numpy()[1:N, 1:N]


@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("smoother,level,mu,fuse", [("jacobi", 7, 3, 1), ("rbgs", 7, 3, 1), ("jacobi", 10, 5, 2),
                                                    ("jacobi", 10, 10, 4), ("jacobi", 10, 7, 5), ("jacobi", 10, 10, 10)])
def test_two_slabs_with_deep_halo_equal_whole_grid(pkg, po, dt, smoother, level, mu, fuse, monkeypatch):
    L = pkg.lib()
    monkeypatch.setenv("MGX_FUSE", str(fuse))
    monkeypatch.setenv("MGX_FUSE_ROWS", "16")
    N = 1 << level
    code = pkg.DTYPE_F64 if dt == np.float64 else pkg.DTYPE_F32
    rng = np.random.default_rng(11)
    v = rng.uniform(-1, 1, (N - 1, N - 1)).astype(dt)
    f = rng.uniform(-1, 1, (N - 1, N - 1)).astype(dt)
    U, B = grid_from_interior(pkg, v, level, dt), grid_from_interior(pkg, f, level, dt)
    ref = po.rbgs(v, f, mu) if smoother == "rbgs" else po.jacobi(v, f, mu)
    depth = 2 * mu if smoother == "rbgs" else mu
    out = np.zeros_like(v)
    half = N // 2
    for rank, (own_lo, own_hi) in enumerate([(0, half), (half, N + 1)]):
        lo = max(own_lo - depth, 0)
        hi = min(own_hi + depth, N + 1)
        u = U[lo:hi].clone(); b = B[lo:hi].clone(); tmp = hm.zeros_like(u)
        s = pkg.Slab(level=level, dtype=code, rows=hi - lo, row0=lo)
        flag = C.c_int()
        rl, rh = max(own_lo, 1) - lo, min(own_hi, N) - lo
        if smoother == "rbgs":
            st = L.mgx_slab_rbgs(C.byref(s), u.data_ptr(), b.data_ptr(), tmp.data_ptr(), rl, rh, mu, 1, C.byref(flag), None)
        else:
            st = L.mgx_slab_jacobi(C.byref(s), u.data_ptr(), b.data_ptr(), tmp.data_ptr(), rl, rh, mu, 2.0 / 3.0, 1, C.byref(flag), None)
        assert st == 0
        hm.synchronize()
        res = (tmp if flag.value else u).cpu().numpy()
        out[max(own_lo, 1) - 1: min(own_hi, N) - 1] = res[rl:rh, 1:N]
    tol = 1e-12 if dt == np.float64 else 1e-5
    assert np.max(np.abs(out - ref)) <= tol
    # and identical to the single-slab call (bitwise: same kernels)
    u = U.clone(); tmp = hm.zeros_like(u)
    s = pkg.Slab(level=level, dtype=code, rows=N + 1, row0=0)
    flag = C.c_int()
    if smoother == "rbgs":
        assert L.mgx_slab_rbgs(C.byref(s), u.data_ptr(), B.data_ptr(), tmp.data_ptr(), 1, N, mu, 0, C.byref(flag), None) == 0
    else:
        assert L.mgx_slab_jacobi(C.byref(s), u.data_ptr(), B.data_ptr(), tmp.data_ptr(), 1, N, mu, 2.0 / 3.0, 0, C.byref(flag), None) == 0
    hm.synchronize()
    whole = interior(tmp if flag.value else u, level)
    assert np.array_equal(out, whole)


@pytest.mark.parametrize("dt", [np.float64, np.float32])
def test_slab_restrict_prolong_and_norm(pkg, po, dt):
    L = pkg.lib()
    level = 7
    N, NC = 1 << level, 1 << (level - 1)
    code = pkg.DTYPE_F64 if dt == np.float64 else pkg.DTYPE_F32
    rng = np.random.default_rng(12)
    v = rng.uniform(-1, 1, (N - 1, N - 1)).astype(dt)
    f = rng.uniform(-1, 1, (N - 1, N - 1)).astype(dt)
    e = rng.uniform(-1, 1, (NC - 1, NC - 1)).astype(dt)
    U, B, E = grid_from_interior(pkg, v, level, dt), grid_from_interior(pkg, f, level, dt), grid_from_interior(pkg, e, level - 1, dt)
    tol = 1e-12 if dt == np.float64 else 2e-5
    # restriction of the residual, coarse rows split 16 | rest, fine slabs with 2-row halos
    ref_c = po.restrict(po.residual(v, f))
    got_c = np.zeros_like(ref_c)
    for (c_lo, c_hi) in [(1, 17), (17, NC)]:
        f_lo, f_hi = 2 * c_lo - 2, 2 * (c_hi - 1) + 3          # fine rows needed (exclusive hi)
        fs = pkg.Slab(level=level, dtype=code, rows=f_hi - f_lo, row0=f_lo)
        cs = pkg.Slab(level=level - 1, dtype=code, rows=c_hi - c_lo, row0=c_lo)
        cb = hm.zeros((c_hi - c_lo, E.shape[1]), E.dtype)
        cz = hm.ones_like(cb)
        st = L.mgx_slab_restrict(C.byref(fs), U[f_lo:f_hi].contiguous().data_ptr(), B[f_lo:f_hi].contiguous().data_ptr(),
                                 C.byref(cs), cb.data_ptr(), cz.data_ptr(), 0, c_hi - c_lo, 0, 1, None)
        assert st == 0
        hm.synchronize()
        got_c[c_lo - 1: c_hi - 1] = cb.cpu().numpy()[:, 1:NC]
        assert np.all(cz.cpu().numpy()[:, :NC] == 0)
    assert np.max(np.abs(got_c - ref_c)) <= tol * 32
    # prolongation + add on the lower half of the fine rows, coarse slab with one halo row
    ref_p = po.prolong_add(v, e)
    f_lo, f_hi = N // 2, N
    c_lo, c_hi = f_lo // 2, NC + 1
    fs = pkg.Slab(level=level, dtype=code, rows=f_hi - f_lo, row0=f_lo)
    cs = pkg.Slab(level=level - 1, dtype=code, rows=c_hi - c_lo, row0=c_lo)
    u = U[f_lo:f_hi].clone()
    st = L.mgx_slab_prolong(C.byref(fs), u.data_ptr(), C.byref(cs), E[c_lo:c_hi].contiguous().data_ptr(), 0, f_hi - f_lo, 1, None)
    assert st == 0
    hm.synchronize()
    assert np.max(np.abs(u.cpu().numpy()[:, 1:N] - ref_p[f_lo - 1:])) <= tol
    # residual sum of squares over a row range
    s = pkg.Slab(level=level, dtype=code, rows=N + 1, row0=0)
    scratch = hm.zeros(L.mgx_slab_scratch_doubles(C.byref(s)), np.float64)
    out = hm.zeros(1, np.float64)
    assert L.mgx_slab_residual_sumsq(C.byref(s), U.data_ptr(), B.data_ptr(), 10, 50, scratch.data_ptr(), out.data_ptr(), None) == 0
    hm.synchronize()
    r = po.residual(v, f)[9:49].astype(np.float64)
    assert abs(out.item() - np.sum(r * r)) <= (1e-12 if dt == np.float64 else 1e-5) * np.sum(r * r)


def test_slab_argument_validation(pkg):
    L = pkg.lib()
    s = pkg.Slab(level=6, dtype=pkg.DTYPE_F64, rows=10, row0=20)
    t = hm.zeros((10, L.mgx_level_pitch(6, pkg.DTYPE_F64)), np.float64)
    flag = C.c_int()
    # row range whose halo rows fall outside the slab must be refused, not run
    assert L.mgx_slab_jacobi(C.byref(s), t.data_ptr(), t.data_ptr(), t.data_ptr(), 0, 10, 1, 0.6, 0, C.byref(flag), None) != 0
    assert L.mgx_slab_jacobi(C.byref(s), t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 10, 1, 0.6, 0, C.byref(flag), None) != 0
    assert L.mgx_slab_rbgs(C.byref(s), t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 9, 1, 0, C.byref(flag), None) != 0
    assert L.mgx_slab_jacobi(C.byref(s), t.data_ptr(), t.data_ptr(), t.data_ptr(), 1, 9, 1, 0.6, 0, C.byref(flag), None) == 0
    hm.synchronize()


# path: "tile" = the register-tile kernel (what a range this small gets), "march" = the marching passes
# (MGX_SLAB_TILE_POINTS=0), "paired" = marching with the one-round paired chunk heights forced on (csrc/mgx_geom.hpp);
# extra: halo rows beyond the sweeps - 3 is exactly the cone of the residual + restriction stage, so the interior body
# runs on chunks whose loads beyond the cone are clamped to the rows that exist
@pytest.mark.parametrize("path,extra", [("tile", 4), ("march", 4), ("march", 3), ("paired", 3)])
@pytest.mark.parametrize("dt", [np.float64, np.float32])
@pytest.mark.parametrize("smoother,mu", [("jacobi", 10), ("jacobi", 3), ("rbgs", 3)])
def test_slab_cycle_equals_the_separate_slab_operators(pkg, po, dt, smoother, mu, path, extra, monkeypatch):
    """mgx_slab_cycle on an INTERIOR slab (halo rows on both sides, window narrower than the grid)
    against the oracle's whole-grid operators: correction on load, residual + restriction of
    the result, and the norm of the result - and its argument checks."""
    for k in ("MGX_SLAB_TILE_POINTS", "MGX_PAIR_MIN_ROWS", "MGX_MIN_CHUNK"):
        monkeypatch.delenv(k, raising=False)
    if path != "tile":
        monkeypatch.setenv("MGX_SLAB_TILE_POINTS", "0")
    if path == "paired":
        monkeypatch.setenv("MGX_PAIR_MIN_ROWS", "8")
        monkeypatch.setenv("MGX_MIN_CHUNK", "8")
    L = pkg.lib()
    level = 11 if path == "paired" else 9          # (the launcher sizes chunks itself from 2048^2 up)
    N, NC = 1 << level, 1 << (level - 1)
    code = pkg.DTYPE_F64 if dt == np.float64 else pkg.DTYPE_F32
    kind = pkg.SMOOTHER_RBGS if smoother == "rbgs" else pkg.SMOOTHER_JACOBI
    per = 2 if smoother == "rbgs" else 1
    rng = np.random.default_rng(21)
    v = rng.uniform(-1, 1, (N - 1, N - 1)).astype(dt)
    f = rng.uniform(-1, 1, (N - 1, N - 1)).astype(dt)
    e = rng.uniform(-1, 1, (NC - 1, NC - 1)).astype(dt)
    sm = po.rbgs if smoother == "rbgs" else po.jacobi
    own_lo, own_hi = N // 4, N // 2                # owned fine rows [128, 256) of 512: an interior slab
    halo = per * mu + extra
    lo, hi = own_lo - halo, own_hi + halo
    clo, chi = lo // 2 - 1, hi // 2 + 2            # coarse rows the slab holds
    U, B = grid_from_interior(pkg, v, level, dt), grid_from_interior(pkg, f, level, dt)
    E = grid_from_interior(pkg, e, level - 1, dt)
    fs = pkg.Slab(level=level, dtype=code, rows=hi - lo, row0=lo)
    cs = pkg.Slab(level=level - 1, dtype=code, rows=chi - clo, row0=clo)
    scratch = hm.zeros(int(L.mgx_slab_scratch_doubles(C.byref(fs))), np.float64)

    def run(pre, post, rl, rh, zero_in=0):
        u, b, tmp = U[lo:hi].clone(), B[lo:hi].clone(), hm.zeros_like(U[lo:hi])
        if zero_in:                                       # the call must not read u: poison it
            u = hm.from_numpy(np.full(u.shape, np.nan, dtype=dt))
        ce = E[clo:chi].clone()
        cb = hm.zeros_like(ce)
        out = hm.zeros(1, np.float64)
        flag = C.c_int()
        st = L.mgx_slab_cycle(C.byref(fs), u.data_ptr(), b.data_ptr(), tmp.data_ptr(), rl - lo, rh - lo, mu, 2.0 / 3.0, kind,
                              C.byref(cs), ce.data_ptr() if pre else None, cb.data_ptr() if post == 1 else None,
                              own_lo // 2 - clo, own_hi // 2 - clo, 0, zero_in, scratch.data_ptr() if post == 2 else None,
                              out.data_ptr() if post == 2 else None, C.byref(flag), None)
        hm.synchronize()
        return st, (tmp if flag.value else u).cpu().numpy(), cb.cpu().numpy(), float(out.item())

    # PRE: smooth(v + P e) on the owned rows
    st, got, _, _ = run(True, 0, own_lo, own_hi)
    assert st == 0
    ref = sm(po.prolong_add(v, e), f, mu)
    assert np.array_equal(got[own_lo - lo:own_hi - lo, 1:N], ref[own_lo - 1:own_hi - 1])
    # POST 1: restricted residual of the result on the owned coarse rows (range must start on an odd row)
    st, got, cb, _ = run(False, 1, own_lo - 1, own_hi + 1)
    assert st == 0
    ref = sm(v, f, mu)
    assert np.array_equal(got[own_lo - 1 - lo:own_hi + 1 - lo, 1:N], ref[own_lo - 2:own_hi])
    rr = po.restrict(po.residual(ref, f))
    assert np.array_equal(cb[own_lo // 2 - clo:own_hi // 2 - clo, 1:NC], rr[own_lo // 2 - 1:own_hi // 2 - 1])
    assert np.all(cb[:own_lo // 2 - clo] == 0) and np.all(cb[own_hi // 2 - clo:] == 0)      # nothing outside the rows asked for
    st, *_ = run(False, 1, own_lo, own_hi)
    assert st != 0                                   # an even first row cannot carry the folded restriction
    # POST 2: sum of squares of the residual of the result over the owned rows
    st, got, _, sq = run(True, 2, own_lo, own_hi)
    assert st == 0
    ref = sm(po.prolong_add(v, e), f, mu)
    r = po.residual(ref, f)[own_lo - 1:own_hi - 1].astype(np.float64)
    assert abs(sq - float(np.sum(r * r))) <= 1e-12 * float(np.sum(r * r))
    assert np.array_equal(got[own_lo - lo:own_hi - lo, 1:N], ref[own_lo - 1:own_hi - 1])
    # zero_in: the guess of a coarse-grid correction (PS:613) is synthesised, not read
    st, got, cb, _ = run(False, 1, own_lo - 1, own_hi + 1, zero_in=1)
    assert st == 0
    ref = sm(np.zeros_like(v), f, mu)
    assert np.array_equal(got[own_lo - 1 - lo:own_hi + 1 - lo, 1:N], ref[own_lo - 2:own_hi])
    rr = po.restrict(po.residual(ref, f))
    assert np.array_equal(cb[own_lo // 2 - clo:own_hi // 2 - clo, 1:NC], rr[own_lo // 2 - 1:own_hi // 2 - 1])
    st, *_ = run(True, 0, own_lo, own_hi, zero_in=1)
    assert st != 0                                   # a correction is added to an iterate that exists
