"""Writes tests/golden/ref_ps.npz: outputs of the REFERENCE's own functions.

The data in ref_ps.npz was computed by code compiled from /root/reference/Poissons_SYCL.cpp
itself (oracle/build_ref.sh cuts the standard-C++ functions out of it, g++, no stand-ins;
library oracle/_ref/libps_ref.so).  These are the only reference-computed vectors that exist:
the reference ships no tests or fixtures, and its smoother / cycle functions need oneMKL + SYCL.

What is pinned by it (tests/test_ref_pins.py, tests/test_gpu_ref_pins.py):
  A4  interpolation2d  PS:337-425  (fp32, the reference's type) - seeded inputs 3^2 .. 63^2
  A10 globalforcefunction PS:283-335 at levels 7..10 (sign: defect D1)
  stencil / numbering: raw COO triplets of globalstiffenssmatrix PS:200-281 on 9x9 and 17x17 meshes
  D2  coo_to_csr PS:55-116 as written (int32 accumulator)    D3  restriction2d PS:531-546 as written (== 0)

Run from the repo root in the build container (needs /root/reference):
    bash oracle/build_ref.sh && python tests/golden/make_ref_golden.py
"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
LIB = os.path.join(ROOT, "oracle", "_ref", "libps_ref.so")


def load():
    L = C.CDLL(LIB)
    fp, ip = C.POINTER(C.c_float), C.POINTER(C.c_int)
    L.ref_interpolation2d.argtypes = [fp, C.c_int, fp]
    L.ref_restriction2d.argtypes = [fp, C.c_int, fp]
    L.ref_globalforcefunction.argtypes = [C.c_int, fp, C.c_int]
    L.ref_globalstiffenssmatrix.argtypes = [C.c_int, ip, ip, fp, ip, ip, ip, fp, ip]
    L.ref_coo_to_csr.argtypes = [C.c_int, C.c_int, C.c_int, ip, ip, fp, fp, ip, ip]
    return L


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def interpolation2d(L, e):
    nc = e.shape[0]
    out = np.zeros((2 * nc + 1, 2 * nc + 1), dtype=np.float32)
    n = L.ref_interpolation2d(fptr(np.ascontiguousarray(e)), nc, fptr(out))
    assert n == out.size
    return out


def restriction2d(L, f):
    nf = f.shape[0]
    nc = (nf - 1) // 2
    out = np.full((nc, nc), np.nan, dtype=np.float32)
    n = L.ref_restriction2d(fptr(np.ascontiguousarray(f)), nf, fptr(out))
    assert n == out.size
    return out


def globalforcefunction(L, level):
    n = (1 << level) - 1
    out = np.zeros((n, n), dtype=np.float32)
    got = L.ref_globalforcefunction(level, fptr(out), out.size)
    assert got == out.size, got
    return out


def stiffness_coo(L, nodes_per_side):
    n_lu, n_d = C.c_int(0), C.c_int(0)
    L.ref_globalstiffenssmatrix(nodes_per_side ** 2, None, None, None, C.byref(n_lu), None, None, None, C.byref(n_d))
    rl, cl = np.zeros(n_lu.value, np.int32), np.zeros(n_lu.value, np.int32)
    vl = np.zeros(n_lu.value, np.float32)
    rd, cd = np.zeros(n_d.value, np.int32), np.zeros(n_d.value, np.int32)
    vd = np.zeros(n_d.value, np.float32)
    L.ref_globalstiffenssmatrix(nodes_per_side ** 2, iptr(rl), iptr(cl), fptr(vl), C.byref(n_lu), iptr(rd), iptr(cd), fptr(vd),
                                C.byref(n_d))
    return rl, cl, vl, rd, cd, vd


def coo_to_csr(L, nrows, r, c, v):
    nnz = len(v)
    data, indices = np.zeros(nnz, np.float32), np.zeros(nnz, np.int32)
    indptr = np.zeros(nrows + 1, np.int32)
    m = L.ref_coo_to_csr(nrows, nrows, nnz, iptr(r), iptr(c), fptr(v), fptr(data), iptr(indptr), iptr(indices))
    return data[:m], indptr, indices[:m]


def main():
    if not os.path.exists(LIB):
        sys.exit(f"{LIB} missing: run bash oracle/build_ref.sh first (needs /root/reference)")
    L = load()
    d = {}
    rng = np.random.default_rng(20261004)
    # A4: interpolation2d on seeded inputs and on ones(3x3) (SURVEY §4's pin)
    for nc in (1, 3, 7, 31, 63):
        e = rng.uniform(-1, 1, (nc, nc)).astype(np.float32)
        d[f"interp_in_{nc}"] = e
        d[f"interp_out_{nc}"] = interpolation2d(L, e)
    d["interp_ones3"] = interpolation2d(L, np.ones((3, 3), np.float32))
    # A10: the load vector at levels 7..10 (constant fields: they compress to nothing)
    for level in (7, 8, 9, 10):
        d[f"force_L{level}"] = globalforcefunction(L, level)
    # stencil + numbering: raw COO triplets, 9x9 and 17x17 node meshes (7^2 / 15^2 unknowns)
    for nodes in (9, 17):
        rl, cl, vl, rd, cd, vd = stiffness_coo(L, nodes)
        d[f"coo{nodes}_rows_lu"], d[f"coo{nodes}_cols_lu"], d[f"coo{nodes}_vals_lu"] = rl, cl, vl
        d[f"coo{nodes}_rows_d"], d[f"coo{nodes}_cols_d"], d[f"coo{nodes}_vals_d"] = rd, cd, vd
        # D2: what coo_to_csr makes of them, as written
        n = (nodes - 2) ** 2
        for nm, (r, c, v) in (("lu", (rl, cl, vl)), ("d", (rd, cd, vd))):
            data, indptr, indices = coo_to_csr(L, n, r, c, v)
            d[f"csr{nodes}_{nm}_data"], d[f"csr{nodes}_{nm}_indptr"], d[f"csr{nodes}_{nm}_indices"] = data, indptr, indices
    # D3: restriction2d as written
    f7 = rng.uniform(-1, 1, (7, 7)).astype(np.float32)
    d["restrict_in_7"] = f7
    d["restrict_out_7"] = restriction2d(L, f7)
    out = os.path.join(HERE, "ref_ps.npz")
    np.savez_compressed(out, **d)
    print(f"wrote {out}: {os.path.getsize(out)} bytes, {len(d)} arrays")
    for k in ("interp_ones3",):
        print(k, d[k])
    print("force L10 distinct values:", np.unique(d["force_L10"]))
    print("coo9: D values", np.unique(d["coo9_vals_d"], return_counts=True), "LU values", np.unique(d["coo9_vals_lu"], return_counts=True))
    print("csr9 lu data distinct:", np.unique(d["csr9_lu_data"]), " d:", np.unique(d["csr9_d_data"]))
    print("restriction2d as written:", np.unique(d["restrict_out_7"]))


if __name__ == "__main__":
    main()
