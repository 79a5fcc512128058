// mgx_reference_api.hpp — the reference's own free-function interface, header
// only, on top of the C-ABI of include/mgx.h.
//
// nikhilTkur/Multigrid_Nikhil_C- has no plugin/FFI layer: its "API" is the set
// of free functions main() calls.  This header reproduces those
// names, argument orders and ownership rules so a driver shaped like
// Poissons_SYCL.cpp's main() (PS:658-731) compiles against libmgx unchanged in
// structure:
//
//   reference (PS = Poissons_SYCL.cpp, MF = Multigrid_functions.cpp)    here
//   ------------------------------------------------------------------  ---------------------------
//   cl::sycl::queue q;                                     PS:659       mgxref::queue q;
//   jacobi_matrices[level - coarsest_level] = {...}        PS:33,661-690 mgxref::build_hierarchy(...)
//   jacobirelaxation(q, a_lu, a_size, v, fh, mu)           PS:125       same name, same six arguments
//                                                                       (+ a 5-argument overload taking the level entry)
//   restriction2d(vec_h)                                   PS:531       same
//   interpolation2d(vec_2h)                                PS:337       same
//   vcyclemultigrid(q, a_h, vec_h, f_h)                    PS:575       same
//   fullmultigrid(q, a_h, f_h)                             PS:629       same
//   globalforcefunction()                                  PS:283       same (SPD sign)
//   multigrid_solver(ProblemVar&)                          MF:193       mgxref::multigrid_solver(b)
//
// Conventions kept from the reference: the caller owns every std::vector, each
// function returns a fresh vector by value, jacobirelaxation also updates `v`
// in place (PS:146), vcyclemultigrid clobbers vec_h (PS:581), level data lives
// in a global (PS:33), everything is synchronous.  Errors, which the reference
// lets escape as SYCL/oneMKL exceptions, surface here as std::runtime_error.
//
// Real = float reproduces PS's precision, Real = double MF's.
#pragma once

#include "mgx.h"

#include <cmath>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace mgxref {

struct queue {};   // stands where cl::sycl::queue stands in the reference's signatures (PS:659)

// oneapi::mkl::sparse::matrix_handle_t in the reference's signatures (PS:26, 28, 125): what a
// sparse-matrix handle is in a matrix-free solver - the hierarchy it belongs to and its level.
struct matrix_handle_t {
    mgx_handle handle = nullptr;
    int level = 0;
};

// PS:24-30 matrix_elements_for_jacobi, same three members in the same order (the diagonal and
// the off-diagonal part of one level's operator are the same matrix-free stencil here), plus the
// two fields the wrappers below read directly.
struct matrix_elements_for_jacobi {
    matrix_handle_t a_d_handle;      // PS:26
    matrix_handle_t a_lu_handle;     // PS:28
    std::int32_t size = 0;           // PS:29: number of unknowns (PS:689)
    mgx_handle handle = nullptr;
    int level = 0;
};

// the reference's compile-time globals (PS:17-22), run-time here
struct parameters {
    int finest_level = 10;
    int coarsest_level = 7;
    int mu0 = 30, mu1 = 10, mu2 = 10;
    double omega = 2.0 / 3.0;                   // PS:127
    int smoother = MGX_SMOOTHER_JACOBI;
    int restrict_mode = MGX_RESTRICT_CONSISTENT;
    int bottom = MGX_BOTTOM_EXACT;
    float f = 4.0f;                             // PS:123
    int device = 0;
    int n_gpus = 1;                             // PS:659 has one queue; > 1: row slabs over that many GPUs
    int devices[MGX_MAX_GPUS] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
};

template <typename Real> struct hierarchy {
    parameters prm;
    mgx_handle handle = nullptr;
    std::vector<matrix_elements_for_jacobi> jacobi_matrices;   // PS:33
    ~hierarchy() { if (handle) mgx_destroy(handle); }
};

template <typename Real> inline hierarchy<Real>& global_hierarchy()
{
    static hierarchy<Real> h;
    return h;
}

inline void check(int status, mgx_handle h, const char* what)
{
    if (status != MGX_OK)
        throw std::runtime_error(std::string(what) + ": " + mgx_status_string(status) + ": " + mgx_last_error(h));
}

// PS:661-690: the per-level loop of main().  No stiffness matrices, no CSR:
// the constant 5-point operator is applied matrix-free.
template <typename Real>
inline std::vector<matrix_elements_for_jacobi>& build_hierarchy(const parameters& prm)
{
    hierarchy<Real>& H = global_hierarchy<Real>();
    if (H.handle) { mgx_destroy(H.handle); H.handle = nullptr; }
    H.prm = prm;
    mgx_config cfg;
    mgx_config_default(&cfg);
    cfg.finest_level = prm.finest_level;
    cfg.coarsest_level = prm.coarsest_level;
    cfg.mu0 = prm.mu0; cfg.mu1 = prm.mu1; cfg.mu2 = prm.mu2;
    cfg.omega = prm.omega;
    cfg.smoother = prm.smoother;
    cfg.dtype = sizeof(Real) == 8 ? MGX_DTYPE_F64 : MGX_DTYPE_F32;
    cfg.schedule = MGX_SCHEDULE_FMG;
    cfg.restrict_mode = prm.restrict_mode;
    cfg.bottom = prm.bottom;
    cfg.device = prm.device;
    cfg.n_gpus = prm.n_gpus;
    for (int i = 0; i < MGX_MAX_GPUS; ++i) cfg.devices[i] = prm.devices[i];
    check(mgx_create(&cfg, &H.handle), nullptr, "mgx_create");
    H.jacobi_matrices.assign(prm.finest_level - prm.coarsest_level + 1, {});
    for (int level = prm.coarsest_level; level <= prm.finest_level; ++level) {   // PS:661
        const std::int32_t n = mgx_level_n(level);
        const matrix_handle_t mh{H.handle, level};
        H.jacobi_matrices[level - prm.coarsest_level] = {mh, mh, n * n, H.handle, level};   // PS:665, 680-689
    }
    return H.jacobi_matrices;
}

// level of a vector from its length, as the reference does (PS:583, 616, 634)
inline int level_of_size(std::size_t size) { return int(std::log2(std::sqrt(double(size)) + 1)); }

template <typename Real> inline void put(mgx_handle h, int level, int which, const std::vector<Real>& v)
{
    check(mgx_set_level(h, level, which, v.data(), v.size()), h, "mgx_set_level");
}
template <typename Real> inline std::vector<Real> get(mgx_handle h, int level, int which)
{
    const std::size_t n = std::size_t(mgx_level_n(level));
    std::vector<Real> v(n * n);
    check(mgx_get_level(h, level, which, v.data(), v.size()), h, "mgx_get_level");
    return v;
}

// PS:125-147, the reference's own six arguments (called as PS:581 does:
// jacobirelaxation(q, a_h.a_lu_handle, a_h.size, vec_h, f_h, mu1)); mutates v AND returns it (PS:146).
// a_size is checked against the handle's level, as the only use the reference makes of it
// (PS:129-131 sizes its scratch vectors with it).
template <typename Real>
inline std::vector<Real> jacobirelaxation(queue&, matrix_handle_t a_lu, std::int32_t a_size, std::vector<Real>& v,
                                          std::vector<Real>& fh, const int& mu)
{
    const std::int32_t n = mgx_level_n(a_lu.level);
    if (a_size != n * n || v.size() != std::size_t(a_size) || fh.size() != std::size_t(a_size))
        throw std::runtime_error("jacobirelaxation: a_size / vector lengths do not match the level of a_lu");
    put(a_lu.handle, a_lu.level, MGX_VEC_U, v);
    put(a_lu.handle, a_lu.level, MGX_VEC_B, fh);
    check(mgx_smooth(a_lu.handle, a_lu.level, mu), a_lu.handle, "mgx_smooth");
    v = get<Real>(a_lu.handle, a_lu.level, MGX_VEC_U);
    return v;
}

// convenience overload on the level entry
template <typename Real>
inline std::vector<Real> jacobirelaxation(queue& q, matrix_elements_for_jacobi& a, std::vector<Real>& v,
                                          std::vector<Real>& fh, const int& mu)
{
    return jacobirelaxation<Real>(q, a.a_lu_handle, a.size, v, fh, mu);
}

// PS:531-546
template <typename Real> inline std::vector<Real> restriction2d(std::vector<Real>& vec_h)
{
    hierarchy<Real>& H = global_hierarchy<Real>();
    const int level = level_of_size(vec_h.size());
    put(H.handle, level, MGX_VEC_B, vec_h);
    check(mgx_restrict_rhs(H.handle, level), H.handle, "mgx_restrict_rhs");
    return get<Real>(H.handle, level - 1, MGX_VEC_B);
}

// PS:337-425
template <typename Real> inline std::vector<Real> interpolation2d(std::vector<Real>& vec_2h)
{
    hierarchy<Real>& H = global_hierarchy<Real>();
    const int level = level_of_size(vec_2h.size()) + 1;
    put(H.handle, level - 1, MGX_VEC_U, vec_2h);
    check(mgx_prolong(H.handle, level), H.handle, "mgx_prolong");
    return get<Real>(H.handle, level, MGX_VEC_U);
}

// PS:575-627.  vec_h is clobbered (PS:581 assigns the pre-smoothed vector to it).
template <typename Real>
inline std::vector<Real> vcyclemultigrid(queue&, matrix_elements_for_jacobi& a_h, std::vector<Real>& vec_h,
                                         std::vector<Real>& f_h)
{
    put(a_h.handle, a_h.level, MGX_VEC_U, vec_h);
    put(a_h.handle, a_h.level, MGX_VEC_B, f_h);
    check(mgx_vcycle(a_h.handle, a_h.level), a_h.handle, "mgx_vcycle");
    vec_h = get<Real>(a_h.handle, a_h.level, MGX_VEC_U);
    return vec_h;
}

// PS:629-650 (a_h must be the finest level, as in PS:727)
template <typename Real>
inline std::vector<Real> fullmultigrid(queue&, matrix_elements_for_jacobi& a_h, std::vector<Real>& f_h)
{
    put(a_h.handle, a_h.level, MGX_VEC_B, f_h);
    check(mgx_fmg(a_h.handle), a_h.handle, "mgx_fmg");
    return get<Real>(a_h.handle, a_h.level, MGX_VEC_U);
}

// PS:283-335: the assembled load vector is the constant f h^2 at every interior
// node (six triangles of area h^2/2, one third each); SPD sign per SURVEY D1.
template <typename Real> inline std::vector<Real> globalforcefunction()
{
    const parameters& prm = global_hierarchy<Real>().prm;
    const std::int32_t highest_size = std::int32_t(std::pow(2, prm.finest_level));   // PS:19
    const Real h = Real(1.0) / Real(highest_size);                                    // PS:289
    const std::size_t n = std::size_t(highest_size) - 1;
    return std::vector<Real>(n * n, Real(prm.f) * h * h);                             // PS:291
}

// MF:193-197 run to a tolerance, with the residual report the reference lacks (D10)
template <typename Real>
inline std::vector<Real> multigrid_solver(std::vector<Real>& b, double tol, int max_cycles,
                                          mgx_stats* stats = nullptr, std::vector<double>* history = nullptr)
{
    hierarchy<Real>& H = global_hierarchy<Real>();
    const int L = H.prm.finest_level;
    put(H.handle, L, MGX_VEC_B, b);
    std::vector<Real> zero(b.size(), Real(0));
    put(H.handle, L, MGX_VEC_U, zero);                                                // PS:630
    std::vector<double> hist(std::size_t(max_cycles) + 1, 0.0);
    mgx_stats st{};
    check(mgx_solve(H.handle, tol, max_cycles, &st, hist.data(), int(hist.size())), H.handle, "mgx_solve");
    hist.resize(std::size_t(st.history_len));
    if (stats) *stats = st;
    if (history) *history = hist;
    return get<Real>(H.handle, L, MGX_VEC_U);
}

} // namespace mgxref
