/*
 * mg_oracle.c — CPU oracle for the 2-D Poisson multigrid hot path.
 * TEST INFRASTRUCTURE ONLY.  Parity: prolongation, load vector and the stencil
 * are pinned by reference-computed data (oracle/_ref, tests/golden/ref_ps.npz);
 * the smoother, residual and the schedules are UNPINNED (the reference's
 * oneMKL/SYCL path cannot be built in this image) — see mg_oracle.h.
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared)
 */
#define _POSIX_C_SOURCE 200809L
#include "mg_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAX_LEVELS 32

struct orc_solver {
    orc_config cfg;
    /* per-level scratch, indexed by level (PS:33 indexes level - coarsest) */
    void* tmp_f64[ORC_MAX_LEVELS];
    void* res_f64[ORC_MAX_LEVELS];
    void* rhs_f64[ORC_MAX_LEVELS];
    void* sol_f64[ORC_MAX_LEVELS];
    void* fmg_rhs_f64[ORC_MAX_LEVELS];
    void* fmg_sol_f64[ORC_MAX_LEVELS];
    void* tmp_f32[ORC_MAX_LEVELS];
    void* res_f32[ORC_MAX_LEVELS];
    void* rhs_f32[ORC_MAX_LEVELS];
    void* sol_f32[ORC_MAX_LEVELS];
    void* fmg_rhs_f32[ORC_MAX_LEVELS];
    void* fmg_sol_f32[ORC_MAX_LEVELS];
    /* banded Cholesky factor of the coarsest operator */
    double* chol;        /* N x (bw+1), chol[i*(bw+1) + (i-j)] = L_ij */
    double* bottom_work; /* N */
    int chol_ready;
    /* sine-transform bottom solve (ORC_BOTTOM_DST) */
    double* dst_S;       /* n x n, S_jk = sqrt(2/(n+1)) sin(pi (j+1)(k+1)/(n+1)) */
    double* dst_s;       /* n, 4 sin^2(pi (i+1) / (2 (n+1))) */
    double* dst_w1;      /* n x n work */
    double* dst_w2;
    /* general per-level operators (ORC_OP_STENCIL5; mg_oracle_var.inc): A = (c, n, s, w, e) and its
     * Jacobi splitting (D_inv, R_n, R_s, R_w, R_e) per level and precision; dense inverse of the coarsest */
    void* var_a_f64[ORC_MAX_LEVELS][5];
    void* var_j_f64[ORC_MAX_LEVELS][5];
    void* var_a_f32[ORC_MAX_LEVELS][5];
    void* var_j_f32[ORC_MAX_LEVELS][5];
    double* var_inv;
};

static inline int orc_n(int level) { return (1 << level) - 1; } /* PS:662-664 */

void orc_config_default(orc_config* c)
{
    c->finest_level = 10;   /* PS:17 */
    c->coarsest_level = 7;  /* PS:18 */
    c->mu0 = 30;            /* PS:20 */
    c->mu1 = 10;            /* PS:21 */
    c->mu2 = 10;            /* PS:22 */
    c->omega = 2.0 / 3.0;   /* PS:127 */
    c->smoother = ORC_SMOOTHER_JACOBI;
    c->dtype = ORC_DTYPE_F64;
    c->schedule = ORC_SCHEDULE_FMG; /* PS:727 calls fullmultigrid */
    c->restrict_mode = ORC_RESTRICT_CONSISTENT;
    c->bottom = ORC_BOTTOM_EXACT;
    c->arith = ORC_ARITH_SEPARATE;
    c->op = ORC_OP_POISSON;
}

/* ---- exact bottom solver: banded Cholesky (stands in for Eigen SparseLU,
 * MF:63-72; any exact solver gives the same x up to rounding). ------------- */
static void orc_chol_factor(orc_solver* s)
{
    const int n = orc_n(s->cfg.coarsest_level);
    const int bw = n;
    const size_t N = (size_t)n * n;
    const size_t ld = (size_t)bw + 1;
    double* L = (double*)calloc(N * ld, sizeof(double));
    for (size_t i = 0; i < N; ++i) {
        L[i * ld + 0] = 4.0;
        if (i % (size_t)n != 0) L[i * ld + 1] = -1.0;   /* west neighbour */
        if (i >= (size_t)n) L[i * ld + bw] = -1.0;      /* north neighbour */
    }
    for (size_t j = 0; j < N; ++j) {
        /* diagonal */
        double d = L[j * ld];
        const size_t k0 = (j > (size_t)bw) ? j - bw : 0;
        for (size_t k = k0; k < j; ++k) {
            const double l = L[j * ld + (j - k)];
            d -= l * l;
        }
        d = sqrt(d);
        L[j * ld] = d;
        const size_t iend = (j + bw < N) ? j + bw : N - 1;
        for (size_t i = j + 1; i <= iend; ++i) {
            double a = L[i * ld + (i - j)];
            const size_t kk0 = (i > (size_t)bw) ? i - bw : 0;
            for (size_t k = kk0; k < j; ++k)
                a -= L[i * ld + (i - k)] * L[j * ld + (j - k)];
            L[i * ld + (i - j)] = a / d;
        }
    }
    s->chol = L;
    s->bottom_work = (double*)malloc(N * sizeof(double));
    s->chol_ready = 1;
}

static void orc_chol_solve(orc_solver* s, double* x)
{
    if (!s->chol_ready) orc_chol_factor(s);
    const int n = orc_n(s->cfg.coarsest_level);
    const int bw = n;
    const size_t N = (size_t)n * n;
    const size_t ld = (size_t)bw + 1;
    const double* L = s->chol;
    for (size_t i = 0; i < N; ++i) {                 /* L y = b */
        double a = x[i];
        const size_t k0 = (i > (size_t)bw) ? i - bw : 0;
        for (size_t k = k0; k < i; ++k) a -= L[i * ld + (i - k)] * x[k];
        x[i] = a / L[i * ld];
    }
    for (size_t ii = N; ii-- > 0;) {                 /* L^T x = y */
        double a = x[ii];
        const size_t kend = (ii + bw < N) ? ii + bw : N - 1;
        for (size_t k = ii + 1; k <= kend; ++k) a -= L[k * ld + (k - ii)] * x[k];
        x[ii] = a / L[ii * ld];
    }
}

/* ---- exact bottom solver, second method: type-I sine transform -------------
 * The 5-point Dirichlet Laplacian is diagonalised by S (symmetric, S S = I):
 * U = S ((S B S) ./ lambda) S, lambda_ij = s_i + s_j.  Mathematically the same x
 * as the Cholesky solve; numerically it differs from it in the last bits.  This
 * method performs the four n^3 products in double with one in-order accumulator
 * per output (k = 0, 1, ... n-1, product then sum, no contraction) - the operation
 * order of the device's bottom solver (csrc/mgx_bottom.hpp) - so that a float
 * hierarchy rounds THE SAME fp64 value on both sides and float / mixed residual
 * histories can be held to the fp64 tolerance.  Independent check of the method:
 * tests compare it with the Cholesky solve (<= 1e-11). */
static void orc_dst_init(orc_solver* s)
{
    const int n = orc_n(s->cfg.coarsest_level);
    const size_t nn = (size_t)n * n;
    s->dst_S = (double*)malloc(nn * sizeof(double));
    s->dst_s = (double*)malloc((size_t)n * sizeof(double));
    s->dst_w1 = (double*)malloc(nn * sizeof(double));
    s->dst_w2 = (double*)malloc(nn * sizeof(double));
    const long double pi = 3.14159265358979323846264338327950288L;
    const long double norm = sqrtl(2.0L / (long double)(n + 1));
    for (int i = 0; i < n; ++i) {
        const long double a = sinl(pi * (long double)(i + 1) / (2.0L * (long double)(n + 1)));
        s->dst_s[i] = (double)(4.0L * a * a);
        for (int k = 0; k < n; ++k) {
            /* argument reduced exactly before the sine */
            const long m = ((long)(i + 1) * (long)(k + 1)) % (2L * (n + 1));
            s->dst_S[(size_t)i * n + k] = (double)(norm * sinl(pi * (long double)m / (long double)(n + 1)));
        }
    }
}

/* C = A * B (n x n, row-major), in-order single accumulator; scale: C_ij /= (s_i + s_j) */
static void orc_dst_gemm(const double* A, const double* B, double* C, const double* sc, int n, int scale)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int k = 0; k < n; ++k) acc += A[(size_t)i * n + k] * B[(size_t)k * n + j];
            if (scale) acc = acc / (sc[i] + sc[j]);
            C[(size_t)i * n + j] = acc;
        }
}

/* x (n x n doubles) <- A^-1 x */
static void orc_dst_solve(orc_solver* s, double* x)
{
    if (!s->dst_S) orc_dst_init(s);
    const int n = orc_n(s->cfg.coarsest_level);
    orc_dst_gemm(s->dst_S, x, s->dst_w1, s->dst_s, n, 0);           /* S B          */
    orc_dst_gemm(s->dst_w1, s->dst_S, s->dst_w2, s->dst_s, n, 1);   /* (S B S)./lam */
    orc_dst_gemm(s->dst_S, s->dst_w2, s->dst_w1, s->dst_s, n, 0);   /* S (...)      */
    orc_dst_gemm(s->dst_w1, s->dst_S, x, s->dst_s, n, 0);           /* ... S        */
}

/* ---- type-generic operators and schedules ------------------------------ */
static void orc_smooth_var_f64(orc_solver* s, int level, double* v, const double* f, int mu);
static void orc_smooth_var_f32(orc_solver* s, int level, float* v, const float* f, int mu);
static void orc_residual_var_f64(orc_solver* s, int level, double* r, const double* v, const double* f);
static void orc_residual_var_f32(orc_solver* s, int level, float* r, const float* v, const float* f);
static void orc_bottom_var_f64(orc_solver* s, double* x, const double* rhs);
static void orc_bottom_var_f32(orc_solver* s, float* x, const float* rhs);
void orc_restrict_inject_f64(double* coarse, const double* fine, int nf, double weight);
void orc_restrict_inject_f32(float* coarse, const float* fine, int nf, double weight);

#define REAL double
#define SUF(x) x##_f64
#define ORC_FMA(a, b, c) fma((a), (b), (c))
#define ORC_FMA_HW(a, b, c) __builtin_fma((a), (b), (c))
#include "mg_oracle_impl.inc"
#include "mg_oracle_var.inc"
static void orc_smooth_var_f64(orc_solver* s, int level, double* v, const double* f, int mu) { var_smooth_f64(s, level, v, f, mu); }
static void orc_residual_var_f64(orc_solver* s, int level, double* r, const double* v, const double* f) { var_residual_f64(s, level, r, v, f); }
static void orc_bottom_var_f64(orc_solver* s, double* x, const double* rhs) { var_bottom_f64(s, x, rhs); }
#undef ORC_FMA
#undef ORC_FMA_HW
#undef REAL
#undef SUF

#define REAL float
#define SUF(x) x##_f32
#define ORC_FMA(a, b, c) fmaf((a), (b), (c))
#define ORC_FMA_HW(a, b, c) __builtin_fmaf((a), (b), (c))
#include "mg_oracle_impl.inc"
#include "mg_oracle_var.inc"
static void orc_smooth_var_f32(orc_solver* s, int level, float* v, const float* f, int mu) { var_smooth_f32(s, level, v, f, mu); }
static void orc_residual_var_f32(orc_solver* s, int level, float* r, const float* v, const float* f) { var_residual_f32(s, level, r, v, f); }
static void orc_bottom_var_f32(orc_solver* s, float* x, const float* rhs) { var_bottom_f32(s, x, rhs); }
#undef ORC_FMA
#undef ORC_FMA_HW
#undef REAL
#undef SUF

/* ---- solver object ------------------------------------------------------ */
static void* orc_zalloc(int level, size_t elem)
{
    const size_t n = (size_t)orc_n(level);
    return calloc(n * n, elem);
}

orc_solver* orc_create(const orc_config* cfg)
{
    if (!cfg || cfg->coarsest_level < 1 || cfg->finest_level < cfg->coarsest_level ||
        cfg->finest_level >= ORC_MAX_LEVELS - 1)
        return NULL;
    orc_solver* s = (orc_solver*)calloc(1, sizeof(orc_solver));
    s->cfg = *cfg;
    const int want64 = (cfg->dtype == ORC_DTYPE_F64 || cfg->dtype == ORC_DTYPE_MIXED);
    const int want32 = (cfg->dtype == ORC_DTYPE_F32 || cfg->dtype == ORC_DTYPE_MIXED);
    for (int l = cfg->coarsest_level; l <= cfg->finest_level; ++l) {
        const int full64 = (cfg->dtype == ORC_DTYPE_F64);
        if (want64 && (full64 || l == cfg->finest_level)) {
            s->res_f64[l] = orc_zalloc(l, sizeof(double));
        }
        if (full64) {
            s->tmp_f64[l] = orc_zalloc(l, sizeof(double));
            if (l < cfg->finest_level) {
                s->rhs_f64[l] = orc_zalloc(l, sizeof(double));
                s->sol_f64[l] = orc_zalloc(l, sizeof(double));
                s->fmg_rhs_f64[l] = orc_zalloc(l, sizeof(double));
                s->fmg_sol_f64[l] = orc_zalloc(l, sizeof(double));
            }
        }
        if (want32) {
            s->tmp_f32[l] = orc_zalloc(l, sizeof(float));
            s->res_f32[l] = orc_zalloc(l, sizeof(float));
            s->rhs_f32[l] = orc_zalloc(l, sizeof(float));
            s->sol_f32[l] = orc_zalloc(l, sizeof(float));
            if (l < cfg->finest_level) {
                s->fmg_rhs_f32[l] = orc_zalloc(l, sizeof(float));
                s->fmg_sol_f32[l] = orc_zalloc(l, sizeof(float));
            }
        }
    }
    return s;
}

void orc_destroy(orc_solver* s)
{
    if (!s) return;
    for (int l = 0; l < ORC_MAX_LEVELS; ++l) {
        free(s->tmp_f64[l]); free(s->res_f64[l]); free(s->rhs_f64[l]); free(s->sol_f64[l]);
        free(s->fmg_rhs_f64[l]); free(s->fmg_sol_f64[l]);
        free(s->tmp_f32[l]); free(s->res_f32[l]); free(s->rhs_f32[l]); free(s->sol_f32[l]);
        free(s->fmg_rhs_f32[l]); free(s->fmg_sol_f32[l]);
    }
    free(s->chol);
    free(s->bottom_work);
    free(s->dst_S); free(s->dst_s); free(s->dst_w1); free(s->dst_w2);
    for (int l = 0; l < ORC_MAX_LEVELS; ++l)
        for (int q = 0; q < 5; ++q) { free(s->var_a_f64[l][q]); free(s->var_j_f64[l][q]); free(s->var_a_f32[l][q]); free(s->var_j_f32[l][q]); }
    free(s->var_inv);
    free(s);
}

/* largest power of two <= x (x > 0): exact scaling for the fp32 inner solve */
static double orc_pow2_floor(double x)
{
    int e;
    (void)frexp(x, &e);      /* x = m 2^e, m in [0.5, 1) */
    return ldexp(1.0, e - 1);
}

/* PS:727 entry point, plus the residual history D10 asks for.
 * dtype f64: everything in double (MF's precision, D11).
 * dtype f32: everything in float (PS's precision).
 * dtype mixed (BASELINE config 5): double residual and solution; each cycle
 *   solves A e = r in float on r scaled by a power of two near its rms, then
 *   u += scale * e  (defect correction; the inner cycle starts from e = 0
 *   exactly as the coarse levels of PS:613 do). */
int orc_solve(orc_solver* s, const double* b, double* u, double tol, int max_cycles, double* hist)
{
    const int L = s->cfg.finest_level;
    const int n = orc_n(L);
    const size_t N = (size_t)n * n;
    const int fmg = (s->cfg.schedule == ORC_SCHEDULE_FMG);
    int k = 0;

    const int var = (s->cfg.op == ORC_OP_STENCIL5);
    if (var && s->cfg.dtype == ORC_DTYPE_MIXED) return -1;     /* general operators: f64 or f32 hierarchies */
#define ORC_RES64(r, u, b) do { if (var) orc_residual_var_f64(s, L, (r), (u), (b)); else orc_residual_f64((r), (u), (b), n); } while (0)
#define ORC_RES32(r, u, b) do { if (var) orc_residual_var_f32(s, L, (r), (u), (b)); else orc_residual_f32((r), (u), (b), n); } while (0)
    if (s->cfg.dtype == ORC_DTYPE_F64) {
        double* r = (double*)s->res_f64[L];
        ORC_RES64(r, u, b);
        hist[0] = orc_norm2_f64(r, N);
        for (k = 0; k < max_cycles; ++k) {
            if (hist[k] <= tol * hist[0]) break;
            if (k == 0 && fmg) orc_fmg_f64(s, L, u, b);
            else orc_vcycle_f64(s, L, u, b);
            ORC_RES64(r, u, b);
            hist[k + 1] = orc_norm2_f64(r, N);
        }
        return k;
    }

    if (s->cfg.dtype == ORC_DTYPE_F32) {
        float* b32 = (float*)s->rhs_f32[L];
        float* u32 = (float*)s->sol_f32[L];
        float* r = (float*)s->res_f32[L];
        for (size_t i = 0; i < N; ++i) { b32[i] = (float)b[i]; u32[i] = (float)u[i]; }
        ORC_RES32(r, u32, b32);
        hist[0] = orc_norm2_f32(r, N);
        for (k = 0; k < max_cycles; ++k) {
            if (hist[k] <= tol * hist[0]) break;
            if (k == 0 && fmg) orc_fmg_f32(s, L, u32, b32);
            else orc_vcycle_f32(s, L, u32, b32);
            ORC_RES32(r, u32, b32);
            hist[k + 1] = orc_norm2_f32(r, N);
        }
        for (size_t i = 0; i < N; ++i) u[i] = (double)u32[i];
        return k;
    }

    /* mixed */
    double* r = (double*)s->res_f64[L];
    float* r32 = (float*)s->rhs_f32[L];
    float* e32 = (float*)s->sol_f32[L];
    orc_residual_f64(r, u, b, n);
    hist[0] = orc_norm2_f64(r, N);
    for (k = 0; k < max_cycles; ++k) {
        if (hist[k] <= tol * hist[0]) break;
        if (k == 0 && fmg) {
            const double scale = orc_pow2_floor(orc_norm2_f64(b, N) / (double)n);
            const double inv = 1.0 / scale;
            for (size_t i = 0; i < N; ++i) r32[i] = (float)(b[i] * inv);
            orc_fmg_f32(s, L, e32, r32);
            for (size_t i = 0; i < N; ++i) u[i] = scale * (double)e32[i];
        } else {
            /* the scale lags one cycle (hist[k-1]) so that a device version can
             * write the float residual in the same pass that measures hist[k] */
            const double scale = orc_pow2_floor(hist[k > 0 ? k - 1 : 0] / (double)n);
            const double inv = 1.0 / scale;
            for (size_t i = 0; i < N; ++i) r32[i] = (float)(r[i] * inv);
            memset(e32, 0, N * sizeof(float));
            orc_vcycle_f32(s, L, e32, r32);
            for (size_t i = 0; i < N; ++i) u[i] = u[i] + scale * (double)e32[i];
        }
        orc_residual_f64(r, u, b, n);
        hist[k + 1] = orc_norm2_f64(r, N);
    }
    return k;
}

/* ---- problem data -------------------------------------------------------- */
/* PS:283-335 assembles b_i = f * (sum of element areas)/3 = f h^2 at every
 * interior node (six triangles of area h^2/2, a third each); sign per D1. */
void orc_rhs_constant(double* b, int level, double f)
{
    const int n = orc_n(level);
    const double h = 1.0 / (double)(1 << level);   /* PS:289 */
    const double val = f * h * h;
    for (size_t k = 0; k < (size_t)n * n; ++k) b[k] = val;
}

void orc_rhs_sine(double* b, int level)
{
    const int n = orc_n(level);
    const double h = 1.0 / (double)(1 << level);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < n; ++i) {
        const double sy = sin(2.0 * pi * (double)(i + 1) * h);
        for (int j = 0; j < n; ++j) {
            const double sx = sin(2.0 * pi * (double)(j + 1) * h);
            b[(size_t)i * n + j] = h * h * 8.0 * pi * pi * sx * sy;
        }
    }
}

/* MT19937-64 (Matsumoto & Nishimura 2004), the engine std::mt19937_64 names. */
typedef struct { uint64_t mt[312]; int idx; } orc_mt64;

static void orc_mt64_seed(orc_mt64* g, uint64_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 312; ++i)
        g->mt[i] = 6364136223846793005ULL * (g->mt[i - 1] ^ (g->mt[i - 1] >> 62)) + (uint64_t)i;
    g->idx = 312;
}

static uint64_t orc_mt64_next(orc_mt64* g)
{
    if (g->idx >= 312) {
        for (int i = 0; i < 312; ++i) {
            const uint64_t x = (g->mt[i] & 0xFFFFFFFF80000000ULL) | (g->mt[(i + 1) % 312] & 0x7FFFFFFFULL);
            uint64_t xa = x >> 1;
            if (x & 1ULL) xa ^= 0xB5026F5AA96619E9ULL;
            g->mt[i] = g->mt[(i + 156) % 312] ^ xa;
        }
        g->idx = 0;
    }
    uint64_t y = g->mt[g->idx++];
    y ^= (y >> 29) & 0x5555555555555555ULL;
    y ^= (y << 17) & 0x71D67FFFEDA60000ULL;
    y ^= (y << 37) & 0xFFF7EEE000000000ULL;
    y ^= (y >> 43);
    return y;
}

void orc_fill_uniform(double* u, size_t len, uint64_t seed)
{
    orc_mt64 g;
    orc_mt64_seed(&g, seed);
    for (size_t k = 0; k < len; ++k)
        u[k] = (double)(orc_mt64_next(&g) >> 11) * (1.0 / 4503599627370496.0) - 1.0;
}

/* ---- CPU baselines -------------------------------------------------------- */
static double orc_now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int orc_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* CSR of LU (the four -1 neighbours; boundary neighbours are eliminated,
 * PS:224) in the column order a sorted, merged coo_to_csr would emit. */
static void orc_build_lu_csr(int n, int32_t** indptr, int32_t** indices)
{
    const size_t N = (size_t)n * n;
    int32_t* ip = (int32_t*)malloc((N + 1) * sizeof(int32_t));
    int32_t* ix = (int32_t*)malloc(4 * N * sizeof(int32_t));
    size_t nnz = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const size_t k = (size_t)i * n + j;
            ip[k] = (int32_t)nnz;
            if (i > 0) ix[nnz++] = (int32_t)(k - n);
            if (j > 0) ix[nnz++] = (int32_t)(k - 1);
            if (j < n - 1) ix[nnz++] = (int32_t)(k + 1);
            if (i < n - 1) ix[nnz++] = (int32_t)(k + n);
        }
    ip[N] = (int32_t)nnz;
    *indptr = ip;
    *indices = ix;
}

#define ORC_DEFINE_CSR_BASELINE(NAME, REAL)                                              \
    double NAME(REAL* v, const REAL* f, int n, int mu, double omega)                     \
    {                                                                                    \
        const size_t N = (size_t)n * n;                                                  \
        int32_t *ip, *ix;                                                                \
        orc_build_lu_csr(n, &ip, &ix);                                                   \
        REAL* val = (REAL*)malloc(4 * N * sizeof(REAL));                                 \
        for (size_t k = 0; k < (size_t)ip[N]; ++k) val[k] = (REAL)-1;                    \
        REAL* ftmp = (REAL*)malloc(N * sizeof(REAL));                                    \
        REAL* lu = (REAL*)calloc(N, sizeof(REAL));                                       \
        REAL* first = (REAL*)calloc(N, sizeof(REAL));                                    \
        memcpy(ftmp, f, N * sizeof(REAL));                                /* PS:129 */   \
        const REAL om = (REAL)omega;                                                     \
        const REAL a_lu = (REAL)(-1.0 * (double)om / 4.0);                               \
        const REAL a_v = (REAL)(1.0 - (double)om);                                       \
        const REAL a_f = (REAL)((double)om / 4.0);                                       \
        const double t0 = orc_now();                                                     \
        for (int s = 0; s < mu; ++s) {                                                   \
            for (size_t r = 0; r < N; ++r) {                              /* PS:138 */   \
                REAL acc = 0;                                                            \
                for (int32_t p = ip[r]; p < ip[r + 1]; ++p) acc += val[p] * v[ix[p]];    \
                lu[r] = a_lu * acc;                                                      \
            }                                                                            \
            for (size_t r = 0; r < N; ++r) v[r] *= a_v;                   /* PS:139 */   \
            for (size_t r = 0; r < N; ++r) ftmp[r] *= a_f;                /* PS:140 */   \
            for (size_t r = 0; r < N; ++r) first[r] = v[r] + ftmp[r];     /* PS:141 */   \
            for (size_t r = 0; r < N; ++r) v[r] = first[r] + lu[r];       /* PS:142 */   \
            memcpy(ftmp, f, N * sizeof(REAL));                            /* PS:144 */   \
        }                                                                                \
        const double t1 = orc_now();                                                     \
        free(ip); free(ix); free(val); free(ftmp); free(lu); free(first);                \
        return t1 - t0;                                                                  \
    }

ORC_DEFINE_CSR_BASELINE(orc_baseline_csr_jacobi_f32, float)
ORC_DEFINE_CSR_BASELINE(orc_baseline_csr_jacobi_f64, double)

#define ORC_DEFINE_OMP_BASELINE(NAME, REAL)                                              \
    double NAME(REAL* v, const REAL* f, int n, int mu, double omega, int threads)        \
    {                                                                                    \
        const size_t N = (size_t)n * n;                                                  \
        REAL* tmp = (REAL*)malloc(N * sizeof(REAL));                                     \
        const REAL om = (REAL)omega;                                                     \
        const REAL c0 = (REAL)(1.0 - (double)om);                                        \
        const REAL c1 = (REAL)((double)om / 4.0);                                        \
        REAL* src = v;                                                                   \
        REAL* dst = tmp;                                                                 \
        if (threads < 1) threads = 1;                                                    \
        const double t0 = orc_now();                                                     \
        for (int s = 0; s < mu; ++s) {                                                   \
            _Pragma("omp parallel for num_threads(threads) schedule(static)")            \
            for (int i = 0; i < n; ++i) {                                                \
                const REAL* c = src + (size_t)i * n;                                     \
                const REAL* up = (i > 0) ? c - n : NULL;                                 \
                const REAL* dn = (i < n - 1) ? c + n : NULL;                             \
                const REAL* fr = f + (size_t)i * n;                                      \
                REAL* o = dst + (size_t)i * n;                                           \
                for (int j = 0; j < n; ++j) {                                            \
                    REAL sum = up ? up[j] : (REAL)0;                                     \
                    sum = sum + ((j > 0) ? c[j - 1] : (REAL)0);                          \
                    sum = sum + ((j < n - 1) ? c[j + 1] : (REAL)0);                      \
                    sum = sum + (dn ? dn[j] : (REAL)0);                                  \
                    o[j] = (c0 * c[j] + c1 * fr[j]) + c1 * sum;                          \
                }                                                                        \
            }                                                                            \
            REAL* sw = src; src = dst; dst = sw;                                         \
        }                                                                                \
        const double t1 = orc_now();                                                     \
        if (src != v) memcpy(v, src, N * sizeof(REAL));                                  \
        free(tmp);                                                                       \
        return t1 - t0;                                                                  \
    }

ORC_DEFINE_OMP_BASELINE(orc_baseline_omp_jacobi_f64, double)
ORC_DEFINE_OMP_BASELINE(orc_baseline_omp_jacobi_f32, float)
