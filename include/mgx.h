/*
 * mgx.h — C-ABI of libmgx: MI355X-native geometric multigrid for the 2-D
 * Poisson problem, the drop-in for the hot path of nikhilTkur/Multigrid_Nikhil_C-.
 *
 * The reference exposes no FFI: its interface is a set of C++ free functions
 * in one translation unit called from main().  Each entry point
 * below names the reference function or block it replaces
 * (PS = Poissons_SYCL.cpp, MF = Multigrid_functions.cpp).  A header-only C++
 * wrapper with the reference's own names and std::vector signatures is in
 * include/mgx_reference_api.hpp; INTEGRATION.md shows the binding a maintainer
 * would add.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types, never throws.
 *   - every function returns an mgx_status (0 = ok); mgx_last_error() gives text.
 *   - host vectors use the reference's layout: interior-only, row-major,
 *     n x n with n = 2^L - 1 (PS:227-233, PS:291, PS:662-664); element type is
 *     double for dtype F64 and MIXED, float for dtype F32.
 *   - device memory is owned by the handle; one host thread per handle; calls
 *     block until the result is available unless the name ends in _async.
 *   - there is no CPU fallback: creation fails with MGX_ERR_NO_DEVICE when no
 *     gfx950 device is usable.
 */
#ifndef MGX_H
#define MGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGX_API __attribute__((visibility("default")))

typedef enum {
    MGX_OK = 0,
    MGX_ERR_INVALID = 1,      /* bad argument / configuration */
    MGX_ERR_NO_DEVICE = 2,    /* no usable HIP device */
    MGX_ERR_HIP = 3,          /* a HIP runtime call failed */
    MGX_ERR_ALLOC = 4,
    MGX_ERR_STATE = 5         /* call not valid in the handle's current state */
} mgx_status;

enum { MGX_SMOOTHER_JACOBI = 0, MGX_SMOOTHER_RBGS = 1 };
enum { MGX_DTYPE_F32 = 0, MGX_DTYPE_F64 = 1, MGX_DTYPE_MIXED = 2 };
enum { MGX_SCHEDULE_V = 0, MGX_SCHEDULE_FMG = 1 };
enum { MGX_RESTRICT_CONSISTENT = 0, MGX_RESTRICT_FW16 = 1,
       MGX_RESTRICT_INJECT = 2,    /* MF:122-130 restriction2D as written: the fine value at the coincident node */
       MGX_RESTRICT_INJECT4 = 3 }; /* 4 x injection: the h^2-consistent weight for operators that carry no h^-2 (cf. D4) */
/* Operator of the hierarchy.  POISSON: the constant five-point stencil of PS (matrix-free, fused kernels).
 * STENCIL5: the data model of the reference's second draft (MF:16-41 ProblemVar) on the structured grid - one
 * general five-point operator PER LEVEL (A_sp_dict[level]) given by the caller as five coefficient grids
 * (mgx_set_stencil) or re-discretised from a nodal coefficient a(x, y) of -div(a grad u) (mgx_set_coefficient),
 * smoothed in MF's form  v <- R_omega v + omega D^-1 b  (MF:86-93), residual b - A v (MF:150-153), direct solve of
 * the coarsest operator (MF:63-72: dense inverse; coarsest_level <= 5).  dtype F64 / F32, Jacobi, one GPU.
 * Algorithmic bytes per point and sweep: 8 sizeof(T) (v, b, D_inv, four R arrays in; v' out). */
enum { MGX_OPERATOR_POISSON = 0, MGX_OPERATOR_STENCIL5 = 1 };
enum { MGX_BOTTOM_EXACT = 0, MGX_BOTTOM_SMOOTH = 1 };
/* Arithmetic of the weighted-Jacobi update  v' = (1-w) v + (w/4) b + (w/4)(N+W+E+S)  (PS:138-142):
 *   SEPARATE: the reference's five library calls as five roundings per point, in its order
 *             (t = c0 v + c1 b ; v' = t + c1 nb) - bit-identical to the CPU oracle's default mode;
 *   FMA:      the same expression with its two multiply-adds contracted, v' = fma(c1, nb, fma(c0, v, c1 b)):
 *             what oneMKL's fused kernels would be free to do; two roundings fewer per point, bit-identical
 *             to the oracle's FMA mode and within 1e-10 of the SEPARATE residual histories (north_star's
 *             tolerance; tests/test_gpu_fma.py).  Every other operator (residual, transfers, red-black
 *             Gauss-Seidel, bottom solve) performs the same operations in both modes. */
enum { MGX_ARITH_SEPARATE = 0, MGX_ARITH_FMA = 1 };
#define MGX_MAX_GPUS 16

/* The reference's compile-time globals as run-time fields.
 * PS:17-22 (finest_level, coarsest_level, mu0, mu1, mu2), PS:127 (omega);
 * MF:43-48 holds the draft's values of the same names. */
typedef struct {
    int finest_level;     /* PS:17  */
    int coarsest_level;   /* PS:18; >= 2, and <= 8 with bottom = EXACT (the dense sine-transform
                             solve holds (2^L - 1)^2 matrices: 255^2 at level 8; PS:18 uses 7) */
    int mu0;              /* PS:20  FMG runs mu0+1 V-cycles per level (PS:646) */
    int mu1;              /* PS:21  pre-smoothing sweeps  */
    int mu2;              /* PS:22  post-smoothing sweeps */
    double omega;         /* PS:127 Jacobi weight */
    int smoother;         /* MGX_SMOOTHER_*  (RBGS: BASELINE config 3) */
    int dtype;            /* MGX_DTYPE_*     (PS is f32, MF is f64; MIXED: config 5) */
    int schedule;         /* MGX_SCHEDULE_*  (PS:727 calls fullmultigrid = FMG) */
    int restrict_mode;    /* MGX_RESTRICT_*   */
    int bottom;           /* MGX_BOTTOM_*    (EXACT: MF:137-139; SMOOTH: PS:581-587, D8) */
    int device;           /* HIP device ordinal (n_gpus <= 1) */
    int profile;          /* 1: HIP events around each operator class, every launch eager (no graph replay);
                             2: events around the finest level's passes, launched one by one, and around ONE
                                graph replay of everything below it (what bench.py times: mgx_solve's own
                                execution, less the two kernel boundaries a whole-cycle graph saves) */
    /* ---- multi-GPU (the reference has one sycl::queue, PS:659) -------------
     * n_gpus > 1: the levels above cut_level are split into n_gpus row slabs, slab g on device
     * devices[g]; one stream pair per slab, halo rows moved device to device between the
     * smoothing blocks (hipMemcpyPeerAsync inside one process, RCCL send/recv between the
     * processes of mgx_create_rank), levels <= cut_level solved redundantly per device
     * ("the bottom solve stays on one GPU").  Several slabs may share a device (devices[g] all 0
     * runs the whole slab plan on one GPU: how the 1-GPU tests check n_gpus = 2, 4, 8 bit for bit
     * against n_gpus = 1).  Multi-GPU handles: dtype F64 or F32, both schedules. */
    int n_gpus;           /* 0 or 1: single GPU */
    int cut_level;        /* 0: chosen from the grid and n_gpus */
    int devices[MGX_MAX_GPUS];   /* -1 entries: slab g on device g modulo the device count */
    int arith;            /* MGX_ARITH_*     (PS:138-142: how the Jacobi update is rounded) */
    int op;               /* MGX_OPERATOR_*  (MF:16-41: per-level general operators) */
} mgx_config;

typedef struct mgx_solver* mgx_handle;

/* Fills the reference's defaults (PS:17-22, 127): levels 10..7, mu 30/10/10,
 * omega 2/3, Jacobi, FMG; dtype F64, consistent restriction, exact bottom. */
MGX_API int mgx_config_default(mgx_config* cfg);

/* Grid-hierarchy constructor: replaces the per-level loop of main()
 * (PS:661-690: globalstiffenssmatrix + coo_to_csr + init_matrix_handle +
 * set_csr_data filling jacobi_matrices[level - coarsest_level], PS:33).
 * Matrix-free: allocates per-level device arrays instead of CSR handles. */
MGX_API int mgx_create(const mgx_config* cfg, mgx_handle* out);
MGX_API int mgx_destroy(mgx_handle h);
MGX_API const char* mgx_last_error(mgx_handle h);   /* h may be NULL: last create error */
MGX_API const char* mgx_status_string(int status);

/* Interior points per side at `level`, n = 2^level - 1 (PS:662-664). */
MGX_API int mgx_level_n(int level);

/* ---- data in / out (host buffers, reference layout) ---------------------- */
/* which: vector selector for level-wise access */
enum { MGX_VEC_U = 0, MGX_VEC_B = 1, MGX_VEC_R = 2 };

/* Right-hand side of the finest level: replaces globalforcefunction()'s
 * output handed to fullmultigrid (PS:725-727).  count must be n*n. */
MGX_API int mgx_set_rhs(mgx_handle h, const void* b, size_t count);
/* Right-hand side with non-homogeneous Dirichlet data folded in (the general
 * input path of SURVEY §8f: the reference hard-wires u = 0 on the boundary by
 * eliminating the boundary nodes, PS:188-198, 224).  `ring` holds the boundary
 * node values g in the order: row 0 (columns 0..N), row N (columns 0..N),
 * column 0 (rows 1..N-1), column N (rows 1..N-1); ring_count = 4 N, N = n + 1.
 * Because every off-diagonal of A is -1, eliminating a boundary neighbour moves
 * +g to the right-hand side: b_ij += sum of its boundary neighbours' g.  The
 * solve then returns the interior of the solution of  A u = b,  u = g on the ring. */
MGX_API int mgx_set_rhs_dirichlet(mgx_handle h, const void* b, size_t count, const void* ring, size_t ring_count);
/* Initial guess / result of the finest level (PS:630 starts from zero). */
MGX_API int mgx_set_guess(mgx_handle h, const void* u, size_t count);
MGX_API int mgx_get_solution(mgx_handle h, void* u, size_t count);
/* Level-wise access for operator tests.  Element type: the level's working
 * type (float for F32; double for F64; for MIXED the finest-level U/B/R are
 * double and everything else float). */
MGX_API int mgx_set_level(mgx_handle h, int level, int which, const void* src, size_t count);
MGX_API int mgx_get_level(mgx_handle h, int level, int which, void* dst, size_t count);
/* Device-resident variants (inputs already in HBM): `grid` is a device pointer
 * to a whole level in the library's own layout, rows 0..N times
 * mgx_level_pitch(level, dtype) elements, Dirichlet ring and padding zero.
 * Device-to-device copies on the handle's stream; block until done. */
MGX_API int mgx_set_level_device(mgx_handle h, int level, int which, const void* grid);
MGX_API int mgx_get_level_device(mgx_handle h, int level, int which, void* grid);
/* zero a level vector (PS:613 / PS:630 initial guesses) */
MGX_API int mgx_zero_level(mgx_handle h, int level, int which);
/* Built-in right-hand sides, generated on the device:
 * kind 0: b = f h^2, the reference's load vector (PS:283-335, f = 4 at PS:123)
 * kind 1: b = h^2 8 pi^2 sin(2 pi x) sin(2 pi y)   (`f` ignored). */
MGX_API int mgx_fill_rhs(mgx_handle h, int kind, double f);
/* u ~ U(-1,1) from a counter-based generator keyed on (seed, index). */
MGX_API int mgx_fill_guess_random(mgx_handle h, uint64_t seed);

/* ---- general per-level operators: ProblemVar (MF:16-41), handles with op = MGX_OPERATOR_STENCIL5 -----------
 * A_sp_dict[level] (MF:19; csr_matrix_elements MF:33-41) as five coefficient grids in the host layout of every
 * vector (interior n x n, row-major, element type = the handle's working type):
 *     (A u)_ij = c u_ij + n u_(i-1)j + s u_(i+1)j + w u_i(j-1) + e u_i(j+1)
 * (coefficients that point at the eliminated Dirichlet ring are ignored, PS:188-198).  The call also builds
 * A_jacobi_sp_dict[level] = {D_inv, R_omega = I - omega D^-1 A} (MF:20, 28-32) and, on the coarsest level, the
 * direct solver of coarsest_level_matrix (MF:18, 63-72).  Every level must be given before a schedule runs
 * (MGX_ERR_STATE otherwise): as in MF, coarse operators are the caller's (the Python front-end of MF:1-3). */
MGX_API int mgx_set_stencil(mgx_handle h, int level, const void* c, const void* n, const void* s, const void* w,
                            const void* e, size_t count);
/* All levels at once from the nodal coefficient a >= a_min > 0 of  -div(a grad u)  on the finest grid's
 * (N + 1)^2 nodes (double, row-major, boundary nodes included): each level samples a at its own nodes and takes
 * face values as the mean of the two nodes (w = -(a_ij + a_i,j-1)/2 ..., c = -(n + s + w + e)): the re-discretised
 * hierarchy of MF:184.  a = 1 gives the Poisson stencil. */
MGX_API int mgx_set_coefficient(mgx_handle h, const double* a_nodes, size_t count);
/* read back: which = 0..4 the operator (c, n, s, w, e), 5..9 its Jacobi splitting (D_inv, R_n, R_s, R_w, R_e) */
MGX_API int mgx_get_stencil(mgx_handle h, int level, int which, void* dst, size_t count);

/* ---- grid operators (one call = the reference function named) ------------
 * On a dtype MIXED handle the finest level holds double data for the accessors above and a
 * float correction / residual pair for the inner cycle, so the operators and schedules below
 * return MGX_ERR_STATE when asked to act on the finest level (use mgx_solve there, or a F64 /
 * F32 handle); the coarser levels of a MIXED handle are ordinary float levels. */
/* jacobirelaxation(q, a_lu, size, v, f, mu)  PS:125-147 / MF:75-96; with
 * smoother = RBGS: mu red-black Gauss-Seidel sweeps.  Acts on U,B of `level`. */
MGX_API int mgx_smooth(mgx_handle h, int level, int mu);
/* residual block of vcyclemultigrid  PS:591-608 / MF:145-153:  R = B - A U. */
MGX_API int mgx_residual(mgx_handle h, int level);
/* restriction2d(residual)  PS:531-546, 611:  B[level-1] = R(B[level] - A U[level]),
 * residual fused in; also zeroes U[level-1] (PS:613). */
MGX_API int mgx_restrict(mgx_handle h, int level);
/* restriction2d(f_h)  PS:641: B[level-1] = R(B[level])  (FMG right-hand sides). */
MGX_API int mgx_restrict_rhs(mgx_handle h, int level);
/* interpolation2d + vm::add  PS:620-624:  U[level] += P U[level-1]. */
MGX_API int mgx_prolong_add(mgx_handle h, int level);
/* interpolation2d  PS:337-425, 645:  U[level] = P U[level-1]. */
MGX_API int mgx_prolong(mgx_handle h, int level);
/* coarsest-level solve U = A^-1 B  (MF:63-72 direct_solver, MF:137-139). */
MGX_API int mgx_bottom_solve(mgx_handle h);
/* ||B - A U||_2 on `level` (the report D10 says the reference lacks). */
MGX_API int mgx_residual_norm(mgx_handle h, int level, double* out);

/* ---- schedules ------------------------------------------------------------- */
/* vcyclemultigrid(q, a_h, vec_h, f_h) from `level` down  PS:575-627 / MF:132-173 */
MGX_API int mgx_vcycle(mgx_handle h, int level);
/* fullmultigrid(q, a_h, f_h)  PS:629-650 / MF:175-191 on the finest level. */
MGX_API int mgx_fmg(mgx_handle h);
/* One V-cycle from the finest level for A e = b starting from e = 0 (PS:613, 617: the
 * coarse-grid correction of a caller that owns the finer levels, e.g. the multi-GPU driver).
 * The zero guess is synthesised by the first smoothing pass where possible (nobody writes or
 * reads it), and with profiling off the cycle is replayed from a hipGraph. */
MGX_API int mgx_vcycle_zero(mgx_handle h);

typedef struct {
    int cycles;                 /* cycles run (an FMG pass counts as cycle 1) */
    int converged;              /* 1 if ||r|| <= tol ||r0|| */
    double initial_residual;    /* ||b - A u0||_2 */
    double final_residual;
    double seconds;             /* wall time of the solve, device-synchronised */
    double fine_updates;        /* finest-level smoother point updates performed */
    int history_len;            /* entries written to `history` (cycles + 1) */
} mgx_stats;

/* main()'s call `fullmultigrid(q, jacobi_matrices.back(), f_global)` PS:727 /
 * multigrid_solver(ProblemVar&) MF:193-197, run to a tolerance: cycles until
 * ||r||_2 <= tol ||r0||_2 or max_cycles.  history (may be NULL) receives
 * ||r||_2 before the first cycle and after each one; capacity history_cap. */
MGX_API int mgx_solve(mgx_handle h, double tol, int max_cycles, mgx_stats* stats,
                      double* history, int history_cap);

/* ---- measurement ------------------------------------------------------------ */
enum {
    MGX_PROF_SMOOTH_FINE = 0,   /* finest-level smoother launches */
    MGX_PROF_RESTRICT_FINE = 1, /* finest-level fused residual+restriction */
    MGX_PROF_PROLONG_FINE = 2,  /* finest-level prolongation+correction */
    MGX_PROF_NORM_FINE = 3,     /* finest-level residual norm */
    MGX_PROF_COARSE = 4,        /* everything below the finest level */
    MGX_PROF_COUNT = 5
};
typedef struct {
    double ms[MGX_PROF_COUNT];        /* accumulated HIP-event time */
    long long launches[MGX_PROF_COUNT]; /* kernel launches inside those intervals */
    long long sweeps[MGX_PROF_COUNT];   /* smoother sweeps those launches performed (a fused
                                           launch does several; 0 for non-smoother classes) */
} mgx_profile;
/* Valid when cfg.profile = 1 or 2; events are recorded on the handle's stream. */
MGX_API int mgx_profile_reset(mgx_handle h);
MGX_API int mgx_profile_get(mgx_handle h, mgx_profile* out);
/* `sweeps` finest-level smoother sweeps bracketed by HIP events on the
 * handle's stream; returns the elapsed milliseconds. */
MGX_API int mgx_time_smoother(mgx_handle h, int sweeps, double* ms);
MGX_API int mgx_synchronize(mgx_handle h);
/* Number of hipGraphs mgx_solve has captured for its loop body ("one V-cycle +
 * residual norm", PS:727 run to a tolerance): 0 until the first graph cycle,
 * -1 when graph replay is off (cfg.profile = 1, mixed precision, MGX_GRAPH=0,
 * or a failed capture). */
MGX_API int mgx_graphs_cached(mgx_handle h);

/* =============================================================================
 * Slab-level operators on caller-owned device memory.  These are what the
 * multi-GPU driver (one process per GPU, RCCL halo exchange between calls)
 * composes; the single-GPU handle above uses the same kernels.
 *
 * A slab is `rows` consecutive grid rows of one level stored with the level's
 * pitch (mgx_level_pitch), columns 0..N as in the handle (Dirichlet columns
 * and padding must be zero).  Row indices are local to the slab.  `stream` is
 * a hipStream_t passed as void* (NULL = default stream).  dtype is
 * MGX_DTYPE_F32 or MGX_DTYPE_F64.  Calls are asynchronous on `stream`.
 * ===========================================================================*/
typedef struct {
    int level;        /* grid level L: N = 2^L, columns 0..N */
    int dtype;        /* MGX_DTYPE_F32 / MGX_DTYPE_F64 */
    int rows;         /* rows allocated in the slab */
    int row0;         /* global row index of local row 0 */
    int arith;        /* MGX_ARITH_* of the smoother calls (mgx_config.arith) */
} mgx_slab;

/* elements per row for (level, dtype); bytes per row = pitch * sizeof(type) */
MGX_API long mgx_level_pitch(int level, int dtype);

/* mu Jacobi sweeps (PS:125-147) on local rows [row_lo,row_hi), ping-ponging
 * u <-> tmp; with shrink = 1 the range shrinks by one row per sweep at each
 * end that is not a global boundary (deep-halo communication avoidance): the
 * first sweep then covers [row_lo - (mu-1), row_hi + (mu-1)) clipped to the
 * unknown rows.  *result_in_tmp is set to 1 when mu is odd. */
MGX_API int mgx_slab_jacobi(const mgx_slab* s, void* u, const void* b, void* tmp,
                            int row_lo, int row_hi, int mu, double omega, int shrink,
                            int* result_in_tmp, void* stream);
/* same for red-black Gauss-Seidel; a sweep consumes two halo rows per side */
MGX_API int mgx_slab_rbgs(const mgx_slab* s, void* u, const void* b, void* tmp,
                          int row_lo, int row_hi, int mu, int shrink,
                          int* result_in_tmp, void* stream);
/* mu smoother sweeps (PS:125-147 / red-black GS) on local rows [row_lo,row_hi) with the
 * deep-halo shrinking of mgx_slab_jacobi(shrink = 1) and the V-cycle's transfer operators
 * folded into the passes, as the single-GPU handle does (one pass over the slab per 5 sweeps
 * instead of separate transfer kernels):
 *   coarse_e != NULL : the input is u + P e (PS:620-624), e on the coarse slab `c`;
 *   coarse_b != NULL : the residual of the result is restricted (PS:604-611) into coarse_b,
 *                      local coarse rows [crow_lo,crow_hi) of `c` (row_lo + row0 must be odd);
 *   sum_dev  != NULL : sum over [row_lo,row_hi) of (b - A u)^2 of the result -> *sum_dev
 *                      (scratch as for mgx_slab_residual_sumsq).
 * At most one of coarse_b / sum_dev.  The rows the passes read must hold valid data:
 * per*mu rows beyond the range, +2 with coarse_b, +1 with sum_dev (per = 1 Jacobi, 2 RB-GS),
 * and with coarse_e the coarse rows around them.
 * zero_in != 0 : the input iterate is known to be all zero (PS:613: the guess of a coarse-grid correction) and
 *                is NOT read - nobody has to write those zeros first; u is then scratch for the passes' ping-pong.
 *                Not with coarse_e; MGX_ERR_INVALID when the first pass cannot synthesise its input (mu = 1). */
MGX_API int mgx_slab_cycle(const mgx_slab* f, void* u, const void* b, void* tmp,
                           int row_lo, int row_hi, int mu, double omega, int smoother,
                           const mgx_slab* c, const void* coarse_e, void* coarse_b,
                           int crow_lo, int crow_hi, int restrict_mode, int zero_in,
                           double* scratch, double* sum_dev, int* result_in_tmp, void* stream);
/* fused residual + restriction (PS:604-611) of fine slab `f` into coarse slab
 * `c`, coarse local rows [crow_lo,crow_hi); zero_u (may be NULL) is the coarse
 * solution slab to zero on the same rows (PS:613). */
MGX_API int mgx_slab_restrict(const mgx_slab* f, const void* u, const void* b,
                              const mgx_slab* c, void* cb, void* zero_u,
                              int crow_lo, int crow_hi, int restrict_mode, int fused, void* stream);
/* u[fine rows row_lo..row_hi) (+)= P e  (PS:337-425, 620-624) */
MGX_API int mgx_slab_prolong(const mgx_slab* f, void* u, const mgx_slab* c, const void* e,
                             int row_lo, int row_hi, int add, void* stream);
/* sum over rows [row_lo,row_hi) of (b - A u)^2 -> *partial_sum_dev (device
 * double, written asynchronously); scratch must hold mgx_slab_scratch_doubles(). */
MGX_API int mgx_slab_residual_sumsq(const mgx_slab* s, const void* u, const void* b,
                                    int row_lo, int row_hi, double* scratch, double* sum_dev, void* stream);
MGX_API long mgx_slab_scratch_doubles(const mgx_slab* s);


/* =============================================================================
 * Multi-GPU: the slab plan as data, ranks, transports.
 *
 * mgx_create with cfg.n_gpus > 1 drives every slab from ONE process (one host thread, a stream
 * pair per device).  mgx_create_rank is the one-process-per-GPU form (torch.distributed.run /
 * mpirun launch `world` processes): this process owns slab `rank` and talks to its neighbours
 * through `transport` - NULL selects the built-in RCCL transport, which needs the 128-byte
 * ncclUniqueId of rank 0 (mgx_rccl_unique_id) handed to every rank by the launcher.
 * On such handles: mgx_set_rhs / mgx_set_guess / mgx_get_solution (whole-grid host vectors; each
 * slab takes / returns its rows), mgx_fill_rhs, mgx_fill_guess_random, mgx_residual_norm (finest
 * level), mgx_vcycle (finest level), mgx_fmg, mgx_solve, mgx_synchronize, mgx_destroy; everything
 * else returns MGX_ERR_STATE.
 * ===========================================================================*/
enum {
    MGX_DOP_EXCHANGE = 1,       /* fill `depth` halo rows of `which` (U / B) of `level` from both neighbours */
    MGX_DOP_ZERO_U = 2,         /* zero the coarse guess slab of `level` (PS:613) */
    MGX_DOP_CYCLE = 3,          /* mgx_slab_cycle: mu sweeps on [row_lo,row_hi) with folded transfers (pre / post) */
    MGX_DOP_SMOOTH = 4,         /* mgx_slab_jacobi / _rbgs, shrink = 1 */
    MGX_DOP_RESTRICT = 5,       /* mgx_slab_restrict, fused residual, coarse rows [crow_lo,crow_hi) */
    MGX_DOP_PROLONG = 6,        /* mgx_slab_prolong, add = 1, fine rows [row_lo,row_hi) */
    MGX_DOP_GATHER_CUT = 7,     /* all-gather the slabs' shares of the cut level's right-hand side */
    MGX_DOP_COARSE = 8,         /* one V-cycle from e = 0 on levels cut..coarsest (replicated) */
    MGX_DOP_SUMSQ = 9,          /* sum of (b - A u)^2 over rows [row_lo,row_hi) of the finest level */
    MGX_DOP_ALLREDUCE_NORM = 10,/* sum the slabs' sums of squares; the norm is its square root */
    /* fullmultigrid on slabs (PS:629-650) */
    MGX_DOP_RESTRICT_RHS = 11,  /* B[level-1] = R B[level] (PS:641), coarse rows [crow_lo,crow_hi) */
    MGX_DOP_PROLONG_SET = 12,   /* U[level] = P U[level-1] (PS:645) on fine rows [row_lo,row_hi) */
    MGX_DOP_COARSE_FMG = 13     /* fullmultigrid on levels cut..coarsest from the gathered right-hand side (replicated) */
};
typedef struct {
    int op, level;
    int which, depth;           /* EXCHANGE */
    int row_lo, row_hi, mu;     /* CYCLE / SMOOTH / PROLONG / SUMSQ: local rows of the slab of `level` */
    int pre;                    /* CYCLE: 1 = the input is u + P e (PS:620-624) */
    int post;                   /* CYCLE: 1 = restrict the residual of the result (PS:604-611), 2 = sum its squares */
    int crow_lo, crow_hi;       /* CYCLE post 1 / RESTRICT: local coarse rows produced */
    int coarse_is_cut;          /* the level below is the cut level (this slab's share / the gathered grid) */
} mgx_dist_op;
typedef struct {
    int level, N;               /* N = 2^level: rows 0..N */
    int own_lo, own_hi;         /* global rows owned: [own_lo, own_hi) */
    int halo;                   /* halo rows allocated per side */
    int row0, rows;             /* global row of local row 0, rows allocated */
} mgx_dist_level;

/* The plan of slab `g` of `n_slabs` for cfg (cfg.n_gpus is ignored, cfg.cut_level 0 = default):
 * pure host logic, no device needed - the CPU tests run these plans over numpy and gloo. */
typedef struct mgx_dist_planner* mgx_plan_handle;
MGX_API int mgx_plan_create(const mgx_config* cfg, int n_slabs, int g, int fold, int deep, mgx_plan_handle* out);
MGX_API int mgx_plan_destroy(mgx_plan_handle p);
MGX_API const char* mgx_plan_last_error(void);
MGX_API int mgx_plan_cut_level(mgx_plan_handle p);
/* geometry of the slab of `level` (cut < level <= finest) */
MGX_API int mgx_plan_level(mgx_plan_handle p, int level, mgx_dist_level* out);
/* this slab's share of the cut level: rows [row0, row0 + rows) */
MGX_API int mgx_plan_cut_share(mgx_plan_handle p, int* row0, int* rows);
/* the caller refilled u of the finest level: all rows incl. halos (all_rows = 1) or owned rows only */
MGX_API int mgx_plan_guess_set(mgx_plan_handle p, int all_rows);
/* operations of one V-cycle / of the residual norm, in order; returns the count (< 0: cap too small) */
MGX_API int mgx_plan_vcycle(mgx_plan_handle p, mgx_dist_op* ops, int cap);
MGX_API int mgx_plan_norm(mgx_plan_handle p, mgx_dist_op* ops, int cap);
/* operations of fullmultigrid (PS:629-650): right-hand sides restricted down to the cut level, FMG on
 * the replicated levels, then per slab level prolongation + mu0 + 1 V-cycles */
MGX_API int mgx_plan_fmg(mgx_plan_handle p, mgx_dist_op* ops, int cap);

/* what moves between slabs of different processes */
typedef struct { int send; int peer; void* ptr; size_t bytes; } mgx_xfer;   /* send = 1: ptr -> peer; 0: peer -> ptr */
typedef struct {
    void* ctx;
    /* every transfer of one halo exchange; device pointers; must be complete, or ordered on `stream`
     * (a hipStream_t), when the call returns */
    int (*sendrecv)(void* ctx, int n, const mgx_xfer* x, void* stream);
    /* recv[r * bytes .. ) = rank r's send[0 .. bytes); device pointers, same completion rule */
    int (*allgather)(void* ctx, const void* send, void* recv, size_t bytes, void* stream);
    /* *value = sum over ranks of *value (host double) */
    int (*allreduce_sum)(void* ctx, double* value);
} mgx_transport;
/* 128 bytes identifying a new RCCL communicator (ncclGetUniqueId); rank 0 calls it, the launcher
 * distributes the bytes */
MGX_API int mgx_rccl_unique_id(void* out128);
/* One rank of `world` (one process per GPU).  transport = NULL: built-in RCCL (rccl_id = the 128
 * bytes of mgx_rccl_unique_id from rank 0); otherwise the caller's transport (rccl_id ignored). */
MGX_API int mgx_create_rank(const mgx_config* cfg, int rank, int world, const void* rccl_id,
                            const mgx_transport* transport, mgx_handle* out);
/* number of halo exchanges a multi-GPU handle has performed (tests: communication plan) */
MGX_API long mgx_dist_exchanges(mgx_handle h);
/* how many of them ran on the slab's second stream beside the rows of the following smoothing pass that
 * need no halo (the pass then finishes with its two edge bands).  Opt-in, MGX_DIST_OVERLAP=1: a band is a
 * 10-level launch of ~50 us however thin it is, more than a halo message of this size takes */
MGX_API long mgx_dist_overlapped(mgx_handle h);
/* The ROCm runtime libraries this process has mapped (libamdhip64, libhsa-runtime64, librccl, libmgx ...), one
 * path per line, from /proc/self/maps.  libmgx is built against /opt/rocm (RUNPATH); a host application that
 * loaded another copy with the same SONAME first - the torch wheel bundles its own - would run this library on
 * that stack.  bench.py reports the list (`runtime_libs`); MGX_LOG_RUNTIME_LIBS=1 prints it at every
 * mgx_create / mgx_create_rank.  Returns the bytes written (without the terminator), < 0 on error. */
MGX_API int mgx_runtime_libs(char* buf, size_t cap);
/* plain copies for callers that implement a transport without a HIP binding of their own */
MGX_API int mgx_memcpy_d2h(void* dst_host, const void* src_dev, size_t bytes, void* stream);
MGX_API int mgx_memcpy_h2d(void* dst_dev, const void* src_host, size_t bytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MGX_H */
