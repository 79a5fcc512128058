// mgx_dist.hpp - the multi-GPU V-cycle: executor of the slab plans of mgx_dist_plan.hpp.
// Included by mgx.hip after the single-GPU solver (it uses cycle_body, the slab operators of the
// C-ABI and the fill kernels).
//
// Two shapes, one executor:
//   * mgx_create(cfg.n_gpus = P): ONE process drives all P slabs, slab g on device cfg.devices[g]
//     with its own stream; halo rows and the cut-level gather are device-to-device copies
//     (hipMemcpyPeerAsync) ordered by events - no host synchronisation inside a cycle.  Slabs may
//     share a device (the 1-GPU tests run P = 2, 4, 8 this way, bit for bit against P = 1).
//   * mgx_create_rank(rank, world): one process per GPU, this process owns slab `rank`; halos are
//     ncclSend / ncclRecv pairs grouped per exchange, the gather an ncclAllGather, the norm an
//     ncclAllReduce, all on the slab's stream (RCCL over xGMI) - or the caller's mgx_transport.
// The reference has none of this (one sycl::queue, PS:659); what runs on every slab is its V-cycle
// (PS:575-627) through the same kernels as on one GPU, hence bit-identical iterates.
#pragma once

#include <rccl/rccl.h>

namespace {

struct DistLevelBuf {
    void *u = nullptr, *b = nullptr, *tmp = nullptr;
    size_t bytes = 0;
    long pitch = 0;
};

struct DistSlab {
    int g = 0, device = 0;
    hipStream_t st = nullptr;                // compute stream: every slab operator
    hipStream_t cm = nullptr;                // second stream: a halo exchange and the edge bands that wait for
                                             // it, while `st` smooths the rows that need no halo (overlap)
    DistPlanner plan;
    std::vector<DistLevelBuf> lv;            // index = level - (cut + 1)
    void* c_own = nullptr;                   // this slab's rows of the cut level's right-hand side
    size_t c_own_bytes = 0;
    long c_pitch = 0;
    mgx_handle coarse = nullptr;             // levels coarsest..cut, replicated; its level-cut B / U are the
                                             // gathered right-hand side and the correction (no copies)
    double *scratch = nullptr, *sum_dev = nullptr, *sum_host = nullptr;
    hipEvent_t ev_ready = nullptr, ev_done = nullptr;
    std::vector<mgx_dist_op> ops;
    std::vector<char> zero_pending;          // per level: the guess is an implicit zero (PS:613) the next pre-smoothing pass synthesises
};

} // namespace

struct mgx_dist {
    mgx_config cfg{};
    int P = 1;                               // slabs in total
    int cut = 0;
    bool f64 = true;
    size_t es = 8;
    std::vector<DistSlab> slabs;             // the slabs this process drives (all P, or one)
    int rank = -1, world = 1;                // rank >= 0: one slab per process
    bool use_rccl = false;
    ncclComm_t comm = nullptr;
    bool comm_broken = false;                // an RCCL call failed: the communicator is aborted, not destroyed
    bool have_ext = false;
    mgx_transport ext{};
    long exchanges = 0;
    long overlapped = 0;                     // exchanges that ran beside the interior rows of the pass they feed
    int zero_in = 1;                         // MGX_ZERO_IN: coarse guesses are implicit zeros (no memset per level and cycle)
    int overlap = 0;                         // MGX_DIST_OVERLAP (off by default: DESIGN.md §7, the bands cost more than the exchange)
    double fine_updates = 0.0;
    // profiling of the finest-level smoothing blocks of the first local slab (cfg.profile)
    std::vector<EventPair> ev_used, ev_free;
    double prof_ms[MGX_PROF_COUNT] = {0};
    long long prof_launches[MGX_PROF_COUNT] = {0};
    long long prof_sweeps[MGX_PROF_COUNT] = {0};
};

namespace {

thread_local std::string g_plan_error;

#define DCHK(s, expr)                                                                    \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess)                                                           \
            return (s)->fail(MGX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__)); \
    } while (0)
#define NCHK(s, expr)                                                                    \
    do {                                                                                 \
        ncclResult_t r__ = (expr);                                                       \
        if (r__ != ncclSuccess) {                                                        \
            if ((s)->dist) (s)->dist->comm_broken = true;                                \
            return (s)->fail(MGX_ERR_HIP, std::string(#expr) + ": " + ncclGetErrorString(r__)); \
        }                                                                                \
    } while (0)

DistPlanCfg plan_cfg_of(const mgx_config& c, int P, int g, int cut, bool fold, bool deep)
{
    DistPlanCfg pc;
    pc.finest = c.finest_level; pc.cut = cut; pc.coarsest = std::min(c.coarsest_level, cut);
    pc.mu0 = c.mu0; pc.mu1 = c.mu1; pc.mu2 = c.mu2; pc.smoother = c.smoother; pc.P = P; pc.g = g; pc.fold = fold; pc.deep = deep;
    return pc;
}

int dist_cut_level(const mgx_config& c, int P)
{
    if (c.cut_level > 0) return c.cut_level;
    return DistPlanner::default_cut(c.finest_level, c.coarsest_level, P, c.mu1, c.mu2, c.smoother);
}

// ---- fills on a slab: the whole-grid fill kernels with a row window -----------------------------
template <typename T>
__global__ void k_fill_rhs_rows(T* b, int N, long pitch, int kind, double f, int row0, int rows)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int lr = blockIdx.y;
    const int r = row0 + lr;
    if (c > N || lr >= rows || r > N) return;
    const double h = 1.0 / (double)N;
    double v = 0.0;
    if (r >= 1 && r < N && c >= 1 && c < N) {
        if (kind == 0) v = f * h * h;                                   // PS:283-335 (sign: D1)
        else v = h * h * 8.0 * 9.869604401089358 * sinpi(2.0 * c * h) * sinpi(2.0 * r * h);
    }
    b[(long)lr * pitch + c] = (T)v;
}

template <typename T>
__global__ void k_fill_random_rows(T* u, int N, long pitch, uint64_t seed, int row0, int rows)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int lr = blockIdx.y;
    const int r = row0 + lr;
    if (c > N || lr >= rows || r > N) return;
    double v = 0.0;
    if (r >= 1 && r < N && c >= 1 && c < N) {
        const uint64_t idx = (uint64_t)(r - 1) * (uint64_t)(N - 1) + (uint64_t)(c - 1);  // PS:227 numbering
        const uint64_t x = splitmix64(seed ^ splitmix64(idx));
        v = (double)(x >> 11) * (1.0 / 4503599627370496.0) - 1.0;
    }
    u[(long)lr * pitch + c] = (T)v;
}

// ---- construction / destruction -------------------------------------------------------------------
void dist_free(mgx_dist* d)
{
    if (!d) return;
    for (auto& sl : d->slabs) {
        (void)hipSetDevice(sl.device);
        if (sl.st) (void)hipStreamSynchronize(sl.st);
        if (sl.coarse) {
            sl.coarse->stream = nullptr;               // the coarse handle borrowed the slab's stream
            mgx_destroy(sl.coarse);
        }
        for (auto& l : sl.lv)
            for (void* p : {l.u, l.b, l.tmp})
                if (p) (void)hipFree(p);
        if (sl.c_own) (void)hipFree(sl.c_own);
        if (sl.scratch) (void)hipFree(sl.scratch);
        if (sl.sum_dev) (void)hipFree(sl.sum_dev);
        if (sl.sum_host) (void)hipHostFree(sl.sum_host);
        if (sl.ev_ready) (void)hipEventDestroy(sl.ev_ready);
        if (sl.ev_done) (void)hipEventDestroy(sl.ev_done);
        if (sl.cm) { (void)hipStreamSynchronize(sl.cm); (void)hipStreamDestroy(sl.cm); }
        if (sl.st) (void)hipStreamDestroy(sl.st);
    }
    for (auto& p : d->ev_used) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto& p : d->ev_free) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    // ncclCommDestroy waits for the peers; after a failed collective they may never come: abort instead
    if (d->comm) (void)(d->comm_broken ? ncclCommAbort(d->comm) : ncclCommDestroy(d->comm));
    delete d;
}

int dist_alloc_slab(mgx_solver* s, mgx_dist* d, DistSlab& sl)
{
    DCHK(s, hipSetDevice(sl.device));
    DCHK(s, hipStreamCreate(&sl.st));
    DCHK(s, hipStreamCreate(&sl.cm));
    DCHK(s, hipEventCreateWithFlags(&sl.ev_ready, hipEventDisableTiming));
    DCHK(s, hipEventCreateWithFlags(&sl.ev_done, hipEventDisableTiming));
    const int dt = d->f64 ? MGX_DTYPE_F64 : MGX_DTYPE_F32;
    long scratch = 0;
    for (const mgx_dist_level& g : sl.plan.geom) {
        // (the level is registered before its arrays are allocated: dist_free then releases whatever a
        // failing hipMalloc left behind)
        sl.lv.emplace_back();
        DistLevelBuf& lb = sl.lv.back();
        lb.pitch = level_pitch(g.level, dt);
        lb.bytes = (size_t)g.rows * (size_t)lb.pitch * d->es;
        for (void** p : {&lb.u, &lb.b, &lb.tmp}) {
            if (hipMalloc(p, lb.bytes) != hipSuccess) { *p = nullptr; return s->fail(MGX_ERR_ALLOC, "hipMalloc failed for a slab level"); }
            DCHK(s, hipMemsetAsync(*p, 0, lb.bytes, sl.st));
        }
        const mgx_slab ms{g.level, dt, g.rows, g.row0, 0};
        scratch = std::max(scratch, mgx_slab_scratch_doubles(&ms));
    }
    sl.c_pitch = level_pitch(d->cut, dt);
    sl.c_own_bytes = (size_t)sl.plan.c_rows * (size_t)sl.c_pitch * d->es;
    if (hipMalloc(&sl.c_own, sl.c_own_bytes) != hipSuccess) return s->fail(MGX_ERR_ALLOC, "hipMalloc failed (cut level share)");
    DCHK(s, hipMemsetAsync(sl.c_own, 0, sl.c_own_bytes, sl.st));
    if (hipMalloc(&sl.scratch, (size_t)(scratch + 8) * sizeof(double)) != hipSuccess ||
        hipMalloc(&sl.sum_dev, sizeof(double)) != hipSuccess || hipHostMalloc(&sl.sum_host, sizeof(double)) != hipSuccess)
        return s->fail(MGX_ERR_ALLOC, "allocation of reduction buffers failed");
    // levels coarsest..cut: an ordinary single-GPU handle on this slab's device and STREAM
    mgx_config cc = d->cfg;
    cc.finest_level = d->cut;
    cc.coarsest_level = std::min(d->cfg.coarsest_level, d->cut);
    cc.schedule = MGX_SCHEDULE_V;
    cc.device = sl.device;
    cc.n_gpus = 0;
    cc.profile = 0;
    int rc = mgx_create(&cc, &sl.coarse);
    if (rc != MGX_OK) return s->fail(rc, std::string("coarse-level handle: ") + mgx_last_error(nullptr));
    DCHK(s, hipSetDevice(sl.device));
    (void)hipStreamSynchronize(sl.coarse->stream);
    (void)hipStreamDestroy(sl.coarse->stream);
    sl.coarse->stream = sl.st;
    return MGX_OK;
}

int dist_create(mgx_solver* s, const mgx_config* cfg, int rank, int world, const void* rccl_id, const mgx_transport* tr)
{
    const bool multi_process = rank >= 0;
    const int P = multi_process ? world : cfg->n_gpus;
    if (P < 1 || P > MGX_MAX_GPUS) return s->fail(MGX_ERR_INVALID, "number of slabs out of range");
    if (cfg->dtype == MGX_DTYPE_MIXED) return s->fail(MGX_ERR_INVALID, "multi-GPU handles support dtype F64 and F32");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return s->fail(MGX_ERR_NO_DEVICE, "no usable HIP device (libmgx has no CPU fallback)");
    mgx_dist* d = new (std::nothrow) mgx_dist();
    if (!d) return s->fail(MGX_ERR_ALLOC, "out of host memory");
    s->dist = d;
    d->cfg = *cfg;
    d->P = P;
    d->f64 = (cfg->dtype == MGX_DTYPE_F64);
    d->es = d->f64 ? 8 : 4;
    d->rank = rank; d->world = world;
    d->cut = dist_cut_level(*cfg, P);
    if (d->cut >= cfg->finest_level || d->cut < 2)
        return s->fail(MGX_ERR_INVALID, "no level above cut_level to distribute (use a single-GPU handle)");
    const bool fold = env_int("MGX_DIST_FOLD", 1) != 0, deep = env_int("MGX_DIST_DEEP", 1) != 0;
    d->overlap = env_int("MGX_DIST_OVERLAP", 0);
    d->zero_in = env_int("MGX_ZERO_IN", 1);
    const int first = multi_process ? rank : 0, count = multi_process ? 1 : P;
    d->slabs.resize(count);
    for (int i = 0; i < count; ++i) {
        DistSlab& sl = d->slabs[i];
        sl.g = first + i;
        int dev = multi_process ? cfg->device : cfg->devices[sl.g];
        if (dev < 0) dev = sl.g % ndev;
        if (dev >= ndev) return s->fail(MGX_ERR_INVALID, "device ordinal out of range");
        sl.device = dev;
        if (sl.plan.init(plan_cfg_of(*cfg, P, sl.g, d->cut, fold, deep)) != MGX_OK) return s->fail(MGX_ERR_INVALID, sl.plan.err);
        sl.zero_pending.assign(cfg->finest_level - d->cut, 0);
    }
    for (auto& sl : d->slabs) {
        int rc = dist_alloc_slab(s, d, sl);
        if (rc) return rc;
    }
    // direct device-to-device copies between the slabs' devices where the hardware allows
    if (!multi_process)
        for (auto& a : d->slabs)
            for (auto& b : d->slabs)
                if (a.device != b.device) {
                    int can = 0;
                    if (hipDeviceCanAccessPeer(&can, a.device, b.device) == hipSuccess && can) {
                        (void)hipSetDevice(a.device);
                        (void)hipDeviceEnablePeerAccess(b.device, 0);
                        (void)hipGetLastError();      // already enabled is not an error
                    }
                }
    if (multi_process && (world > 1 || rccl_id)) {
        if (tr) { d->have_ext = true; d->ext = *tr; }
        else {
            if (!rccl_id) return s->fail(MGX_ERR_INVALID, "the built-in RCCL transport needs the ncclUniqueId of rank 0");
            ncclUniqueId id;
            std::memcpy(&id, rccl_id, sizeof(id) < 128 ? sizeof(id) : 128);
            DCHK(s, hipSetDevice(d->slabs[0].device));
            // RCCL checks hipGetLastError() after its own launches: an error some EARLIER call of this thread left
            // behind (the host application's, or a refused call of ours) must not be taken for RCCL's
            (void)hipGetLastError();
            NCHK(s, ncclCommInitRank(&d->comm, world, id, rank));
            d->use_rccl = true;
        }
    }
    for (auto& sl : d->slabs) {
        DCHK(s, hipSetDevice(sl.device));
        DCHK(s, hipStreamSynchronize(sl.st));
    }
    return MGX_OK;
}

// ---- pieces of the executor ------------------------------------------------------------------------
inline mgx_slab slab_of(const mgx_dist* d, const mgx_dist_level& g)
{
    return mgx_slab{g.level, d->f64 ? MGX_DTYPE_F64 : MGX_DTYPE_F32, g.rows, g.row0, d->cfg.arith};
}
inline DistLevelBuf& buf(DistSlab& sl, int level) { return sl.lv[level - (sl.plan.c.cut + 1)]; }

// rows [r, r + depth) of a slab tensor
inline char* rows_ptr(const mgx_dist* d, void* base, long pitch, int r) { return (char*)base + (size_t)r * (size_t)pitch * d->es; }

// device-to-device copy on `st` (a stream of dst_dev): over xGMI between two devices, plain D2D on one
inline hipError_t copy_between(void* dst, int dst_dev, const void* src, int src_dev, size_t bytes, hipStream_t st)
{
    if (dst_dev == src_dev) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, st);
    return hipMemcpyPeerAsync(dst, dst_dev, src, src_dev, bytes, st);
}

// on_comm: run the exchange on the slabs' second stream.  after(slab, stream): what to enqueue on that
// stream once the slab's halo rows are in (the edge bands of an overlapped smoothing pass);
// meanwhile(slab): what the compute stream does in the meantime (the rows that need no halo), enqueued
// before the compute stream joins the exchange.
template <typename After, typename Meanwhile>
int dist_exchange(mgx_solver* s, mgx_dist* d, const mgx_dist_op& o, bool on_comm, After after, Meanwhile meanwhile)
{
    const int P = d->P;
    const size_t es = d->es;
    auto xs = [&](DistSlab& sl) { return on_comm ? sl.cm : sl.st; };
    if (d->rank < 0) {
        // one process: pull the neighbours' edge rows with device-to-device copies on the receiver's
        // stream; events order them after the producers and before the next writers of the source rows
        for (auto& sl : d->slabs) { DCHK(s, hipSetDevice(sl.device)); DCHK(s, hipEventRecord(sl.ev_ready, sl.st)); }
        for (auto& sl : d->slabs) {
            const mgx_dist_level& g = sl.plan.L(o.level);
            DistLevelBuf& lb = buf(sl, o.level);
            void* t = (o.which == MGX_VEC_U) ? lb.u : lb.b;
            const size_t bytes = (size_t)o.depth * (size_t)lb.pitch * es;
            const int lo = g.own_lo - g.row0;                       // lower interior slab edge (local row)
            const int up = (sl.g + 1) * (g.N / P) - g.row0;         // upper interior slab edge (local row)
            DCHK(s, hipSetDevice(sl.device));
            if (on_comm) DCHK(s, hipStreamWaitEvent(sl.cm, sl.ev_ready, 0));
            if (sl.g > 0) {
                DistSlab& nb = d->slabs[sl.g - 1];
                const mgx_dist_level& ng = nb.plan.L(o.level);
                DistLevelBuf& nlb = buf(nb, o.level);
                void* nt = (o.which == MGX_VEC_U) ? nlb.u : nlb.b;
                const int nup = sl.g * (ng.N / P) - ng.row0;        // the neighbour's upper edge
                DCHK(s, hipStreamWaitEvent(xs(sl), nb.ev_ready, 0));
                DCHK(s, copy_between(rows_ptr(d, t, lb.pitch, lo - o.depth), sl.device,
                                     rows_ptr(d, nt, nlb.pitch, nup - o.depth), nb.device, bytes, xs(sl)));
            }
            if (sl.g < P - 1) {
                DistSlab& nb = d->slabs[sl.g + 1];
                const mgx_dist_level& ng = nb.plan.L(o.level);
                DistLevelBuf& nlb = buf(nb, o.level);
                void* nt = (o.which == MGX_VEC_U) ? nlb.u : nlb.b;
                const int nlo = ng.own_lo - ng.row0;
                DCHK(s, hipStreamWaitEvent(xs(sl), nb.ev_ready, 0));
                DCHK(s, copy_between(rows_ptr(d, t, lb.pitch, up), sl.device,
                                     rows_ptr(d, nt, nlb.pitch, nlo), nb.device, bytes, xs(sl)));
            }
            const int rc = after(sl, xs(sl));
            if (rc != MGX_OK) return rc;
            DCHK(s, hipEventRecord(sl.ev_done, xs(sl)));
        }
        for (auto& sl : d->slabs) {
            const int rc = meanwhile(sl);
            if (rc != MGX_OK) return rc;
        }
        // nobody overwrites rows a neighbour is still reading (and the compute stream joins its own
        // second stream)
        for (auto& sl : d->slabs) {
            DCHK(s, hipSetDevice(sl.device));
            if (on_comm) DCHK(s, hipStreamWaitEvent(sl.st, sl.ev_done, 0));
            if (sl.g > 0) DCHK(s, hipStreamWaitEvent(sl.st, d->slabs[sl.g - 1].ev_done, 0));
            if (sl.g < P - 1) DCHK(s, hipStreamWaitEvent(sl.st, d->slabs[sl.g + 1].ev_done, 0));
        }
    } else {
        DistSlab& sl = d->slabs[0];
        const mgx_dist_level& g = sl.plan.L(o.level);
        DistLevelBuf& lb = buf(sl, o.level);
        void* t = (o.which == MGX_VEC_U) ? lb.u : lb.b;
        const size_t bytes = (size_t)o.depth * (size_t)lb.pitch * es;
        const int lo = g.own_lo - g.row0, up = (sl.g + 1) * (g.N / P) - g.row0;
        mgx_xfer x[4];
        int n = 0;
        if (sl.g > 0) {
            x[n++] = mgx_xfer{1, sl.g - 1, rows_ptr(d, t, lb.pitch, lo), bytes};
            x[n++] = mgx_xfer{0, sl.g - 1, rows_ptr(d, t, lb.pitch, lo - o.depth), bytes};
        }
        if (sl.g < P - 1) {
            x[n++] = mgx_xfer{1, sl.g + 1, rows_ptr(d, t, lb.pitch, up - o.depth), bytes};
            x[n++] = mgx_xfer{0, sl.g + 1, rows_ptr(d, t, lb.pitch, up), bytes};
        }
        DCHK(s, hipSetDevice(sl.device));
        if (on_comm) {
            DCHK(s, hipEventRecord(sl.ev_ready, sl.st));
            DCHK(s, hipStreamWaitEvent(sl.cm, sl.ev_ready, 0));
        }
        if (d->use_rccl) {
            // one group per exchange: the sends and receives of both neighbours progress together
            NCHK(s, ncclGroupStart());
            ncclResult_t gr = ncclSuccess;
            const char* what = "";
            for (int i = 0; i < n && gr == ncclSuccess; ++i) {
                if (x[i].send) { gr = ncclSend(x[i].ptr, x[i].bytes, ncclChar, x[i].peer, d->comm, xs(sl)); what = "ncclSend"; }
                else { gr = ncclRecv(x[i].ptr, x[i].bytes, ncclChar, x[i].peer, d->comm, xs(sl)); what = "ncclRecv"; }
            }
            // the group is closed whatever happened inside it (a return between Start and End would leave
            // every later RCCL call of this thread inside an open group)
            const ncclResult_t ge = ncclGroupEnd();
            if (gr != ncclSuccess) { d->comm_broken = true; return s->fail(MGX_ERR_HIP, std::string(what) + ": " + ncclGetErrorString(gr)); }
            if (ge != ncclSuccess) { d->comm_broken = true; return s->fail(MGX_ERR_HIP, std::string("ncclGroupEnd: ") + ncclGetErrorString(ge)); }
        } else if (d->have_ext) {
            if (d->ext.sendrecv(d->ext.ctx, n, x, (void*)xs(sl)) != 0) return s->fail(MGX_ERR_HIP, "transport sendrecv failed");
        }
        int rc = after(sl, xs(sl));
        if (rc != MGX_OK) return rc;
        if (on_comm) DCHK(s, hipEventRecord(sl.ev_done, sl.cm));
        if ((rc = meanwhile(sl)) != MGX_OK) return rc;
        if (on_comm) DCHK(s, hipStreamWaitEvent(sl.st, sl.ev_done, 0));
    }
    d->exchanges += 1;
    return MGX_OK;
}

int dist_gather_cut(mgx_solver* s, mgx_dist* d)
{
    const int cut = d->cut;
    if (d->rank < 0) {
        for (auto& sl : d->slabs) { DCHK(s, hipSetDevice(sl.device)); DCHK(s, hipEventRecord(sl.ev_ready, sl.st)); }
        for (auto& sl : d->slabs) {
            DCHK(s, hipSetDevice(sl.device));
            char* dst = (char*)sl.coarse->lv[cut].b;
            for (auto& src : d->slabs) {
                if (src.g != sl.g) DCHK(s, hipStreamWaitEvent(sl.st, src.ev_ready, 0));
                DCHK(s, copy_between(dst + (size_t)src.g * src.c_own_bytes, sl.device, src.c_own, src.device,
                                     src.c_own_bytes, sl.st));
            }
            DCHK(s, hipEventRecord(sl.ev_done, sl.st));
        }
        for (auto& sl : d->slabs) {
            DCHK(s, hipSetDevice(sl.device));
            for (auto& other : d->slabs)
                if (other.g != sl.g) DCHK(s, hipStreamWaitEvent(sl.st, other.ev_done, 0));
        }
        return MGX_OK;
    }
    DistSlab& sl = d->slabs[0];
    DCHK(s, hipSetDevice(sl.device));
    void* dst = sl.coarse->lv[cut].b;
    if (d->use_rccl) {
        NCHK(s, ncclAllGather(sl.c_own, dst, sl.c_own_bytes, ncclChar, d->comm, sl.st));
    } else if (d->world == 1) {
        DCHK(s, hipMemcpyAsync(dst, sl.c_own, sl.c_own_bytes, hipMemcpyDeviceToDevice, sl.st));
    } else if (d->have_ext) {
        if (d->ext.allgather(d->ext.ctx, sl.c_own, dst, sl.c_own_bytes, (void*)sl.st) != 0)
            return s->fail(MGX_ERR_HIP, "transport allgather failed");
    }
    return MGX_OK;
}

struct DistProf {
    mgx_dist* d; int idx = -1;
    DistProf(mgx_dist* d_, DistSlab& sl, bool on, int cls, long long sweeps) : d(d_)
    {
        if (!on) return;
        EventPair p;
        if (!d->ev_free.empty()) { p = d->ev_free.back(); d->ev_free.pop_back(); }
        else { if (hipEventCreate(&p.a) != hipSuccess || hipEventCreate(&p.b) != hipSuccess) return; }
        p.cls = cls; p.launches = 0; p.sweeps = sweeps;
        (void)hipEventRecord(p.a, sl.st);
        d->ev_used.push_back(p);
        idx = (int)d->ev_used.size() - 1;
        st = sl.st;
    }
    void launches(long long n) { if (idx >= 0) d->ev_used[idx].launches = n; }
    ~DistProf() { if (idx >= 0) (void)hipEventRecord(d->ev_used[idx].b, st); }
    hipStream_t st = nullptr;
};

// passes a folded block of mu sweeps is split into (what mgx_slab_cycle will launch)
int dist_block_launches(const mgx_dist* d, int N, int mu, int post, bool pre = false)
{
    int parts[64];
    FuseCfg fc = fuse_cfg();
    fc.arith = d->cfg.arith;
    return plan_folded(fc, d->cfg.smoother, N, mu, post, d->f64, parts, pre);
}

// the CYCLE operation of a plan on local rows [row_lo,row_hi) of its range, on `stream` (the whole
// range normally; the interior rows and the two edge bands separately when a halo exchange overlaps it)
int dist_cycle_rows(mgx_solver* s, mgx_dist* d, DistSlab& sl, const mgx_dist_op& o, int row_lo, int row_hi, hipStream_t stream,
                    int* flag, int zero_in = 0)
{
    const int cut = d->cut;
    const int dt = d->f64 ? MGX_DTYPE_F64 : MGX_DTYPE_F32;
    const mgx_dist_level& g = sl.plan.L(o.level);
    DistLevelBuf& lb = buf(sl, o.level);
    const mgx_slab fs = slab_of(d, g);
    mgx_slab cs{};
    const void* ce = nullptr;
    void* cb = nullptr;
    if (o.pre) {
        if (o.coarse_is_cut) { cs = mgx_slab{cut, dt, (1 << cut) + 1, 0, d->cfg.arith}; ce = sl.coarse->lv[cut].u; }
        else { cs = slab_of(d, sl.plan.L(o.level - 1)); ce = buf(sl, o.level - 1).u; }
    }
    if (o.post == 1) {
        if (o.coarse_is_cut) { cs = mgx_slab{cut, dt, sl.plan.c_rows, sl.plan.c_row0, d->cfg.arith}; cb = sl.c_own; }
        else { cs = slab_of(d, sl.plan.L(o.level - 1)); cb = buf(sl, o.level - 1).b; }
    }
    const int rc = mgx_slab_cycle(&fs, lb.u, lb.b, lb.tmp, row_lo, row_hi, o.mu, d->cfg.omega, d->cfg.smoother,
                                  (o.pre || o.post == 1) ? &cs : nullptr, ce, cb, o.crow_lo, o.crow_hi, d->cfg.restrict_mode, zero_in,
                                  o.post == 2 ? sl.scratch : nullptr, o.post == 2 ? sl.sum_dev : nullptr, flag, (void*)stream);
    if (rc != MGX_OK) return s->fail(rc, "mgx_slab_cycle failed on a slab");
    return MGX_OK;
}

// An exchange followed (after the zero fills of coarser guesses) by the pre-smoothing block of the same
// level that is ONE pass: the rows whose dependency cone touches no halo row are smoothed on the compute
// stream while the halos travel; the two edge bands follow the halos on the second stream.  Three
// launches of the same kernel on disjoint row ranges reading the same input: same bits as one launch.
// Returns the index of the CYCLE operation, or -1 when the pattern / the pass count does not fit.
int dist_overlap_candidate(const mgx_dist* d, const DistSlab& sl, size_t i)
{
    const std::vector<mgx_dist_op>& ops = sl.ops;
    const mgx_dist_op& x = ops[i];
    size_t j = i + 1;
    while (j < ops.size() && ops[j].op == MGX_DOP_ZERO_U) ++j;
    if (j >= ops.size()) return -1;
    const mgx_dist_op& c = ops[j];
    if (c.op != MGX_DOP_CYCLE || c.level != x.level || c.pre != 0 || c.post != 1) return -1;
    if (dist_block_launches(d, 1 << c.level, c.mu, c.post) != 1) return -1;     // several passes ping-pong the rows the bands rewrite
    return (int)j;
}

// interior rows [*lo,*hi) of the CYCLE range of slab `sl`: outputs that depend on no halo row being
// exchanged; both ends on odd global rows (the folded restriction wants its ranges to start there)
bool dist_interior_rows(const mgx_dist* d, const DistSlab& sl, const mgx_dist_op& c, int* lo, int* hi)
{
    const mgx_dist_level& g = sl.plan.L(c.level);
    const int per = d->cfg.smoother == MGX_SMOOTHER_RBGS ? 2 : 1;
    const int D = per * c.mu + 3;                                   // sweeps + the residual / restriction rows
    const int own_lo = g.own_lo - g.row0, own_hi = (sl.g + 1) * (g.N / d->P) - g.row0;    // interior slab edges (local)
    int a = c.row_lo, b = c.row_hi;
    if (sl.g > 0) { a = own_lo + D; if (((a + g.row0) & 1) == 0) ++a; }
    if (sl.g < d->P - 1) { b = own_hi - D; if (((b + g.row0) & 1) == 0) --b; }
    *lo = a; *hi = b;
    return b - a >= 64;                                            // worth three launches
}

int dist_local_op(mgx_solver* s, mgx_dist* d, DistSlab& sl, const mgx_dist_op& o)
{
    DCHK(s, hipSetDevice(sl.device));
    const int cut = d->cut;
    const int dt = d->f64 ? MGX_DTYPE_F64 : MGX_DTYPE_F32;
    const bool timed = d->cfg.profile && (&sl == &d->slabs[0]) && o.level == d->cfg.finest_level;
    switch (o.op) {
        case MGX_DOP_ZERO_U: {
            // PS:613.  The zeros are synthesised by the pre-smoothing pass that consumes them (nobody writes or
            // reads them: the memsets were 79 us of a 870 us cycle per GPU at 8 slabs of 16384^2) when that pass is
            // the next operation on this level's u and is a fused / folded kernel; otherwise they are written now.
            DistLevelBuf& lb = buf(sl, o.level);
            const size_t me = (size_t)(&o - sl.ops.data());
            bool implicit = d->zero_in != 0 && me < sl.ops.size();
            if (implicit) {
                implicit = false;
                for (size_t k = me + 1; k < sl.ops.size(); ++k) {
                    const mgx_dist_op& n = sl.ops[k];
                    if (n.level != o.level) continue;
                    if (n.op == MGX_DOP_EXCHANGE && n.which == MGX_VEC_B) continue;          // the right-hand side's halos
                    implicit = n.op == MGX_DOP_CYCLE && n.pre == 0 && (n.mu >= 2 || d->cfg.smoother == MGX_SMOOTHER_RBGS);
                    break;
                }
            }
            if (implicit) sl.zero_pending[o.level - (cut + 1)] = 1;
            else DCHK(s, hipMemsetAsync(lb.u, 0, lb.bytes, sl.st));
            return MGX_OK;
        }
        case MGX_DOP_CYCLE: {
            DistProf pr(d, sl, timed, MGX_PROF_SMOOTH_FINE, o.mu);
            pr.launches(dist_block_launches(d, sl.plan.L(o.level).N, o.mu, o.post, o.pre != 0));
            int flag = 0;
            const int zin = sl.zero_pending[o.level - (cut + 1)];
            sl.zero_pending[o.level - (cut + 1)] = 0;
            const int rc = dist_cycle_rows(s, d, sl, o, o.row_lo, o.row_hi, sl.st, &flag, zin);
            if (rc != MGX_OK) return rc;
            if (flag) std::swap(buf(sl, o.level).u, buf(sl, o.level).tmp);
            return MGX_OK;
        }
        case MGX_DOP_SMOOTH: {
            const mgx_dist_level& g = sl.plan.L(o.level);
            DistLevelBuf& lb = buf(sl, o.level);
            const mgx_slab fs = slab_of(d, g);
            int flag = 0;
            DistProf pr(d, sl, timed, MGX_PROF_SMOOTH_FINE, o.mu);
            pr.launches(o.mu);
            const int rc = d->cfg.smoother == MGX_SMOOTHER_RBGS
                ? mgx_slab_rbgs(&fs, lb.u, lb.b, lb.tmp, o.row_lo, o.row_hi, o.mu, 1, &flag, (void*)sl.st)
                : mgx_slab_jacobi(&fs, lb.u, lb.b, lb.tmp, o.row_lo, o.row_hi, o.mu, d->cfg.omega, 1, &flag, (void*)sl.st);
            if (rc != MGX_OK) return s->fail(rc, "slab smoother failed");
            if (flag) std::swap(lb.u, lb.tmp);
            return MGX_OK;
        }
        case MGX_DOP_RESTRICT: {
            const mgx_dist_level& g = sl.plan.L(o.level);
            DistLevelBuf& lb = buf(sl, o.level);
            const mgx_slab fs = slab_of(d, g);
            mgx_slab cs;
            void* cb;
            if (o.coarse_is_cut) { cs = mgx_slab{cut, dt, sl.plan.c_rows, sl.plan.c_row0, d->cfg.arith}; cb = sl.c_own; }
            else { cs = slab_of(d, sl.plan.L(o.level - 1)); cb = buf(sl, o.level - 1).b; }
            const int rc = mgx_slab_restrict(&fs, lb.u, lb.b, &cs, cb, nullptr, o.crow_lo, o.crow_hi, d->cfg.restrict_mode, 1, (void*)sl.st);
            if (rc != MGX_OK) return s->fail(rc, "mgx_slab_restrict failed on a slab");
            return MGX_OK;
        }
        case MGX_DOP_RESTRICT_RHS: {
            const mgx_dist_level& g = sl.plan.L(o.level);
            DistLevelBuf& lb = buf(sl, o.level);
            const mgx_slab fs = slab_of(d, g);
            mgx_slab cs;
            void* cb;
            if (o.coarse_is_cut) { cs = mgx_slab{cut, dt, sl.plan.c_rows, sl.plan.c_row0, d->cfg.arith}; cb = sl.c_own; }
            else { cs = slab_of(d, sl.plan.L(o.level - 1)); cb = buf(sl, o.level - 1).b; }
            const int rc = mgx_slab_restrict(&fs, nullptr, lb.b, &cs, cb, nullptr, o.crow_lo, o.crow_hi, d->cfg.restrict_mode, 0, (void*)sl.st);
            if (rc != MGX_OK) return s->fail(rc, "mgx_slab_restrict (right-hand side) failed on a slab");
            return MGX_OK;
        }
        case MGX_DOP_PROLONG_SET:
        case MGX_DOP_PROLONG: {
            const mgx_dist_level& g = sl.plan.L(o.level);
            DistLevelBuf& lb = buf(sl, o.level);
            const mgx_slab fs = slab_of(d, g);
            mgx_slab cs;
            const void* ce;
            if (o.coarse_is_cut) { cs = mgx_slab{cut, dt, (1 << cut) + 1, 0, d->cfg.arith}; ce = sl.coarse->lv[cut].u; }
            else { cs = slab_of(d, sl.plan.L(o.level - 1)); ce = buf(sl, o.level - 1).u; }
            const int rc = mgx_slab_prolong(&fs, lb.u, &cs, ce, o.row_lo, o.row_hi, o.op == MGX_DOP_PROLONG ? 1 : 0, (void*)sl.st);
            if (rc != MGX_OK) return s->fail(rc, "mgx_slab_prolong failed on a slab");
            return MGX_OK;
        }
        case MGX_DOP_COARSE: {
            // levels cut..coarsest from e = 0 (PS:613, 617), on this slab's stream, replayed from a hipGraph
            double unused = 0.0;
            const int rc = cycle_body(sl.coarse, false, true, &unused);
            if (rc != MGX_OK) return s->fail(rc, std::string("coarse V-cycle: ") + sl.coarse->err);
            return MGX_OK;
        }
        case MGX_DOP_COARSE_FMG: {
            // PS:629-650 on levels cut..coarsest (the gathered right-hand side is the handle's level-cut B)
            const int rc = fmg(sl.coarse);
            if (rc != MGX_OK) return s->fail(rc, std::string("coarse fullmultigrid: ") + sl.coarse->err);
            return MGX_OK;
        }
        case MGX_DOP_SUMSQ: {
            const mgx_dist_level& g = sl.plan.L(o.level);
            DistLevelBuf& lb = buf(sl, o.level);
            const mgx_slab fs = slab_of(d, g);
            const int rc = mgx_slab_residual_sumsq(&fs, lb.u, lb.b, o.row_lo, o.row_hi, sl.scratch, sl.sum_dev, (void*)sl.st);
            if (rc != MGX_OK) return s->fail(rc, "mgx_slab_residual_sumsq failed on a slab");
            return MGX_OK;
        }
        default:
            return s->fail(MGX_ERR_STATE, "unknown operation in a slab plan");
    }
}

// run the operations the planners emitted (the same sequence of op codes on every slab)
int dist_run(mgx_solver* s, mgx_dist* d, double* norm_out)
{
    const size_t nops = d->slabs[0].ops.size();
    for (auto& sl : d->slabs)
        if (sl.ops.size() != nops) return s->fail(MGX_ERR_STATE, "slab plans differ in length");
    const int Lf = d->cfg.finest_level;
    for (size_t i = 0; i < nops; ++i) {
        const mgx_dist_op& o = d->slabs[0].ops[i];
        int rc = MGX_OK;
        switch (o.op) {
            case MGX_DOP_EXCHANGE: {
                auto none_after = [](DistSlab&, hipStream_t) { return (int)MGX_OK; };
                auto none_meanwhile = [](DistSlab&) { return (int)MGX_OK; };
                const int j = d->overlap ? dist_overlap_candidate(d, d->slabs[0], i) : -1;
                bool fits = j > 0;
                for (auto& sl : d->slabs) {
                    int a, b;
                    fits = fits && dist_overlap_candidate(d, sl, i) == j && dist_interior_rows(d, sl, sl.ops[j], &a, &b);
                }
                if (!fits) { rc = dist_exchange(s, d, o, false, none_after, none_meanwhile); break; }
                // operations between the exchange and the pass (zero fills of coarser guesses) first
                for (size_t k = i + 1; k < (size_t)j && rc == MGX_OK; ++k)
                    for (auto& sl : d->slabs)
                        if ((rc = dist_local_op(s, d, sl, sl.ops[k])) != MGX_OK) break;
                if (rc != MGX_OK) break;
                const bool timed = d->cfg.profile && d->slabs[0].ops[j].level == Lf;
                std::vector<int> flags(d->slabs.size(), 0);
                {
                    DistProf pr(d, d->slabs[0], timed, MGX_PROF_SMOOTH_FINE, d->slabs[0].ops[j].mu);
                    pr.launches(1);
                    auto bands = [&](DistSlab& sl, hipStream_t xst) {
                        const mgx_dist_op& c = sl.ops[j];
                        int a, b, f = 0, r = MGX_OK;
                        (void)dist_interior_rows(d, sl, c, &a, &b);
                        const int zin = sl.zero_pending[c.level - (d->cut + 1)];       // (three launches, one implicit-zero input)
                        if (a > c.row_lo) r = dist_cycle_rows(s, d, sl, c, c.row_lo, a, xst, &f, zin);
                        if (r == MGX_OK && b < c.row_hi) r = dist_cycle_rows(s, d, sl, c, b, c.row_hi, xst, &f, zin);
                        return r;
                    };
                    auto interior = [&](DistSlab& sl) {
                        const mgx_dist_op& c = sl.ops[j];
                        int a, b;
                        (void)dist_interior_rows(d, sl, c, &a, &b);
                        return dist_cycle_rows(s, d, sl, c, a, b, sl.st, &flags[&sl - &d->slabs[0]], sl.zero_pending[c.level - (d->cut + 1)]);
                    };
                    rc = dist_exchange(s, d, o, true, bands, interior);
                }
                if (rc != MGX_OK) break;
                for (auto& sl : d->slabs) {
                    sl.zero_pending[sl.ops[j].level - (d->cut + 1)] = 0;
                    if (flags[&sl - &d->slabs[0]]) std::swap(buf(sl, sl.ops[j].level).u, buf(sl, sl.ops[j].level).tmp);
                }
                if (d->slabs[0].ops[j].level == Lf) {
                    const double n = (double)((1 << Lf) - 1);
                    d->fine_updates += (double)d->slabs[0].ops[j].mu * n * n;
                }
                d->overlapped += 1;
                i = (size_t)j;                               // the pass is done
                break;
            }
            case MGX_DOP_GATHER_CUT: rc = dist_gather_cut(s, d); break;
            case MGX_DOP_ALLREDUCE_NORM: {
                double sum = 0.0;
                for (auto& sl : d->slabs) {
                    DCHK(s, hipSetDevice(sl.device));
                    if (d->use_rccl) NCHK(s, ncclAllReduce(sl.sum_dev, sl.sum_dev, 1, ncclDouble, ncclSum, d->comm, sl.st));
                    DCHK(s, hipMemcpyAsync(sl.sum_host, sl.sum_dev, sizeof(double), hipMemcpyDeviceToHost, sl.st));
                }
                for (auto& sl : d->slabs) {
                    DCHK(s, hipSetDevice(sl.device));
                    DCHK(s, hipStreamSynchronize(sl.st));
                    sum += *sl.sum_host;                        // slab order: deterministic
                }
                if (d->have_ext && d->world > 1 && d->ext.allreduce_sum(d->ext.ctx, &sum) != 0)
                    return s->fail(MGX_ERR_HIP, "transport allreduce failed");
                if (norm_out) *norm_out = std::sqrt(sum);
                break;
            }
            default:
                for (auto& sl : d->slabs) {
                    if (sl.ops[i].op != o.op) return s->fail(MGX_ERR_STATE, "slab plans differ");
                    if ((rc = dist_local_op(s, d, sl, sl.ops[i])) != MGX_OK) break;
                }
                if (rc == MGX_OK && (o.op == MGX_DOP_CYCLE || o.op == MGX_DOP_SMOOTH) && o.level == Lf) {
                    const double n = (double)((1 << Lf) - 1);
                    d->fine_updates += (double)o.mu * n * n;
                }
        }
        if (rc != MGX_OK) return rc;
    }
    return MGX_OK;
}

int dist_vcycle(mgx_solver* s, mgx_dist* d)
{
    for (auto& sl : d->slabs) { sl.ops.clear(); sl.plan.emit_vcycle(sl.ops); }
    return dist_run(s, d, nullptr);
}

int dist_fmg(mgx_solver* s, mgx_dist* d)
{
    for (auto& sl : d->slabs) { sl.ops.clear(); sl.plan.emit_fmg(sl.ops); }
    return dist_run(s, d, nullptr);
}

int dist_norm(mgx_solver* s, mgx_dist* d, double* out)
{
    for (auto& sl : d->slabs) { sl.ops.clear(); sl.plan.emit_norm(sl.ops); }
    return dist_run(s, d, out);
}

int dist_sync(mgx_solver* s, mgx_dist* d)
{
    for (auto& sl : d->slabs) { DCHK(s, hipSetDevice(sl.device)); DCHK(s, hipStreamSynchronize(sl.st)); }
    return MGX_OK;
}

// ---- data in / out (finest level only) ---------------------------------------------------------------
int dist_fill(mgx_solver* s, mgx_dist* d, int which, int kind, double f, uint64_t seed)
{
    const int Lf = d->cfg.finest_level;
    for (auto& sl : d->slabs) {
        DCHK(s, hipSetDevice(sl.device));
        const mgx_dist_level& g = sl.plan.L(Lf);
        DistLevelBuf& lb = buf(sl, Lf);
        const dim3 blk(256), grd((g.N + 1 + 255) / 256, g.rows);
        if (which == MGX_VEC_B) {
            if (d->f64) hipLaunchKernelGGL(k_fill_rhs_rows<double>, grd, blk, 0, sl.st, (double*)lb.b, g.N, lb.pitch, kind, f, g.row0, g.rows);
            else hipLaunchKernelGGL(k_fill_rhs_rows<float>, grd, blk, 0, sl.st, (float*)lb.b, g.N, lb.pitch, kind, f, g.row0, g.rows);
        } else {
            if (d->f64) hipLaunchKernelGGL(k_fill_random_rows<double>, grd, blk, 0, sl.st, (double*)lb.u, g.N, lb.pitch, seed, g.row0, g.rows);
            else hipLaunchKernelGGL(k_fill_random_rows<float>, grd, blk, 0, sl.st, (float*)lb.u, g.N, lb.pitch, seed, g.row0, g.rows);
            sl.plan.guess_set();                       // every row of the slab, halos included, holds the global field
        }
    }
    return dist_sync(s, d);
}

// whole-grid host vector (reference layout, n x n interior) -> every slab's rows, halos included
int dist_set(mgx_solver* s, mgx_dist* d, int which, const void* src, size_t count)
{
    const int Lf = d->cfg.finest_level;
    const size_t n = (size_t)(1 << Lf) - 1;
    if (count != n * n) return s->fail(MGX_ERR_INVALID, "vector length must be n*n with n = 2^level - 1");
    for (auto& sl : d->slabs) {
        DCHK(s, hipSetDevice(sl.device));
        const mgx_dist_level& g = sl.plan.L(Lf);
        DistLevelBuf& lb = buf(sl, Lf);
        char* t = (char*)((which == MGX_VEC_U) ? lb.u : lb.b);
        const int r_lo = std::max(g.row0, 1), r_hi = std::min(g.row0 + g.rows, g.N);    // global unknown rows held
        if (r_hi > r_lo) {
            char* dst = t + ((size_t)(r_lo - g.row0) * (size_t)lb.pitch + 1) * d->es;
            const char* sp = (const char*)src + (size_t)(r_lo - 1) * n * d->es;
            DCHK(s, hipMemcpy2DAsync(dst, (size_t)lb.pitch * d->es, sp, n * d->es, n * d->es, (size_t)(r_hi - r_lo),
                                     hipMemcpyHostToDevice, sl.st));
        }
        if (which == MGX_VEC_U) sl.plan.guess_set();
    }
    return dist_sync(s, d);
}

// owned rows of every local slab -> the whole-grid host vector (rows of other ranks are left untouched)
int dist_get(mgx_solver* s, mgx_dist* d, int which, void* dst, size_t count)
{
    const int Lf = d->cfg.finest_level;
    const size_t n = (size_t)(1 << Lf) - 1;
    if (count != n * n) return s->fail(MGX_ERR_INVALID, "vector length must be n*n with n = 2^level - 1");
    for (auto& sl : d->slabs) {
        DCHK(s, hipSetDevice(sl.device));
        const mgx_dist_level& g = sl.plan.L(Lf);
        DistLevelBuf& lb = buf(sl, Lf);
        const char* t = (const char*)((which == MGX_VEC_U) ? lb.u : lb.b);
        const int r_lo = std::max(g.own_lo, 1), r_hi = std::min(g.own_hi, g.N);
        if (r_hi > r_lo) {
            const char* sp = t + ((size_t)(r_lo - g.row0) * (size_t)lb.pitch + 1) * d->es;
            char* dp = (char*)dst + (size_t)(r_lo - 1) * n * d->es;
            DCHK(s, hipMemcpy2DAsync(dp, n * d->es, sp, (size_t)lb.pitch * d->es, n * d->es, (size_t)(r_hi - r_lo),
                                     hipMemcpyDeviceToHost, sl.st));
        }
    }
    return dist_sync(s, d);
}

int dist_zero_u(mgx_solver* s, mgx_dist* d)
{
    const int Lf = d->cfg.finest_level;
    for (auto& sl : d->slabs) {
        DCHK(s, hipSetDevice(sl.device));
        DistLevelBuf& lb = buf(sl, Lf);
        DCHK(s, hipMemsetAsync(lb.u, 0, lb.bytes, sl.st));
        sl.plan.guess_set();
    }
    return dist_sync(s, d);
}

int dist_prof_collect(mgx_solver* s, mgx_dist* d)
{
    if (d->ev_used.empty()) return MGX_OK;
    int rc = dist_sync(s, d);
    if (rc) return rc;
    for (auto& p : d->ev_used) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            d->prof_ms[p.cls] += ms;
            d->prof_launches[p.cls] += p.launches;
            d->prof_sweeps[p.cls] += p.sweeps;
        }
        d->ev_free.push_back(p);
    }
    d->ev_used.clear();
    return MGX_OK;
}

// mgx_solve on a multi-GPU handle: PS:727 run to a tolerance with V-cycles (PS:575-627)
int dist_solve(mgx_solver* s, mgx_dist* d, double tol, int max_cycles, mgx_stats* stats, double* history, int history_cap)
{
    std::vector<double> hist;
    hist.reserve(max_cycles + 1);
    d->fine_updates = 0.0;
    int rc = dist_sync(s, d);
    if (rc) return rc;
    const auto t0 = std::chrono::steady_clock::now();
    double r = 0.0;
    if ((rc = dist_norm(s, d, &r))) return rc;
    hist.push_back(r);
    int k = 0;
    for (; k < max_cycles; ++k) {
        if (hist[k] <= tol * hist[0]) break;
        if (k == 0 && d->cfg.schedule == MGX_SCHEDULE_FMG) { if ((rc = dist_fmg(s, d))) return rc; }   // PS:727
        else if ((rc = dist_vcycle(s, d))) return rc;
        if ((rc = dist_norm(s, d, &r))) return rc;
        hist.push_back(r);
    }
    if ((rc = dist_sync(s, d))) return rc;
    for (auto& sl : d->slabs) { DCHK(s, hipSetDevice(sl.device)); DCHK(s, hipGetLastError()); }
    const auto t1 = std::chrono::steady_clock::now();
    if (stats) {
        stats->cycles = k;
        stats->initial_residual = hist.front();
        stats->final_residual = hist.back();
        stats->converged = (hist.back() <= tol * hist.front()) ? 1 : 0;
        stats->seconds = std::chrono::duration<double>(t1 - t0).count();
        stats->fine_updates = d->fine_updates;
        stats->history_len = (int)hist.size();
    }
    if (history)
        for (int i = 0; i < (int)hist.size() && i < history_cap; ++i) history[i] = hist[i];
    return MGX_OK;
}

} // namespace
