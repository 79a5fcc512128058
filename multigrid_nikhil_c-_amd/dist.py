"""Multi-GPU V-cycle: the finest levels slab-decomposed by rows, one process per
GPU, halo rows exchanged with torch.distributed (backend "nccl" = RCCL over
xGMI), the levels at and below a cut-over level solved redundantly on every
rank ("the bottom solve stays on one GPU": no coarse level is ever distributed
below the cut, and no scatter is needed because every rank holds the result).

The reference has no distributed path at all (one sycl::queue, PS:659); what is
reproduced is its V-cycle (PS:575-627) with the grid operators applied slab by
slab.  Because every slab operator is the same kernel as on one GPU and the
halo rows carry exactly the neighbour's values, the distributed iterate is
bit-identical to the single-GPU iterate.

Communication plan (halo messages are latency-bound, so send few):
  * deep halos: a block of mu Jacobi sweeps needs mu halo rows (2 mu for RB-GS)
    exchanged ONCE; sweep k then updates a range that shrinks by one row per
    sweep at each interior slab edge (mgx_slab_jacobi(..., shrink=1)),
    recomputing the neighbour's rows redundantly instead of talking per sweep.
  * redundant rows instead of messages: pre-smoothing is carried out on the owned
    rows plus the rows the restriction (2) and the post-smoothing (mu2) will
    need, and the correction is prolongated onto that extended range as well,
    so neither the restriction nor the post-smoothing needs an exchange.
  * deep correction halos: every level below the finest leaves its post-smoothed correction
    valid as far into its halos as the level above reads it when prolongating, so the
    correction is never exchanged either.
  * what is left per cycle: ONE exchange of u on the finest level (it serves the next
    pre-smoothing), and ONE exchange per further distributed level, of the restricted
    right-hand side (coarse guesses are zero, halos included, so they need none).
  * one all_gather of the restricted residual at the cut-over level, one
    all_reduce of a double for ||r||^2.

Row ownership on level l (N = 2^l rows 0..N): rank g owns rows
[g N/P, (g+1) N/P), the last rank also the boundary row N.  Coarse row I sits
on fine row 2I, so ownership nests across levels.

`SlabOps` is the only thing that touches the device.  HipSlabOps (below) calls
the C-ABI slab operators of libmgx on torch CUDA tensors; the CPU/gloo tests
inject a numpy implementation (tests/dist_cpu_ops.py) - the product never does.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import torch
import torch.distributed as dist

from . import binding as B


# ---------------------------------------------------------------------------------
# device operators on slabs
# ---------------------------------------------------------------------------------
class HipSlabOps:
    """libmgx slab operators on torch CUDA tensors (rows x pitch, C-contiguous)."""

    def __init__(self, dtype=torch.float64, device=None):
        if not torch.cuda.is_available():
            hint = ""
            if B.lib_loaded():
                # libmgx links the system ROCm runtime, the torch wheel carries its own: whichever is
                # loaded first serves both, and torch's CUDA layer only comes up on its own one
                hint = ("; libmgx.so was loaded before torch in this process - import torch (or this module) "
                        "before the first libmgx call")
            raise B.MgxError("HipSlabOps needs a HIP device (no CPU fallback)" + hint)
        self.lib = B.lib()
        self.dtype = dtype
        self.code = B.DTYPE_F64 if dtype == torch.float64 else B.DTYPE_F32
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        self._scratch = {}

    def pitch(self, level):
        return int(self.lib.mgx_level_pitch(level, self.code))

    def zeros(self, rows, level):
        return torch.zeros((rows, self.pitch(level)), dtype=self.dtype, device=self.device)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _slab(self, level, t, row0):
        return B.Slab(level=level, dtype=self.code, rows=t.shape[0], row0=row0)

    @staticmethod
    def _chk(st, what):
        if st != 0:
            raise B.MgxError(f"{what} failed with status {st} ({B.lib().mgx_status_string(st).decode()})")

    def smooth(self, kind, level, row0, u, b, tmp, lo, hi, mu, omega, shrink=True):
        """mu sweeps on local rows [lo,hi); returns (result, scratch) tensors."""
        s = self._slab(level, u, row0)
        flag = C.c_int(0)
        if kind == "rbgs":
            st = self.lib.mgx_slab_rbgs(C.byref(s), u.data_ptr(), b.data_ptr(), tmp.data_ptr(), lo, hi, mu,
                                        1 if shrink else 0, C.byref(flag), self._stream())
        else:
            st = self.lib.mgx_slab_jacobi(C.byref(s), u.data_ptr(), b.data_ptr(), tmp.data_ptr(), lo, hi, mu,
                                          float(omega), 1 if shrink else 0, C.byref(flag), self._stream())
        self._chk(st, "mgx_slab_" + kind)
        return (tmp, u) if flag.value else (u, tmp)

    def cycle(self, kind, level, row0, u, b, tmp, lo, hi, mu, omega, crow0=0, coarse_e=None, coarse_b=None,
              clo=0, chi=0, mode=0, want_sumsq=False):
        """mu sweeps on local rows [lo,hi) (deep-halo shrinking) with the transfers folded in
        (mgx_slab_cycle): input u + P coarse_e, and / or the restricted residual of the result
        -> coarse_b rows [clo,chi), or sum (b - A u)^2 over [lo,hi).  Returns (result, scratch, sumsq)."""
        s = self._slab(level, u, row0)
        ct = coarse_e if coarse_e is not None else coarse_b
        cs = self._slab(level - 1, ct, crow0) if ct is not None else None
        out = scratch = None
        if want_sumsq:
            key = (level, u.shape[0])
            if key not in self._scratch:
                n = int(self.lib.mgx_slab_scratch_doubles(C.byref(s)))
                self._scratch[key] = torch.zeros(n, dtype=torch.float64, device=self.device)
            scratch = self._scratch[key]
            out = torch.zeros(1, dtype=torch.float64, device=self.device)
        flag = C.c_int(0)
        st = self.lib.mgx_slab_cycle(C.byref(s), u.data_ptr(), b.data_ptr(), tmp.data_ptr(), lo, hi, mu, float(omega),
                                     B.SMOOTHER_RBGS if kind == "rbgs" else B.SMOOTHER_JACOBI,
                                     C.byref(cs) if cs is not None else None,
                                     coarse_e.data_ptr() if coarse_e is not None else None,
                                     coarse_b.data_ptr() if coarse_b is not None else None, clo, chi, mode,
                                     scratch.data_ptr() if scratch is not None else None,
                                     out.data_ptr() if out is not None else None, C.byref(flag), self._stream())
        self._chk(st, "mgx_slab_cycle")
        return ((tmp, u) if flag.value else (u, tmp)) + (out,)

    def restrict(self, flevel, frow0, u, b, crow0, cb, czero, clo, chi, mode, fused=True):
        fs = self._slab(flevel, b, frow0)
        cs = self._slab(flevel - 1, cb, crow0)
        st = self.lib.mgx_slab_restrict(C.byref(fs), u.data_ptr() if u is not None else None, b.data_ptr(), C.byref(cs),
                                        cb.data_ptr(), czero.data_ptr() if czero is not None else None, clo, chi,
                                        mode, 1 if fused else 0, self._stream())
        self._chk(st, "mgx_slab_restrict")

    def prolong(self, flevel, frow0, u, crow0, e, lo, hi, add=True):
        fs = self._slab(flevel, u, frow0)
        cs = self._slab(flevel - 1, e, crow0)
        st = self.lib.mgx_slab_prolong(C.byref(fs), u.data_ptr(), C.byref(cs), e.data_ptr(), lo, hi, 1 if add else 0,
                                       self._stream())
        self._chk(st, "mgx_slab_prolong")

    def sumsq(self, level, row0, u, b, lo, hi):
        """sum over rows [lo,hi) of (b - A u)^2 as a 1-element float64 device tensor"""
        s = self._slab(level, u, row0)
        key = (level, u.shape[0])
        if key not in self._scratch:
            n = int(self.lib.mgx_slab_scratch_doubles(C.byref(s)))
            self._scratch[key] = torch.zeros(n, dtype=torch.float64, device=self.device)
        out = torch.zeros(1, dtype=torch.float64, device=self.device)
        st = self.lib.mgx_slab_residual_sumsq(C.byref(s), u.data_ptr(), b.data_ptr(), lo, hi,
                                              self._scratch[key].data_ptr(), out.data_ptr(), self._stream())
        self._chk(st, "mgx_slab_residual_sumsq")
        return out


class HipCoarseSolver:
    """Levels coarsest..cut on one GPU: a plain libmgx handle, fed and read in
    device memory.  Every rank runs it on the same gathered right-hand side."""

    def __init__(self, cut_level, coarsest_level, cfg, dtype):
        self.level = cut_level
        self.mg = B.Multigrid(finest_level=cut_level, coarsest_level=coarsest_level, mu1=cfg["mu1"], mu2=cfg["mu2"],
                              omega=cfg["omega"], smoother=B.SMOOTHER_RBGS if cfg["smoother"] == "rbgs" else B.SMOOTHER_JACOBI,
                              dtype=B.DTYPE_F64 if dtype == torch.float64 else B.DTYPE_F32, schedule=B.SCHEDULE_V,
                              restrict_mode=cfg["restrict_mode"], bottom=cfg["bottom"],
                              device=torch.cuda.current_device())

    def vcycle_from_zero(self, b_full, e_full):
        """e_full <- one V-cycle (PS:575-627) for A e = b_full from e = 0 (PS:613)"""
        torch.cuda.current_stream().synchronize()          # the handle runs on its own stream
        self.mg.set_level_device(self.level, B.VEC_B, b_full.data_ptr())
        self.mg.vcycle_zero()            # from e = 0, replayed from a hipGraph
        self.mg.get_level_device(self.level, B.VEC_U, e_full.data_ptr())


# ---------------------------------------------------------------------------------
# the distributed hierarchy
# ---------------------------------------------------------------------------------
@dataclass
class SlabLevel:
    level: int
    N: int
    own_lo: int          # global rows owned: [own_lo, own_hi)
    own_hi: int
    halo: int
    row0: int            # global row of local row 0
    rows: int
    u: torch.Tensor = None
    b: torch.Tensor = None
    tmp: torch.Tensor = None
    u_halo: int = 0      # how many halo rows of u currently hold the neighbours' values

    # local index helpers
    def loc(self, g):
        return g - self.row0

    @property
    def upd_lo(self):    # first owned unknown row (local)
        return max(self.own_lo, 1) - self.row0

    @property
    def upd_hi(self):    # one past the last owned unknown row (local)
        return min(self.own_hi, self.N) - self.row0


class DistMultigrid:
    """V-cycle multigrid with the levels above `cut_level` split into row slabs."""

    def __init__(self, ops, coarse, finest_level, cut_level, mu1=10, mu2=10, omega=2.0 / 3.0, smoother="jacobi",
                 restrict_mode=0, group=None, staged_halo=False, fold=None, deep=None):
        self.ops, self.coarse = ops, coarse
        self.Lf, self.Lcut = finest_level, cut_level
        self.mu1, self.mu2, self.omega, self.smoother, self.restrict_mode = mu1, mu2, omega, smoother, restrict_mode
        self.group = group
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.g = dist.get_rank(group) if dist.is_initialized() else 0
        self.staged = staged_halo          # move halos through host memory (gloo with device tensors)
        if finest_level <= cut_level:
            raise ValueError("finest_level must be above cut_level (otherwise use the single-GPU handle)")
        per_sweep = 2 if smoother == "rbgs" else 1
        self.per = per_sweep
        # fold the transfers into the smoother passes (ops.cycle) as the single-GPU handle does:
        # the restriction rides on the last pre-smoothing pass, the correction on the first
        # post-smoothing pass, the residual norm on the last one (MGX_DIST_FOLD=0: separate kernels)
        if fold is None:
            import os
            fold = os.environ.get("MGX_DIST_FOLD", "1") != "0"
        self.fold = bool(fold) and hasattr(ops, "cycle")
        self._sumsq = None                            # ||r||^2 share produced by the last post-smoothing
        # Halo plan per level, from the finest level down.  keep_post: rows beyond the owned ones the
        # post-smoothing leaves valid - 0 on the finest level; on every level below, as many as the
        # level above reads when it prolongates the correction (ext_coarse), so that the correction
        # needs NO exchange (deep = False: it is exchanged, and keep_post is 0 everywhere).
        if deep is None:
            import os
            deep = os.environ.get("MGX_DIST_DEEP", "1") != "0"
        self.deep = bool(deep)
        self._plan_halos(finest_level, cut_level, per_sweep, mu1, mu2)
        if self.deep and any((1 << l) // self.P < self.halo_of[l] for l in self.halo_of):
            self.deep = False                         # slabs too thin for the deep halos: exchange the correction
            self._plan_halos(finest_level, cut_level, per_sweep, mu1, mu2)
        self.halo = self.halo_of[finest_level]
        self.lv = {}
        self._alloc_levels(ops, finest_level, cut_level)

    def _plan_halos(self, finest_level, cut_level, per_sweep, mu1, mu2):
        self.keep_post, self.ext_post, self.ext_keep, self.ext_coarse, self.halo_of = {}, {}, {}, {}, {}
        kp = 0
        for l in range(finest_level, cut_level, -1):
            self.keep_post[l] = kp
            # halo rows the post-smoothing consumes (+1 on the finest level: the folded norm needs
            # the result one row beyond the owned rows)
            ep = per_sweep * mu2 + kp + (1 if (self.fold and l == finest_level) else 0)
            ek = max(ep, 2)                           # rows beyond the owned ones pre-smoothing leaves valid
            if self.fold:
                ek |= 1                               # the folded restriction wants its range to start on an odd row
            ec = ep // 2 + 2                          # coarse halo rows the extended prolongation reads
            self.ext_post[l], self.ext_keep[l], self.ext_coarse[l] = ep, ek, ec
            self.halo_of[l] = max(per_sweep * mu1 + ek, ec, kp)
            kp = ec if self.deep else 0

    def _alloc_levels(self, ops, finest_level, cut_level):
        for l in range(cut_level + 1, finest_level + 1):
            N = 1 << l
            halo = self.halo_of[l]
            if N % self.P or N // self.P < halo:
                raise ValueError(f"level {l}: {N} rows cannot be split over {self.P} ranks with a {halo}-row halo; "
                                 f"raise cut_level")
            own_lo = self.g * (N // self.P)
            own_hi = (self.g + 1) * (N // self.P) + (1 if self.g == self.P - 1 else 0)
            row0 = max(own_lo - halo, 0)
            row1 = min(own_hi + halo, N + 1)
            L = SlabLevel(l, N, own_lo, own_hi, halo, row0, row1 - row0)
            L.u, L.b, L.tmp = ops.zeros(L.rows, l), ops.zeros(L.rows, l), ops.zeros(L.rows, l)
            self.lv[l] = L
        # cut level: this rank's share of the restricted residual, the gathered
        # right-hand side and the correction every rank computes
        Nc = 1 << cut_level
        if Nc % self.P:
            raise ValueError("cut level rows must divide evenly over the ranks")
        self.c_rows = Nc // self.P
        self.c_row0 = self.g * self.c_rows
        self.c_own = ops.zeros(self.c_rows, cut_level)
        self.c_b = ops.zeros(Nc + 1, cut_level)
        self.c_e = ops.zeros(Nc + 1, cut_level)
        self.fine_updates = 0.0
        self.exchanges = 0
        self.exchanges_timed = 0
        self.profile = False           # record device events around finest-level smoothing
        self._events = []

    # ---- halo exchange ---------------------------------------------------------------
    def exchange(self, L, t, depth):
        """fill `depth` halo rows of tensor t (level L) on both sides from the neighbours"""
        if self.P == 1 or depth <= 0:
            return
        assert depth <= L.halo
        g, P = self.g, self.P
        lo = L.own_lo - L.row0                             # lower interior slab edge (local row)
        up = (g + 1) * (L.N // P) - L.row0                 # upper interior slab edge (local row)
        ops_list, staged = [], []

        def post(kind, view, peer):
            if self.staged:
                host = torch.empty(view.shape, dtype=view.dtype, device="cpu")
                if kind == "send":
                    host.copy_(view)
                staged.append((kind, view, host))
                view = host
            fn = dist.isend if kind == "send" else dist.irecv
            ops_list.append(dist.P2POp(fn, view, peer, self.group))

        if g > 0:
            post("send", t[lo:lo + depth], g - 1)
            post("recv", t[lo - depth:lo], g - 1)
        if g < P - 1:
            post("send", t[up - depth:up], g + 1)
            post("recv", t[up:up + depth], g + 1)
        for r in dist.batch_isend_irecv(ops_list):
            r.wait()
        for kind, view, host in staged:
            if kind == "recv":
                view.copy_(host)
        self.exchanges += 1

    # ---- operators -------------------------------------------------------------------
    def _range(self, L, ext):
        """owned unknown rows widened by `ext` rows into the halos (local indices)"""
        first, last = 1 - L.row0, L.N - L.row0
        return max(L.upd_lo - ext, first), min(L.upd_hi + ext, last)

    def _smooth(self, L, mu, keep):
        """mu sweeps whose result is valid on the owned rows +- keep"""
        if mu <= 0:
            return
        need = self.per * mu + keep
        if L.u_halo < need:
            self.exchange(L, L.u, L.halo)
            L.u_halo = L.halo
        lo, hi = self._range(L, keep)
        L.u, L.tmp = self._timed(L, mu, lo, hi, lambda: self.ops.smooth(
            self.smoother, L.level, L.row0, L.u, L.b, L.tmp, lo, hi, mu, self.omega, shrink=True))
        L.u_halo = keep

    def _own_coarse_rows(self, C_N):
        """coarse unknown rows this rank produces: global [lo, hi)"""
        per = C_N // self.P
        return max(self.g * per, 1), min((self.g + 1) * per, C_N)

    def _timed(self, L, mu, lo, hi, fn):
        """run fn() with device events around it when profiling the finest level"""
        timed = self.profile and L.level == self.Lf and L.u.is_cuda
        if timed:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        out = fn()
        if timed:
            e1.record()
            first, last = 1 - L.row0, L.N - L.row0
            rows = sum(min(hi + self.per * (mu - 1 - k), last) - max(lo - self.per * (mu - 1 - k), first)
                       for k in range(mu))
            self._events.append((e0, e1, rows * (L.N - 1) * 3 * L.u.element_size(), mu))
        if L.level == self.Lf:
            self.fine_updates += float(mu) * (L.N - 1) * (L.N - 1)
        return out

    def vcycle(self, level=None):
        """PS:575-627 on the slab hierarchy, from `level` (default: finest) down"""
        l = self.Lf if level is None else level
        L = self.lv[l]
        self._sumsq = None
        NC = L.N // 2
        glo, ghi = self._own_coarse_rows(NC)
        ext_post, ext_keep, ext_coarse, keep_post = self.ext_post[l], self.ext_keep[l], self.ext_coarse[l], self.keep_post[l]
        plo, phi = self._range(L, ext_post)               # fine rows that receive the correction
        to_cut = not (l - 1 > self.Lcut)
        Cl = None if to_cut else self.lv[l - 1]
        crow0 = self.c_row0 if to_cut else Cl.row0
        cb = self.c_own if to_cut else Cl.b
        fold_pre = self.fold and self.mu1 > 0
        fold_post = self.fold and self.mu2 > 0
        # ---- pre-smoothing (PS:581) + residual, restriction, zero coarse guess (PS:604-613) ----
        if fold_pre:
            need = self.per * self.mu1 + ext_keep
            if L.u_halo < need:
                self.exchange(L, L.u, L.halo)
                L.u_halo = L.halo
            if Cl is not None:
                Cl.u.zero_()                                                            # PS:613
            lo, hi = self._range(L, ext_keep)
            L.u, L.tmp, _ = self._timed(L, self.mu1, lo, hi, lambda: self.ops.cycle(
                self.smoother, l, L.row0, L.u, L.b, L.tmp, lo, hi, self.mu1, self.omega, crow0=crow0, coarse_b=cb,
                clo=glo - crow0, chi=ghi - crow0, mode=self.restrict_mode))
            L.u_halo = ext_keep
        else:
            if self.mu1 > 0:
                self._smooth(L, self.mu1, ext_keep)                                     # PS:581
            elif L.u_halo < ext_keep:
                self.exchange(L, L.u, L.halo)
                L.u_halo = L.halo
            if Cl is not None:
                Cl.u.zero_()                                                            # PS:613
            self.ops.restrict(l, L.row0, L.u, L.b, crow0, cb, None, glo - crow0, ghi - crow0,
                              self.restrict_mode, fused=True)                          # PS:604-611
        # ---- coarse-grid correction (PS:617) ----------------------------------------------------
        if Cl is not None:
            self.exchange(Cl, Cl.b, Cl.halo)
            Cl.u_halo = Cl.halo                        # zeros are exact halo values
            self.vcycle(l - 1)
            self._sumsq = None
            if Cl.u_halo < ext_coarse:               # never with deep halos: the level below left them valid
                self.exchange(Cl, Cl.u, ext_coarse)
                Cl.u_halo = ext_coarse
            ce, ce_row0 = Cl.u, Cl.row0
        else:
            if self.P > 1:
                if self.staged:
                    host = self.c_own.cpu()
                    full = torch.empty((NC, host.shape[1]), dtype=host.dtype)
                    dist.all_gather_into_tensor(full.view(-1), host.view(-1), group=self.group)
                    self.c_b[:NC].copy_(full)
                else:
                    dist.all_gather_into_tensor(self.c_b[:NC].view(-1), self.c_own.view(-1), group=self.group)
            else:
                self.c_b[:NC].copy_(self.c_own)
            self.coarse.vcycle_from_zero(self.c_b, self.c_e)                            # levels cut..coarsest
            ce, ce_row0 = self.c_e, 0
        # ---- correction (PS:620-624) + post-smoothing (PS:625) (+ the residual norm) -----------
        L.u_halo = min(L.u_halo, ext_post)
        if fold_post:
            lo, hi = self._range(L, keep_post)
            want = (l == self.Lf)
            L.u, L.tmp, sq = self._timed(L, self.mu2, lo, hi, lambda: self.ops.cycle(
                self.smoother, l, L.row0, L.u, L.b, L.tmp, lo, hi, self.mu2, self.omega, crow0=ce_row0, coarse_e=ce,
                want_sumsq=want))
            L.u_halo = keep_post
            self._sumsq = sq if want else None
        else:
            self.ops.prolong(l, L.row0, L.u, ce_row0, ce, plo, phi, add=True)           # PS:620-624
            self._smooth(L, self.mu2, keep_post)                                        # PS:625
            if self.mu2 <= 0:
                L.u_halo = min(L.u_halo, keep_post)

    def residual_norm(self):
        """||b - A u||_2 on the finest level (all ranks get the value)"""
        L = self.lv[self.Lf]
        if self._sumsq is not None:
            s = self._sumsq                      # produced by the last post-smoothing pass
            self._sumsq = None
        else:
            if L.u_halo < 1:
                self.exchange(L, L.u, L.halo)    # deep enough for the next pre-smoothing too
                L.u_halo = L.halo
            s = self.ops.sumsq(self.Lf, L.row0, L.u, L.b, L.upd_lo, L.upd_hi)
        if self.P > 1:
            if self.staged:
                h = s.cpu()
                dist.all_reduce(h, group=self.group)
                return float(h.item()) ** 0.5
            dist.all_reduce(s, group=self.group)
        return float(s.item()) ** 0.5

    def reset_profile(self):
        self._events = []
        self.exchanges_timed = -self.exchanges

    def collect_profile(self):
        if self._events:
            torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b, _, _ in self._events)
        out = {"ms": ms, "bytes": float(sum(x[2] for x in self._events)), "launches": sum(x[3] for x in self._events)}
        self.exchanges_timed += self.exchanges
        return out

    # ---- data ------------------------------------------------------------------------
    def set_fine(self, which, fn):
        """fill the finest level's `which` ('u' or 'b') from fn(global_rows, cols) -> values
        (torch tensors; rows/cols are 1-D index tensors of the local slab incl. halos)"""
        L = self.lv[self.Lf]
        t = getattr(L, which)
        rows = torch.arange(L.row0, L.row0 + L.rows, device=t.device)
        cols = torch.arange(0, L.N + 1, device=t.device)
        vals = fn(rows[:, None], cols[None, :], L.N).to(t.dtype)
        interior = ((rows[:, None] >= 1) & (rows[:, None] <= L.N - 1) & (cols[None, :] >= 1) & (cols[None, :] <= L.N - 1))
        t.zero_()
        t[:, : L.N + 1] = torch.where(interior, vals, torch.zeros_like(vals))
        self._sumsq = None
        if which == "u":
            L.u_halo = L.halo

    def own_interior(self, which="u"):
        """this rank's owned unknown rows of the finest level, interior columns only"""
        L = self.lv[self.Lf]
        return getattr(L, which)[L.upd_lo:L.upd_hi, 1:L.N]

    def solve(self, tol=1e-8, max_cycles=50):
        hist = [self.residual_norm()]
        k = 0
        while k < max_cycles and not (hist[k] <= tol * hist[0]):
            self.vcycle()
            hist.append(self.residual_norm())
            k += 1
        return k, hist
