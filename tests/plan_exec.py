"""Executes the slab plans of libmgx (mgx_plan_*: csrc/mgx_dist_plan.hpp, the host logic of the
multi-GPU driver) on the CPU: numpy slab operators (dist_cpu_ops.CpuSlabOps) for the device side,
torch.distributed/gloo for the halo exchange, the all-gather at the cut level and the all-reduce
of the norm.  Test infrastructure only - the product's executor is csrc/mgx_dist.hpp."""
import numpy as np
import torch
import torch.distributed as dist


class PlanRunner:
    def __init__(self, pkg, ops, coarse, plan, cfg, group=None):
        self.pkg, self.ops, self.coarse, self.plan, self.cfg, self.group = pkg, ops, coarse, plan, cfg, group
        self.P = dist.get_world_size(group) if dist.is_initialized() else 1
        self.g = dist.get_rank(group) if dist.is_initialized() else 0
        self.kind = "rbgs" if cfg["smoother"] == 1 else "jacobi"
        self.Lf, self.cut = cfg["finest_level"], plan.cut
        self.geom, self.u, self.b, self.tmp = {}, {}, {}, {}
        for l in range(self.cut + 1, self.Lf + 1):
            g = plan.level(l)
            self.geom[l] = g
            self.u[l], self.b[l], self.tmp[l] = ops.zeros(g.rows, l), ops.zeros(g.rows, l), ops.zeros(g.rows, l)
        Nc = 1 << self.cut
        self.c_own = ops.zeros(plan.c_rows, self.cut)
        self.c_b = ops.zeros(Nc + 1, self.cut)
        self.c_e = ops.zeros(Nc + 1, self.cut)
        self.sumsq = None
        self.exchanges = 0

    # ---- data ---------------------------------------------------------------------------------
    def set_fine(self, which, full):
        """full: (N+1) x (N+1) numpy array incl. the zero ring; every row the slab holds is filled"""
        g = self.geom[self.Lf]
        t = getattr(self, which)[self.Lf]
        t.zero_()
        t[:, : g.N + 1] = torch.from_numpy(full[g.row0:g.row0 + g.rows].copy())
        if which == "u":
            self.plan.guess_set(True)

    def own_interior(self):
        g = self.geom[self.Lf]
        lo, hi = max(g.own_lo, 1), min(g.own_hi, g.N)
        return lo, hi, self.u[self.Lf][lo - g.row0:hi - g.row0, 1:g.N].numpy().copy()

    # ---- collectives -----------------------------------------------------------------------------
    def _exchange(self, o):
        g = self.geom[o.level]
        t = (self.u if o.which == self.pkg.VEC_U else self.b)[o.level]
        lo = g.own_lo - g.row0
        up = (self.g + 1) * (g.N // self.P) - g.row0
        reqs = []
        if self.g > 0:
            reqs += [dist.P2POp(dist.isend, t[lo:lo + o.depth], self.g - 1, self.group),
                     dist.P2POp(dist.irecv, t[lo - o.depth:lo], self.g - 1, self.group)]
        if self.g < self.P - 1:
            reqs += [dist.P2POp(dist.isend, t[up - o.depth:up], self.g + 1, self.group),
                     dist.P2POp(dist.irecv, t[up:up + o.depth], self.g + 1, self.group)]
        for r in dist.batch_isend_irecv(reqs):
            r.wait()
        self.exchanges += 1

    # ---- one operation ---------------------------------------------------------------------------
    def _coarse(self, o, want):
        """(row0, tensor) of the level below `o.level` as operand `want` ('e' correction, 'b' right-hand side)"""
        if o.coarse_is_cut:
            return (0, self.c_e) if want == "e" else (self.plan.c_row0, self.c_own)
        g = self.geom[o.level - 1]
        return g.row0, (self.u if want == "e" else self.b)[o.level - 1]

    def run(self, oplist):
        pkg, cfg = self.pkg, self.cfg
        norm = None
        for o in oplist:
            l = o.level
            if o.op == pkg.binding.DOP_EXCHANGE:
                self._exchange(o)
            elif o.op == pkg.binding.DOP_ZERO_U:
                self.u[l].zero_()
            elif o.op == pkg.binding.DOP_CYCLE:
                g = self.geom[l]
                kw = {}
                if o.pre:
                    r0, e = self._coarse(o, "e")
                    kw.update(crow0=r0, coarse_e=e)
                if o.post == 1:
                    r0, cb = self._coarse(o, "b")
                    kw.update(crow0=r0, coarse_b=cb, clo=o.crow_lo, chi=o.crow_hi, mode=cfg["restrict_mode"])
                self.u[l], self.tmp[l], sq = self.ops.cycle(self.kind, l, g.row0, self.u[l], self.b[l], self.tmp[l], o.row_lo,
                                                            o.row_hi, o.mu, cfg["omega"], want_sumsq=(o.post == 2), **kw)
                if o.post == 2:
                    self.sumsq = sq
            elif o.op == pkg.binding.DOP_SMOOTH:
                g = self.geom[l]
                self.u[l], self.tmp[l] = self.ops.smooth(self.kind, l, g.row0, self.u[l], self.b[l], self.tmp[l], o.row_lo,
                                                         o.row_hi, o.mu, cfg["omega"], shrink=True)
            elif o.op == pkg.binding.DOP_RESTRICT:
                g = self.geom[l]
                r0, cb = self._coarse(o, "b")
                self.ops.restrict(l, g.row0, self.u[l], self.b[l], r0, cb, None, o.crow_lo, o.crow_hi, cfg["restrict_mode"], fused=True)
            elif o.op in (pkg.binding.DOP_PROLONG, pkg.binding.DOP_PROLONG_SET):
                g = self.geom[l]
                r0, e = self._coarse(o, "e")
                self.ops.prolong(l, g.row0, self.u[l], r0, e, o.row_lo, o.row_hi, add=(o.op == pkg.binding.DOP_PROLONG))
            elif o.op == pkg.binding.DOP_RESTRICT_RHS:
                g = self.geom[l]
                r0, cb = self._coarse(o, "b")
                self.ops.restrict(l, g.row0, None, self.b[l], r0, cb, None, o.crow_lo, o.crow_hi, cfg["restrict_mode"], fused=False)
            elif o.op == pkg.binding.DOP_COARSE_FMG:
                self.coarse.fmg(self.c_b, self.c_e)
            elif o.op == pkg.binding.DOP_GATHER_CUT:
                NC = 1 << self.cut
                if self.P > 1:
                    dist.all_gather_into_tensor(self.c_b[:NC].view(-1), self.c_own.view(-1), group=self.group)
                else:
                    self.c_b[:NC].copy_(self.c_own)
            elif o.op == pkg.binding.DOP_COARSE:
                self.coarse.vcycle_from_zero(self.c_b, self.c_e)
            elif o.op == pkg.binding.DOP_SUMSQ:
                g = self.geom[l]
                self.sumsq = self.ops.sumsq(l, g.row0, self.u[l], self.b[l], o.row_lo, o.row_hi)
            elif o.op == pkg.binding.DOP_ALLREDUCE_NORM:
                s = self.sumsq.clone()
                self.sumsq = None
                if self.P > 1:
                    dist.all_reduce(s, group=self.group)
                norm = float(s.item()) ** 0.5
            else:
                raise ValueError(f"unknown plan operation {o.op}")
        return norm

    def solve(self, tol=1e-8, max_cycles=50, fmg=False):
        hist = [self.run(self.plan.norm())]
        k = 0
        while k < max_cycles and not (hist[k] <= tol * hist[0]):
            self.run(self.plan.fmg() if (fmg and k == 0) else self.plan.vcycle())
            hist.append(self.run(self.plan.norm()))
            k += 1
        return k, hist
