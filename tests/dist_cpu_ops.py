"""CPU stand-ins for the device side of the multi-GPU driver, used ONLY by the
gloo tests: numpy slab operators with the semantics of the mgx_slab_* C-ABI
(include/mgx.h) and an oracle-backed coarse solver.  The product never imports
this module (it lives under tests/)."""
import numpy as np
import torch


def _pitch(level, itemsize):
    N = 1 << level
    align = 256 // itemsize
    return (N + 1 + align - 1) // align * align


def _nbr(a, lo, hi, N):
    """((north + west) + east) + south on rows [lo,hi), columns 1..N-1 of padded array a"""
    c = slice(1, N)
    return ((a[lo - 1:hi - 1, c] + a[lo:hi, 0:N - 1]) + a[lo:hi, 2:N + 1]) + a[lo + 1:hi + 1, c]


class CpuSlabOps:
    def __init__(self, dtype=torch.float64):
        self.dtype = dtype
        self.np = np.float64 if dtype == torch.float64 else np.float32

    def pitch(self, level):
        return _pitch(level, 8 if self.dtype == torch.float64 else 4)

    def zeros(self, rows, level):
        return torch.zeros((rows, self.pitch(level)), dtype=self.dtype)

    def smooth(self, kind, level, row0, u, b, tmp, lo, hi, mu, omega, shrink=True):
        N = 1 << level
        first, last = 1 - row0, N - row0
        src, dst = u.numpy(), tmp.numpy()
        bb = b.numpy()
        T = self.np
        om = T(omega)
        c0, c1 = T(1.0 - float(om)), T(float(om) / 4.0)
        c = slice(1, N)
        flip = False
        for k in range(mu):
            per = 2 if kind == "rbgs" else 1
            ext = per * (mu - 1 - k) if shrink else 0
            l, h = max(lo - ext, first), min(hi + ext, last)
            if kind == "jacobi":
                assert l >= 1 and h <= src.shape[0] - 1
                dst[l:h, c] = (c0 * src[l:h, c] + c1 * bb[l:h, c]) + c1 * _nbr(src, l, h, N)
            else:
                # red on rows [l-1, h+1) clipped to the unknown rows, then black on [l, h)
                rl, rh = max(l - 1, first), min(h + 1, last)
                assert max(rl - 1, -row0) >= 0 and min(rh, N - row0) <= src.shape[0] - 1
                work = src.copy()
                rows = np.arange(work.shape[0])[:, None] + row0
                cols = np.arange(work.shape[1])[None, :]
                red = ((rows + cols) & 1) == 0
                new = T(0.25) * (bb[rl:rh, c] + _nbr(src, rl, rh, N))
                m = red[rl:rh, c]
                work[rl:rh, c] = np.where(m, new, src[rl:rh, c])
                newb = T(0.25) * (bb[l:h, c] + _nbr(work, l, h, N))
                mb = ~red[l:h, c]
                dst[l:h, c] = np.where(mb, newb, work[l:h, c])
            src, dst = dst, src
            flip = not flip
        return (tmp, u) if flip else (u, tmp)

    def cycle(self, kind, level, row0, u, b, tmp, lo, hi, mu, omega, crow0=0, coarse_e=None, coarse_b=None,
              clo=0, chi=0, mode=0, want_sumsq=False):
        """mgx_slab_cycle composed from the separate operators, and STRICTER than the device about
        what it leaves behind: every row outside [lo,hi) (the only rows the call promises) is
        poisoned with NaN, so a driver that relied on anything else would fail the test."""
        N = 1 << level
        per = 2 if kind == "rbgs" else 1
        first, last = 1 - row0, N - row0
        elo = 1 if want_sumsq else 0                       # the folded norm needs the result one row beyond
        if coarse_e is not None:                           # the kernel corrects every row it reads
            self.prolong(level, row0, u, crow0, coarse_e, max(lo - elo - per * mu, first), min(hi + elo + per * mu, last),
                         add=True)
        res, scr = self.smooth(kind, level, row0, u, b, tmp, max(lo - elo, first), min(hi + elo, last), mu, omega,
                               shrink=True)
        if coarse_b is not None:
            self.restrict(level, row0, res, b, crow0, coarse_b, None, clo, chi, mode, fused=True)
        sq = self.sumsq(level, row0, res, b, lo, hi) if want_sumsq else None
        rows = np.arange(res.shape[0])
        bad = ((rows < lo) | (rows >= hi)) & (rows + row0 >= 1) & (rows + row0 <= N - 1)
        for t in (res, scr):
            t.numpy()[bad, 1:N] = np.nan
        return res, scr, sq

    def restrict(self, flevel, frow0, u, b, crow0, cb, czero, clo, chi, mode, fused=True):
        N = 1 << flevel
        NC = N // 2
        T = self.np
        w = T(0.0625) if mode == 1 else T(0.25)
        bb = b.numpy()
        off = 2 * crow0 - frow0
        if chi <= clo:
            return
        ylo, yhi = 2 * clo + off - 1, 2 * (chi - 1) + off + 2      # field rows [ylo, yhi)
        f = np.zeros_like(bb)
        c = slice(1, N)
        if fused:
            uu = u.numpy()
            f[ylo:yhi, c] = bb[ylo:yhi, c] - (-_nbr(uu, ylo, yhi, N) + T(4) * uu[ylo:yhi, c])
        else:
            f[ylo:yhi, c] = bb[ylo:yhi, c]
        out = cb.numpy()
        for I in range(clo, chi):
            y = 2 * I + off
            J = np.arange(1, NC)
            x = 2 * J
            corners = ((f[y - 1, x - 1] + f[y - 1, x + 1]) + f[y + 1, x - 1]) + f[y + 1, x + 1]
            edges = ((f[y, x - 1] + f[y, x + 1]) + f[y - 1, x]) + f[y + 1, x]
            out[I, 1:NC] = w * ((corners + T(2) * edges) + T(4) * f[y, x])
            if czero is not None:
                czero.numpy()[I, :NC] = 0

    def prolong(self, flevel, frow0, u, crow0, e, lo, hi, add=True):
        N = 1 << flevel
        T = self.np
        ee, uu = e.numpy(), u.numpy()
        off = 2 * crow0 - frow0
        x = np.arange(1, N)
        for r in range(lo, hi):
            y = r - off
            I = y >> 1
            J = x >> 1
            even_x = (x & 1) == 0
            if (y & 1) == 0:
                val = np.where(even_x, ee[I, J], T(0.5) * (ee[I, J] + ee[I, J + 1]))
            else:
                val = np.where(even_x, T(0.5) * (ee[I, J] + ee[I + 1, J]),
                               T(0.25) * (((ee[I, J] + ee[I + 1, J]) + ee[I, J + 1]) + ee[I + 1, J + 1]))
            uu[r, 1:N] = (uu[r, 1:N] + val) if add else val

    def sumsq(self, level, row0, u, b, lo, hi):
        N = 1 << level
        uu, bb = u.numpy(), b.numpy()
        c = slice(1, N)
        r = (bb[lo:hi, c] - (-_nbr(uu, lo, hi, N) + self.np(4) * uu[lo:hi, c])).astype(np.float64)
        return torch.tensor([float(np.sum(r * r))], dtype=torch.float64)


class OracleCoarseSolver:
    """levels coarsest..cut with the CPU oracle (tests only)"""

    def __init__(self, po, cut_level, coarsest_level, cfg):
        self.level = cut_level
        self.s = po.Solver(finest_level=cut_level, coarsest_level=coarsest_level, mu0=cfg.get("mu0", 0), mu1=cfg["mu1"], mu2=cfg["mu2"],
                           omega=cfg["omega"], smoother=1 if cfg["smoother"] == "rbgs" else 0, schedule=0,
                           restrict_mode=cfg["restrict_mode"], bottom=cfg["bottom"])

    def vcycle_from_zero(self, b_full, e_full):
        N = 1 << self.level
        b = np.ascontiguousarray(b_full.numpy()[1:N, 1:N].astype(np.float64))
        e = self.s.vcycle(self.level, np.zeros_like(b), b)
        e_full.zero_()
        e_full.numpy()[1:N, 1:N] = e

    def fmg(self, b_full, e_full):
        """e_full <- fullmultigrid (PS:629-650) on levels cut..coarsest for the gathered right-hand side"""
        N = 1 << self.level
        b = np.ascontiguousarray(b_full.numpy()[1:N, 1:N].astype(np.float64))
        e = self.s.fmg(self.level, b)
        e_full.zero_()
        e_full.numpy()[1:N, 1:N] = e
