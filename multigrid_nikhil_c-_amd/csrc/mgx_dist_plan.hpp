// mgx_dist_plan.hpp - the multi-GPU V-cycle as DATA: which slab operator, halo exchange or
// collective each row slab performs, in order.  Pure host C++ (no HIP): the executor in
// mgx_dist.hpp runs a plan on the GPUs, and the CPU tests run the very same plans over numpy
// slab operators and gloo (tests/test_dist_plan_cpu.py), so the logic that cannot be exercised
// without several GPUs is exercised without any.
//
// The reference has no distributed path (one sycl::queue, PS:659); what is reproduced is its
// V-cycle (PS:575-627) with every grid operator applied slab by slab.  Because a slab operator
// is the single-GPU kernel on a row range and halo rows carry the neighbour's values, the
// distributed iterate is bit-identical to the single-GPU one.
//
// Decomposition: levels above `cut` are split into P row slabs, slab g owning rows
// [g N/P, (g+1) N/P) (the last one also the boundary row N); coarse row I sits on fine row 2I, so
// ownership nests across levels.  Levels <= cut are solved redundantly by every slab's device
// from the gathered right-hand side ("the bottom solve stays on one GPU": nothing below the cut
// is ever distributed, and no scatter is needed because every slab holds the correction).
//
// Communication plan (halo messages are latency-bound, so send few):
//   * deep halos: a block of mu sweeps consumes mu halo rows (2 mu for red-black GS) exchanged
//     ONCE; the passes then update ranges that shrink towards the owned rows;
//   * redundant rows instead of messages: pre-smoothing leaves the result valid on the owned rows
//     plus the rows the restriction (2) and the post-smoothing will read, and the correction is
//     applied on that extended range too: neither needs an exchange;
//   * deep correction halos: every level below the finest leaves its post-smoothed correction
//     valid as far into its halos as the level above reads it when prolongating;
//   * left per cycle: ONE exchange of u on the finest level and ONE of the restricted right-hand
//     side per further distributed level, one all-gather at the cut, one all-reduce for ||r||^2.
#pragma once

#include "../../include/mgx.h"

#include <algorithm>
#include <string>
#include <vector>

namespace mgx {

struct DistPlanCfg {
    int finest = 0, cut = 0, coarsest = 0;
    int mu0 = 0, mu1 = 0, mu2 = 0;
    int smoother = MGX_SMOOTHER_JACOBI;
    int P = 1, g = 0;          // number of slabs, this slab
    bool fold = true;          // transfers folded into the smoother passes (mgx_slab_cycle)
    bool deep = true;          // deep correction halos
};

class DistPlanner {
public:
    DistPlanCfg c;
    std::vector<mgx_dist_level> geom;     // index = level - (cut + 1)
    int c_rows = 0, c_row0 = 0;           // this slab's share of the cut level's rows
    std::string err;
    // dynamic state: how many halo rows of u currently hold the neighbours' values, per level
    std::vector<int> u_halo;
    bool sumsq_ready = false;             // the last post-smoothing pass left this slab's sum of r^2

    int per() const { return c.smoother == MGX_SMOOTHER_RBGS ? 2 : 1; }
    mgx_dist_level& L(int level) { return geom[level - (c.cut + 1)]; }
    const mgx_dist_level& L(int level) const { return geom[level - (c.cut + 1)]; }
    int& uh(int level) { return u_halo[level - (c.cut + 1)]; }
    static int upd_lo(const mgx_dist_level& l) { return std::max(l.own_lo, 1) - l.row0; }      // first owned unknown row (local)
    static int upd_hi(const mgx_dist_level& l) { return std::min(l.own_hi, l.N) - l.row0; }    // one past the last (local)

    // smallest cut level for which every distributed level's slabs are thick enough for their halos
    static int default_cut(int finest, int coarsest, int P, int mu1, int mu2, int smoother)
    {
        // levels up to 1024^2 are replicated: below that a replicated level (launch-bound, ~10 us a
        // pass) costs less than the exchange a distributed one needs; 2048^2 and up are distributed
        int cut = std::max(coarsest, std::min(finest - 1, 10));
        for (; cut < finest - 1; ++cut) {
            DistPlanner p;
            DistPlanCfg cfg;
            cfg.finest = finest; cfg.cut = cut; cfg.coarsest = std::min(coarsest, cut); cfg.mu1 = mu1; cfg.mu2 = mu2;
            cfg.smoother = smoother; cfg.P = P; cfg.g = 0;
            if (p.init(cfg) == MGX_OK && p.c.deep) break;
        }
        return cut;
    }

    int init(const DistPlanCfg& cfg)
    {
        c = cfg;
        if (c.P < 1 || c.g < 0 || c.g >= c.P || c.finest <= c.cut || c.cut < c.coarsest || c.cut < 2 || c.mu1 < 0 || c.mu2 < 0) {
            err = "invalid slab decomposition";
            return MGX_ERR_INVALID;
        }
        plan_halos();
        if (c.deep) {
            bool thin = false;
            for (int l = c.cut + 1; l <= c.finest; ++l) thin = thin || ((1 << l) / c.P < halo_of[l - (c.cut + 1)]);
            if (thin) { c.deep = false; plan_halos(); }        // slabs too thin for the deep halos: exchange the correction
        }
        geom.clear();
        for (int l = c.cut + 1; l <= c.finest; ++l) {
            const int N = 1 << l, halo = halo_of[l - (c.cut + 1)];
            if (N % c.P || N / c.P < halo) {
                err = "level " + std::to_string(l) + ": " + std::to_string(N) + " rows cannot be split over " + std::to_string(c.P) +
                      " slabs with a " + std::to_string(halo) + "-row halo; raise cut_level";
                return MGX_ERR_INVALID;
            }
            mgx_dist_level g;
            g.level = l; g.N = N; g.halo = halo;
            g.own_lo = c.g * (N / c.P);
            g.own_hi = (c.g + 1) * (N / c.P) + (c.g == c.P - 1 ? 1 : 0);
            g.row0 = std::max(g.own_lo - halo, 0);
            g.rows = std::min(g.own_hi + halo, N + 1) - g.row0;
            geom.push_back(g);
        }
        const int Nc = 1 << c.cut;
        if (Nc % c.P) { err = "cut level rows must divide evenly over the slabs"; return MGX_ERR_INVALID; }
        c_rows = Nc / c.P;
        c_row0 = c.g * c_rows;
        u_halo.assign(geom.size(), 0);
        sumsq_ready = false;
        return MGX_OK;
    }

    int halo(int level) const { return halo_of[level - (c.cut + 1)]; }

    // the caller has (re)filled u of the finest level on ALL rows of the slab, halos included
    void guess_set() { uh(c.finest) = L(c.finest).halo; sumsq_ready = false; }
    // the caller changed u on the owned rows only
    void guess_changed() { uh(c.finest) = 0; sumsq_ready = false; }

    // ---- one V-cycle from `level` down (PS:575-627) -------------------------------------------
    void emit_vcycle(std::vector<mgx_dist_op>& out, int level = -1)
    {
        const int l = level < 0 ? c.finest : level;
        mgx_dist_level& Lv = L(l);
        sumsq_ready = false;
        const int NC = Lv.N / 2;
        const int glo = std::max(c.g * (NC / c.P), 1), ghi = std::min((c.g + 1) * (NC / c.P), NC);   // coarse rows produced (global)
        const int i = l - (c.cut + 1);
        const int e_post = ext_post[i], e_keep = ext_keep[i], e_coarse = ext_coarse[i], k_post = keep_post[i];
        const bool to_cut = !(l - 1 > c.cut);
        const int crow0 = to_cut ? c_row0 : L(l - 1).row0;
        const bool fold_pre = c.fold && c.mu1 > 0, fold_post = c.fold && c.mu2 > 0;
        // ---- pre-smoothing (PS:581) + residual, restriction, zero coarse guess (PS:604-613) ----
        if (fold_pre) {
            if (uh(l) < per() * c.mu1 + e_keep) { exchange(out, l, MGX_VEC_U, Lv.halo); uh(l) = Lv.halo; }
            if (!to_cut) out.push_back(op(MGX_DOP_ZERO_U, l - 1));                                  // PS:613
            mgx_dist_op o = op(MGX_DOP_CYCLE, l);
            range(Lv, e_keep, &o.row_lo, &o.row_hi);
            o.mu = c.mu1; o.post = 1; o.coarse_is_cut = to_cut ? 1 : 0;
            o.crow_lo = glo - crow0; o.crow_hi = ghi - crow0;
            out.push_back(o);
            uh(l) = e_keep;
        } else {
            if (c.mu1 > 0) smooth(out, l, c.mu1, e_keep);                                           // PS:581
            else if (uh(l) < e_keep) { exchange(out, l, MGX_VEC_U, Lv.halo); uh(l) = Lv.halo; }
            if (!to_cut) out.push_back(op(MGX_DOP_ZERO_U, l - 1));                                  // PS:613
            mgx_dist_op o = op(MGX_DOP_RESTRICT, l);                                                // PS:604-611
            o.coarse_is_cut = to_cut ? 1 : 0;
            o.crow_lo = glo - crow0; o.crow_hi = ghi - crow0;
            out.push_back(o);
        }
        // ---- coarse-grid correction (PS:617) -----------------------------------------------------
        if (!to_cut) {
            mgx_dist_level& Cl = L(l - 1);
            exchange(out, l - 1, MGX_VEC_B, Cl.halo);
            uh(l - 1) = Cl.halo;                              // zeros are exact halo values
            emit_vcycle(out, l - 1);
            sumsq_ready = false;
            if (uh(l - 1) < e_coarse) { exchange(out, l - 1, MGX_VEC_U, e_coarse); uh(l - 1) = e_coarse; }   // never with deep halos
        } else {
            out.push_back(op(MGX_DOP_GATHER_CUT, c.cut));
            out.push_back(op(MGX_DOP_COARSE, c.cut));
        }
        // ---- correction (PS:620-624) + post-smoothing (PS:625) (+ the residual norm) ------------
        uh(l) = std::min(uh(l), e_post);
        if (fold_post) {
            mgx_dist_op o = op(MGX_DOP_CYCLE, l);
            range(Lv, k_post, &o.row_lo, &o.row_hi);
            o.mu = c.mu2; o.pre = 1; o.post = (l == c.finest) ? 2 : 0; o.coarse_is_cut = to_cut ? 1 : 0;
            out.push_back(o);
            uh(l) = k_post;
            sumsq_ready = (l == c.finest);
        } else {
            mgx_dist_op o = op(MGX_DOP_PROLONG, l);                                                 // PS:620-624
            range(Lv, e_post, &o.row_lo, &o.row_hi);
            o.coarse_is_cut = to_cut ? 1 : 0;
            out.push_back(o);
            if (c.mu2 > 0) smooth(out, l, c.mu2, k_post);                                           // PS:625
            else uh(l) = std::min(uh(l), k_post);
        }
    }

    // ---- fullmultigrid (PS:629-650) on the slab hierarchy ------------------------------------------
    // PS:641: the right-hand side of every coarser level is the restriction of the finer one; the levels
    // <= cut run fullmultigrid themselves on the gathered right-hand side (replicated); then, level by
    // level, the interpolated solution (PS:645) and mu0 + 1 V-cycles (PS:646-648).  The finest
    // right-hand side must hold the global field on every row of the slab, halos included.
    void emit_fmg(std::vector<mgx_dist_op>& out)
    {
        sumsq_ready = false;
        for (int l = c.finest; l > c.cut; --l) {
            mgx_dist_level& Lv = L(l);
            const int NC = Lv.N / 2;
            const int glo = std::max(c.g * (NC / c.P), 1), ghi = std::min((c.g + 1) * (NC / c.P), NC);
            const bool to_cut = !(l - 1 > c.cut);
            const int crow0 = to_cut ? c_row0 : L(l - 1).row0;
            mgx_dist_op o = op(MGX_DOP_RESTRICT_RHS, l);                       // PS:641
            o.coarse_is_cut = to_cut ? 1 : 0;
            o.crow_lo = glo - crow0; o.crow_hi = ghi - crow0;
            out.push_back(o);
            if (!to_cut) exchange(out, l - 1, MGX_VEC_B, L(l - 1).halo);        // its restriction and its V-cycles read the halos
        }
        out.push_back(op(MGX_DOP_GATHER_CUT, c.cut));
        out.push_back(op(MGX_DOP_COARSE_FMG, c.cut));
        for (int l = c.cut + 1; l <= c.finest; ++l) {
            mgx_dist_level& Lv = L(l);
            const bool to_cut = !(l - 1 > c.cut);
            if (!to_cut && uh(l - 1) < 1) { exchange(out, l - 1, MGX_VEC_U, L(l - 1).halo); uh(l - 1) = L(l - 1).halo; }
            mgx_dist_op o = op(MGX_DOP_PROLONG_SET, l);                         // PS:645
            o.row_lo = upd_lo(Lv); o.row_hi = upd_hi(Lv);
            o.coarse_is_cut = to_cut ? 1 : 0;
            out.push_back(o);
            uh(l) = 0;                                                          // owned rows only: the first V-cycle exchanges
            for (int i = 0; i <= c.mu0; ++i) emit_vcycle(out, l);               // PS:646-648
        }
    }

    // ---- ||b - A u||_2 on the finest level ------------------------------------------------------
    void emit_norm(std::vector<mgx_dist_op>& out)
    {
        mgx_dist_level& Lv = L(c.finest);
        if (!sumsq_ready) {
            if (uh(c.finest) < 1) { exchange(out, c.finest, MGX_VEC_U, Lv.halo); uh(c.finest) = Lv.halo; }   // deep enough for the next pre-smoothing too
            mgx_dist_op o = op(MGX_DOP_SUMSQ, c.finest);
            o.row_lo = upd_lo(Lv); o.row_hi = upd_hi(Lv);
            out.push_back(o);
        }
        sumsq_ready = false;
        out.push_back(op(MGX_DOP_ALLREDUCE_NORM, c.finest));
    }

private:
    std::vector<int> keep_post, ext_post, ext_keep, ext_coarse, halo_of;

    // Halo plan per level, from the finest level down.  keep_post: rows beyond the owned ones the
    // post-smoothing leaves valid - 0 on the finest level; on every level below, as many as the
    // level above reads when it prolongates the correction (ext_coarse), so that the correction needs
    // NO exchange (deep = false: it is exchanged, and keep_post is 0 everywhere).
    void plan_halos()
    {
        const int nl = c.finest - c.cut;
        keep_post.assign(nl, 0); ext_post.assign(nl, 0); ext_keep.assign(nl, 0); ext_coarse.assign(nl, 0); halo_of.assign(nl, 0);
        int kp = 0;
        for (int l = c.finest; l > c.cut; --l) {
            const int i = l - (c.cut + 1);
            keep_post[i] = kp;
            // halo rows the post-smoothing consumes (+1 on the finest level: the folded norm needs the
            // result one row beyond the owned rows)
            const int ep = per() * c.mu2 + kp + ((c.fold && l == c.finest) ? 1 : 0);
            int ek = std::max(ep, 2);                  // rows beyond the owned ones pre-smoothing leaves valid
            if (c.fold) ek |= 1;                       // the folded restriction wants its range to start on an odd row
            const int ec = ep / 2 + 2;                 // coarse halo rows the extended prolongation reads
            ext_post[i] = ep; ext_keep[i] = ek; ext_coarse[i] = ec;
            halo_of[i] = std::max(std::max(per() * c.mu1 + ek, ec), kp);
            kp = c.deep ? ec : 0;
        }
    }

    static mgx_dist_op op(int code, int level)
    {
        mgx_dist_op o{};
        o.op = code; o.level = level;
        return o;
    }
    // owned unknown rows widened by `ext` rows into the halos (local indices)
    static void range(const mgx_dist_level& l, int ext, int* lo, int* hi)
    {
        const int first = 1 - l.row0, last = l.N - l.row0;
        *lo = std::max(upd_lo(l) - ext, first);
        *hi = std::min(upd_hi(l) + ext, last);
    }
    void exchange(std::vector<mgx_dist_op>& out, int level, int which, int depth)
    {
        if (c.P == 1 || depth <= 0) return;
        mgx_dist_op o = op(MGX_DOP_EXCHANGE, level);
        o.which = which; o.depth = depth;
        out.push_back(o);
    }
    // mu sweeps whose result is valid on the owned rows +- keep
    void smooth(std::vector<mgx_dist_op>& out, int l, int mu, int keep)
    {
        mgx_dist_level& Lv = L(l);
        if (uh(l) < per() * mu + keep) { exchange(out, l, MGX_VEC_U, Lv.halo); uh(l) = Lv.halo; }
        mgx_dist_op o = op(MGX_DOP_SMOOTH, l);
        range(Lv, keep, &o.row_lo, &o.row_hi);
        o.mu = mu;
        out.push_back(o);
        uh(l) = keep;
    }
};

} // namespace mgx
