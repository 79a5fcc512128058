"""When did every wave of a deep folded pass start and end?  Needs the debug build
(make -C multigrid_nikhil_c-_amd/csrc trace -> libmgx_trace.so).  Runs a few V(10,10) cycles with the level
given as the finest one, then reads the trace of the LAST k_jacobi_cycle launch (the finest level's
post-smoothing pass) and prints how the wave durations and end times are distributed.

    MGX_LIBMGX_PATH=$PWD/multigrid_nikhil_c-_amd/libmgx_trace.so python tools/wave_trace.py 12
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as ge

pkg = ge.load_package()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 12


class WT(C.Structure):
    _fields_ = [("t0", C.c_longlong), ("t1", C.c_longlong), ("strip", C.c_int), ("r0", C.c_int), ("r1", C.c_int), ("hw", C.c_int)]


with pkg.Multigrid(finest_level=L, coarsest_level=L - 1 if L <= 8 else 7, mu0=0, mu1=10, mu2=10, schedule=0, profile=1,
                   arith=pkg.ARITH_FMA) as mg:
    mg.fill_rhs(1, 0.0)
    mg.fill_guess_random(1)
    mg.solve(tol=0.0, max_cycles=3)
    mg.synchronize()
    lib = pkg.binding.lib()
    buf = (WT * 65536)()
    lib.mgx_debug_wave_trace.restype = C.c_int
    n = lib.mgx_debug_wave_trace(C.byref(buf), 65536)
a = np.array([(w.t0, w.t1, w.strip, w.r0, w.r1, w.hw) for w in buf[:n]], dtype=np.int64)
act = a[a[:, 2] >= 0]
t0 = act[:, 0].min()
start = (act[:, 0] - t0) / 100.0          # us (100 MHz)
end = (act[:, 1] - t0) / 100.0
dur = end - start
rows = act[:, 4] - act[:, 3]
print(f"L{L}: {len(act)} active waves of {n}; kernel span {end.max():.1f} us")
# how full were the 2048 wave slots (256 CUs x 4 SIMDs x 2 waves) over the launch?
span = end.max()
print(f"  slot occupancy: sum of wave durations / (2048 slots x span) = {dur.sum() / (2048 * span):.3f}; "
      f"resident waves at 5 % steps of the span: " +
      " ".join(str(int(((start <= t) & (end > t)).sum())) for t in np.linspace(0.025, 0.975, 20) * span))
steps = rows + 23
print(f"  us per row step by start time: first round (start < 5 us) {np.median((dur / steps)[start < 5]):.3f}, later {np.median((dur / steps)[start >= 5]):.3f}")
print(f"  wave start: min {start.min():.1f}  median {np.median(start):.1f}  p90 {np.percentile(start, 90):.1f}  max {start.max():.1f} us")
print(f"  wave end  : min {end.min():.1f}  median {np.median(end):.1f}  p90 {np.percentile(end, 90):.1f}  max {end.max():.1f} us")
print(f"  duration  : min {dur.min():.1f}  median {np.median(dur):.1f}  p90 {np.percentile(dur, 90):.1f}  max {dur.max():.1f} us")
S = act[:, 2].max() + 1
edge = (act[:, 2] == 0) | (act[:, 2] == S - 1)
for name, m in (("interior strips", ~edge), ("edge strips", edge)):
    if m.any():
        print(f"  {name}: {m.sum()} waves, rows {np.median(rows[m]):.0f} (median), duration median {np.median(dur[m]):.1f}  max {dur[m].max():.1f}, us per row step {np.median(dur[m] / (rows[m] + 23)):.3f}")
# by XCD and by how many waves shared the SIMD is not in HW_ID; print duration by XCD (hw id bits: see ISA) and by start decile
xcc = (act[:, 5] >> 16) & 0xF if False else None
q = np.argsort(start)
for i in range(0, 10):
    sel = q[i * len(q) // 10:(i + 1) * len(q) // 10]
    print(f"  start decile {i}: start {start[sel].mean():7.1f}  dur {dur[sel].mean():7.1f}  end {end[sel].mean():7.1f}")
late = np.argsort(end)[-12:]
print("  last waves to end: " + ", ".join(f"(strip {act[i, 2]}, rows {act[i, 3]}..{act[i, 4]}, start {start[i]:.0f}, dur {dur[i]:.0f})" for i in late))

# ---- where did the slow waves run?  HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
hw = act[:, 5]
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
key = (se * 2 + sh) * 16 + cu            # CU within its XCD (the XCC id is in another register)
first = start < 5.0                      # first-round waves
import collections
per_cu = collections.defaultdict(list)
for k, sm, d0, f in zip(key, simd, dur, first):
    if f:
        per_cu[(int(k), int(sm))].append(d0)
n_on_simd = np.array([len(v) for v in per_cu.values()])
mean_on_simd = np.array([np.mean(v) for v in per_cu.values()])
print(f"  first-round waves per (CU-in-XCD, SIMD) slot group (8 XCDs share each key): counts {np.bincount(n_on_simd)[:24]}")
q = np.argsort(dur[first])
d1 = dur[first]
print(f"  first-round duration deciles: {np.percentile(d1, [0, 10, 25, 50, 75, 90, 100]).round(1)}")
for name, arr in (("simd", simd), ("se", se), ("cu", cu)):
    vals = [f"{v}:{np.median(dur[first & (arr == v)]):.0f}" for v in np.unique(arr)]
    print(f"  median first-round duration by {name}: " + " ".join(vals))
xcc = (hw >> 16) & 15
print("  first-round duration by XCC (median / max / waves): " +
      " ".join(f"{v}:{np.median(dur[first & (xcc == v)]):.0f}/{dur[first & (xcc == v)].max():.0f}/{(first & (xcc == v)).sum()}" for v in np.unique(xcc)))
print("  waves per SIMD by XCC (min..max over its SIMDs): " +
      " ".join(f"{v}:{np.bincount((key * 4 + simd)[first & (xcc == v)]).min()}..{np.bincount((key * 4 + simd)[first & (xcc == v)]).max()}" for v in np.unique(xcc)))
# by tile position
r0s = act[:, 3]
print("  median first-round duration by chunk row (first 12 / last 6): " +
      " ".join(f"{int(r)}:{np.median(dur[first & (r0s == r)]):.0f}" for r in list(np.unique(r0s[first]))[:12] + list(np.unique(r0s[first]))[-6:]))
# ---- the two first-round waves of one SIMD: how long does each take?
pairs = collections.defaultdict(list)
for i in np.nonzero(first)[0]:
    pairs[(int(xcc[i]), int(key[i]), int(simd[i]))].append(i)
two = [v for v in pairs.values() if len(v) == 2]
if two:
    sh_ = np.array([min(dur[a], dur[b]) for a, b in two]); lo_ = np.array([max(dur[a], dur[b]) for a, b in two])
    st_ = np.array([steps[a] for a, b in two], dtype=float)
    print(f"  SIMDs with exactly two first-round waves: {len(two)}; shorter wave {np.median(sh_):.1f} us, longer {np.median(lo_):.1f} us "
          f"(median row steps {np.median(st_):.0f}: {np.median(sh_ / st_):.3f} / {np.median(lo_ / st_):.3f} us per step)")
    lone = [v for v in pairs.values() if len(v) == 1]
    if lone:
        d_ = np.array([dur[v[0]] for v in lone]); s_ = np.array([steps[v[0]] for v in lone], dtype=float)
        print(f"  SIMDs with ONE first-round wave: {len(lone)}; {np.median(d_):.1f} us, {np.median(d_ / s_):.3f} us per step")
    # which of the two is the faster one: the lower block index?  how far apart are the two blocks of a SIMD?
    widx = np.nonzero(a[:, 2] >= 0)[0]          # wave number = 4 * blockIdx + wave in block
    blk = widx // 4
    fa = np.array([(a_ if dur[a_] <= dur[b_] else b_) for a_, b_ in two]); sl = np.array([(b_ if dur[a_] <= dur[b_] else a_) for a_, b_ in two])
    d_blk = blk[sl] - blk[fa]
    print(f"  faster wave has the LOWER block index in {np.mean(d_blk > 0):.3f} of the pairs; block distance slow - fast: "
          f"median {np.median(d_blk):.0f}, p10 {np.percentile(d_blk, 10):.0f}, p90 {np.percentile(d_blk, 90):.0f}; "
          f"starts: fast {np.median(start[fa]):.2f} us, slow {np.median(start[sl]):.2f} us")
    print(f"  block index of the faster waves: p5 {np.percentile(blk[fa], 5):.0f} median {np.median(blk[fa]):.0f} p95 {np.percentile(blk[fa], 95):.0f}; "
          f"of the slower: p5 {np.percentile(blk[sl], 5):.0f} median {np.median(blk[sl]):.0f} p95 {np.percentile(blk[sl], 95):.0f}")
    hwslot = hw & 15
    print(f"  hardware wave slot of the faster: {np.bincount(hwslot[fa], minlength=10)[:10]}, of the slower: {np.bincount(hwslot[sl], minlength=10)[:10]}")
# ---- by chunk height (the paired geometry of csrc/mgx_geom.hpp: tall chunks on the workgroups dispatched first, short ones on
# the rest, edge-class chunks in between): when does each class end?
print("  by chunk height (rows: waves, median start / end us, us per row step, share on hardware wave slot 0):")
hs, cnt = np.unique(rows, return_counts=True)
for h in hs[np.argsort(-cnt)][:6]:
    m = rows == h
    print(f"    {int(h):4d} rows: {int(m.sum()):5d} waves, start {np.median(start[m]):6.1f}, end {np.median(end[m]):6.1f} (p10 {np.percentile(end[m], 10):6.1f}, p90 {np.percentile(end[m], 90):6.1f}),"
          f" {np.median(dur[m] / (rows[m] + 23)):.3f} us/step, slot 0: {np.mean((hw[m] & 15) == 0):.2f}")
