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


with pkg.Multigrid(finest_level=L, coarsest_level=L - 1 if L <= 8 else 7, mu0=0, mu1=10, mu2=10, schedule=0, profile=1) as mg:
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
