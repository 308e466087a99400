"""Diagnostic: per-wave cycle counts of jb_step_kernel (needs a -DJB_WAVE_STATS build)."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv
from jitterbug_amd import _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
epw = int(sys.argv[2]) if len(sys.argv) > 2 else 0
env = JitterbugVecEnv(n, "move_from_origin", seed=0, envs_per_wave=epw)
env.reset()
rng = np.random.default_rng(0)
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 150
for t in range(nsteps):
    env.step((np.ones(n) if (len(sys.argv) > 4 and sys.argv[4] == 'const') else rng.uniform(-1, 1, size=n)).astype(np.float32))
L = _lib.load()
buf = np.zeros((n * 5, 16), dtype=np.uint64)
L.jb_debug_wave_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
e = L.jb_debug_wave_stats(env._h, buf.ctypes.data, n * 5)
hist = buf[n:].reshape(-1, 64).sum(axis=0)
nw = (n + e - 1) // e
b = buf[:nw].astype(np.float64)
cyc = b[:, 0] / 100.0   # shader cycles / 100
rt = b[:, 10] / 100.0   # s_memrealtime ticks (100 MHz) -> microseconds
print('real time us: mean %.1f max %.1f ; clock MHz %.0f' % (rt.mean(), rt.max(), (b[:,0].sum()/b[:,10].sum())*100))
ph = b[:, 4:9]
print('phase cycles per substep: A %.0f check %.0f full %.0f solve %.0f integrate %.0f ; sum/total %.2f' % (*(ph.mean(0)/50), ph.sum()/b[:,0].sum()))
print("  finer: rows build + plan %.0f, star_solve calls %.0f, final-pass rhs + loop logic before the solve %.0f  (cycles per substep)" % tuple(b[:, 12:15].mean(0) / 50))
print("rank-one passes per step (wave mean): %.1f" % b[:, 15].mean())
print("epw", e, "waves", nw)
print("wave time us: mean %.1f median %.1f p90 %.1f p99 %.1f max %.1f" % (cyc.mean(), np.median(cyc), np.quantile(cyc, .9), np.quantile(cyc, .99), cyc.max()))
print("rare-path substeps/50: mean %.2f max %d  frac waves with any %.3f" % (b[:, 1].mean(), b[:, 1].max(), (b[:, 1] > 0).mean()))
print("newton sweeps per step: mean %.1f max %d ; contact substeps mean %.1f" % (b[:, 2].mean(), b[:, 2].max(), b[:, 3].mean()))
order = np.argsort(-cyc)[:8]
print("slowest waves [us, xtra, sweeps, contact]:", [(round(cyc[i], 1), int(b[i, 1]), int(b[i, 2]), int(b[i, 3])) for i in order])
c = np.corrcoef(cyc, b[:, 1])[0, 1]; c2 = np.corrcoef(cyc, b[:, 2])[0, 1]
print("corr(time, xtra) %.3f corr(time, sweeps) %.3f" % (c, c2))
# linear fit time ~ a + b*xtra + c*sweeps
A = np.stack([np.ones(nw), b[:, 1], b[:, 2], b[:, 3]], 1)
coef = np.linalg.lstsq(A, cyc, rcond=None)[0]
print("fit us: base %.1f + %.2f/xtra-substep + %.2f/sweep + %.2f/contact-substep" % tuple(coef))
print("phase cycles per substep by wave class (A, check, full, solve, integrate):")
for name, sel in (("no rare path", b[:, 1] == 0), ("rare path all 50 substeps", b[:, 1] == 50)):
    if sel.any():
        print("  %-28s n=%4d  total %.0f :" % (name, sel.sum(), b[sel, 0].mean() / 50), np.round(b[sel, 4:9].mean(0) / 50),
              "live slots per contact substep %.2f, full sweeps per substep %.2f" % ((b[sel, 11] / np.maximum(b[sel, 3], 1)).mean(), b[sel, 2].mean() / 50))
nx = b[:, 1] == 0
idx = np.argsort(-cyc * nx)[:12]
print("slowest waves without rare path [cycles/substep, full sweeps/substep, live slots/contact substep, phases A chk full solve int]:")
for i in idx:
    print("   %6.0f %5.2f %5.2f " % (b[i, 0] / 50, b[i, 2] / 50, b[i, 11] / max(b[i, 3], 1)), np.round(b[i, 4:9] / 50))
qs = np.quantile(b[nx, 0] / 50, [0.1, 0.5, 0.9, 0.99, 1.0])
print("cycles/substep quantiles (no rare path) p10 p50 p90 p99 max:", np.round(qs))
A = np.stack([np.ones(nx.sum()), b[nx, 2] / 50, b[nx, 11] / np.maximum(b[nx, 3], 1)], 1)
coef = np.linalg.lstsq(A, b[nx, 0] / 50, rcond=None)[0]
print("fit cycles/substep = %.0f + %.0f * fulls + %.0f * live_slots" % tuple(coef))

print("all-geom substeps: live-slot counts (cumulative over the run) by slot:")
names = ["foot"] + ["lowcyl%d" % i for i in range(4)] + ["upcyl%d" % i for i in range(4)] + ["tip"] + ["rcyl%d" % i for i in range(4)] + ["rell"] + ["box%d" % i for i in range(8)] + ["mcyl%d" % i for i in range(4)] + ["mell"]
tot = max(1, int(hist[28:36].sum()))
print("  " + "  ".join("%s %.2f" % (names[i], hist[i] / tot) for i in range(28)))
print("rounds histogram all-geom substeps:", (hist[28:36] / tot).round(3), " ordinary:", (hist[36:44] / max(1, hist[36:44].sum())).round(3))

if hist[44] > 0:
    print("checks that found a changed set (ordinary substeps): %d; of those with at most ONE flipped bit in every unconverged env: %.3f" % (hist[44], hist[45] / hist[44]))
    e = hist[46:50].astype(float)
    print("flipped bits per unconverged env [1, 2, 3, 4+]:", (e / max(1.0, e.sum())).round(3))
    multi = max(1.0, float(e[1:].sum()))
    print("multi-flip envs whose flips sit in different legs (at most one per lane): %.3f of the multi-flip envs (%.3f of the 2-flip ones); changed-set checks that need a full pass only because of such envs: %.3f of all, %.3f of those needing a full pass"
          % (hist[50] / multi, hist[51] / max(1.0, float(e[1])), hist[52] / hist[44], hist[52] / max(1.0, float(hist[44] - hist[45]))))
