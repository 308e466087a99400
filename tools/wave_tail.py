"""Diagnostic (needs a -DJB_WAVE_STATS build, JITTERBUG_HIP_LIB=...): where the LAUNCH time of jb_step_kernel goes in the
steady state of the benchmark workload - per sampled control step the slowest wave, the mean wave, and what the slowest
waves were doing (all-geom path or not, Newton sweeps).  python tools/wave_tail.py [n] [first] [last] [every] [const]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv
from jitterbug_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
first = int(sys.argv[2]) if len(sys.argv) > 2 else 100
last = int(sys.argv[3]) if len(sys.argv) > 3 else 400
every = int(sys.argv[4]) if len(sys.argv) > 4 else 10
const = len(sys.argv) > 5 and sys.argv[5] == "const"
env = JitterbugVecEnv(n, "move_from_origin", seed=0)
env.reset()
rng = np.random.default_rng(0)
L = _lib.load()
L.jb_debug_wave_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
rows = []
cls_acc = {}
heavy_log = []
NAMES = ["foot"] + ["lc%d" % i for i in range(4)] + ["uc%d" % i for i in range(4)] + ["tip"] + ["rcyl%d" % i for i in range(4)] + ["rell"] + ["box%d" % i for i in range(8)] + ["mcyl%d" % i for i in range(4)] + ["mell"]
prev_hist = None
for t in range(last):
    env.step((np.ones(n) if const else rng.uniform(-1, 1, size=n)).astype(np.float32))
    if t < first or (t - first) % every:
        continue
    big = np.zeros((n * 5, 16), dtype=np.uint64)
    e = L.jb_debug_wave_stats(env._h, big.ctypes.data, n * 5)
    buf = big[:n]
    nw = (n + e - 1) // e
    hist = big[n:].reshape(-1, 64)[:nw].astype(np.float64)          # per-wave cumulative live-slot counts over all-geom substeps
    dh = hist if prev_hist is None else hist - prev_hist
    prev_hist = hist
    b = buf[:nw].astype(np.float64)
    cyc = b[:, 0] / 50.0                 # cycles per substep
    xt = b[:, 1] > 0
    order = np.argsort(-cyc)
    top = order[:3]
    rows.append((t, cyc.mean(), np.median(cyc), np.quantile(cyc, .99), cyc.max(), int(xt.sum()), bool(xt[order[0]]),
                 cyc[~xt].max(), cyc[xt].mean() if xt.any() else 0.0, cyc[xt].max() if xt.any() else 0.0))
    for i in order[:3]:
        if xt[i] and prev_hist is not None:
            heavy_log.append((t, cyc[i], b[i, 9] / 50, b[i, 2] / 50, b[i, 15] / 50, b[i, 11] / max(b[i, 3], 1), b[i, 1],
                              " ".join("%s" % NAMES[k] for k in range(28) if hist[i, k] > 0 and (dh[i, k] > 0 or every > 1))))
    for name, sel in (("ordinary", ~xt), ("all-geom", xt)):
        if sel.any():
            a = cls_acc.setdefault(name, [])
            a.append(np.concatenate([[sel.sum(), cyc[sel].mean()], b[sel, 4:9].mean(0) / 50, b[sel, 12:15].mean(0) / 50, [b[sel, 2].mean() / 50, (b[sel, 11] / np.maximum(b[sel, 3], 1)).mean()]]))
r = np.array([x[1:] for x in rows], dtype=float)
print("step   mean  median   p99    max  #allgeom  max_is_allgeom  max_ordinary  mean_allgeom  max_allgeom   (cycles per substep)")
for x in rows:
    print("%4d %6.0f %6.0f %6.0f %6.0f   %4d      %s      %6.0f   %6.0f   %6.0f" % x)
print("AVERAGE over %d sampled steps: mean wave %.0f, slowest wave %.0f (ratio %.3f); slowest ORDINARY wave %.0f (ratio to mean %.3f); steps whose slowest wave was all-geom: %.2f"
      % (len(rows), r[:, 0].mean(), r[:, 3].mean(), r[:, 0].mean() / r[:, 3].mean(), r[:, 6].mean(), r[:, 6].mean() / r[:, 0].mean(), r[:, 5].mean()))
for name, a in cls_acc.items():
    a = np.array(a).mean(0)
    print("%-9s waves/step %.1f  cycles/substep %.0f : A %.0f check %.0f full %.0f solve-tail %.0f integrate %.0f | rows-build %.0f star-solves %.0f logic %.0f | full sweeps/substep %.2f live slots/contact substep %.2f"
          % ((name,) + tuple(a)))

# what the slow ORDINARY waves of the last sampled step did, against the batch: Newton checks, full sweeps, chained rank-one passes
b = buf[:nw].astype(np.float64); cyc = b[:, 0] / 50.0; xt = b[:, 1] > 0
X = np.stack([np.ones(nw), b[:, 9] / 50, b[:, 2] / 50, b[:, 15] / 50, b[:, 11] / np.maximum(b[:, 3], 1), b[:, 3] / 50], 1)[~xt]
coef = np.linalg.lstsq(X, cyc[~xt], rcond=None)[0]
print("ordinary waves, last step: cycles/substep = %.0f + %.0f*checks + %.0f*fulls + %.0f*chain passes + %.0f*live slots + %.0f*contact fraction   (per substep)" % tuple(coef))
print("  batch means: checks %.2f fulls %.2f chain passes %.2f live slots %.2f contact fraction %.2f" % tuple(X[:, 1:].mean(0)))
o = np.argsort(-np.where(xt, 0, cyc))[:10]
print("  slowest ordinary waves [cycles/substep | checks fulls passes live contact | A check full solve-tail integrate | rows star logic]:")
for i in o:
    print("   %6.0f | %.2f %.2f %.2f %.2f %.2f |" % (cyc[i], b[i, 9] / 50, b[i, 2] / 50, b[i, 15] / 50, b[i, 11] / max(b[i, 3], 1), b[i, 3] / 50), np.round(b[i, 4:9] / 50), np.round(b[i, 12:15] / 50))

print("slow all-geom waves [step | cycles/substep | checks fulls rank-one-passes per substep | live slots | all-geom substeps | slots ever live in this wave's all-geom substeps]:")
for h in heavy_log[-40:]:
    print("  %4d | %6.0f | %.2f %.2f %.2f | %.2f | %2d | %s" % h)
