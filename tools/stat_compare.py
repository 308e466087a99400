"""Open-loop distribution comparison, GPU vs fp64 oracle (test infrastructure; run on the GPU box)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd import model
from jitterbug_amd.vec_env import JitterbugVecEnv
from oracle import oracle as O
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 250
P = model.default_params()
g = JitterbugVecEnv(n, "move_from_origin", seed=12)
o = O.OracleEnv(n, "move_from_origin", P, seed=12)
g.reset(), o.reset()
rng = np.random.default_rng(3)
rg, ro = np.zeros(n), np.zeros(n)
for t in range(steps):
    a = rng.uniform(-1, 1, size=n).astype(np.float32)
    og, r1, _, _ = g.step(a)
    oo, r2, _ = o.step(a)
    rg += r1; ro += r2
    if t in (0, 1, 2, 4, 9, 19, 49, 99):
        d = np.abs(og - oo).max(axis=1)
        print("step %3d: max|dobs| median %.2e p90 %.2e p99 %.2e" % (t + 1, np.median(d), np.quantile(d, .9), np.quantile(d, .99)), flush=True)
qg, vg, _ = g.get_state(); qo, vo, _ = o.get_state()
def z(x, y): return (x.mean() - y.mean()) / np.sqrt(x.var() / len(x) + y.var() / len(y))
dg, do = np.hypot(qg[:, 0], qg[:, 1]), np.hypot(qo[:, 0], qo[:, 1])
print("return     gpu %.3f oracle %.3f z %.2f" % (rg.mean(), ro.mean(), z(rg, ro)))
print("displace   gpu %.5f oracle %.5f z %.2f" % (dg.mean(), do.mean(), z(dg, do)))
print("z          gpu %.5f oracle %.5f z %.2f" % (qg[:, 2].mean(), qo[:, 2].mean(), z(qg[:, 2], qo[:, 2])))
sg, so = np.linalg.norm(vg[:, :3], axis=1), np.linalg.norm(vo[:, :3], axis=1)
print("speed      gpu %.5f oracle %.5f z %.2f" % (sg.mean(), so.mean(), z(sg, so)))
print("motor rate gpu %.3f oracle %.3f z %.2f" % (vg[:, 14].mean(), vo[:, 14].mean(), z(vg[:, 14], vo[:, 14])))
print("paired: corr(return) %.3f  corr(displacement) %.3f" % (np.corrcoef(rg, ro)[0, 1], np.corrcoef(dg, do)[0, 1]))
