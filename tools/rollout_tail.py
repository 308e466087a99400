"""Diagnostic (needs a -DJB_WAVE_STATS build: JITTERBUG_HIP_LIB=ab_build/libjb_ws.so): what bounds a FUSED K-step rollout - the waves
whose SUM over the K steps is largest - and where their cycles go, against the mean wave.

    JITTERBUG_HIP_LIB=ab_build/libjb_ws.so python tools/rollout_tail.py [n] [K] [uniform|const1]
"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv
from jitterbug_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
mode = sys.argv[3] if len(sys.argv) > 3 else "uniform"
augmented = len(sys.argv) > 4 and sys.argv[4] == "augmented"
dev = torch.device("cuda", 0)
env = JitterbugVecEnv(n, "move_to_pose" if augmented else "move_from_origin", seed=0)
if augmented:
    env.randomise_models(seed=1000, return_params=False)
env.reset_device()
g = torch.Generator(device=dev); g.manual_seed(1234)
tape = torch.rand((K, n), generator=g, device=dev, dtype=torch.float32) * 2 - 1
if mode == "const1":
    tape.fill_(1.0)
env.step_many_device(K, tape.data_ptr())
env.synchronize()
L = _lib.load()
L.jb_debug_wave_stats.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
big = np.zeros((n * 5, 16), dtype=np.uint64)
e = L.jb_debug_wave_stats(env._h, big.ctypes.data, n * 5)
nw = (n + e - 1) // e
b = big[:nw].astype(np.float64)
S = K * 50.0
cyc = b[:, 0] / S
print("fused rollout, %d envs, K = %d, %s actions, instrumented build: cycles per substep (s_memtime) mean %.0f median %.0f p99 %.0f max %.0f; mean/max %.3f" % (
    n, K, mode, cyc.mean(), np.median(cyc), np.quantile(cyc, .99), cyc.max(), cyc.mean() / cyc.max()))
hdr = "cycles/substep | A check-sweeps full-sweeps solve-tail integrate | rows-build star-solves logic | checks fulls rank-one per substep | live slots per contact substep | all-geom substep share"
def row(x):
    return "%6.0f | %5.0f %5.0f %5.0f %5.0f %5.0f | %5.0f %5.0f %5.0f | %.2f %.2f %.2f | %.2f | %.2f" % (
        x[0] / S, x[4] / S, x[5] / S, x[6] / S, x[7] / S, x[8] / S, x[12] / S, x[13] / S, x[14] / S, x[9] / S, x[2] / S, x[15] / S, x[11] / max(x[3], 1), x[1] / S)
print(hdr)
print("MEAN WAVE      ", row(b.mean(0)))
o = np.argsort(-cyc)
for i in o[:12]:
    print("wave %4d      " % i, row(b[i]))
print("p50 wave       ", row(b[o[nw // 2]]))
hist = big[n:].reshape(-1, 64)[:nw].astype(np.float64)
print("full sweeps by rounds [spread rounds 0 1 2 3+ | rest (non-spread) rounds 0 1 2 3+], share of the wave's full sweeps:")
for name, sel in (("MEAN WAVE", slice(None)),) + tuple(("wave %4d" % i, [i]) for i in o[:6]):
    h = hist[sel].sum(0)
    tot = max(h[53:57].sum(), 1)
    print("  %-10s | %s | %s" % (name, " ".join("%.3f" % (x / tot) for x in h[53:57]), " ".join("%.3f" % (x / tot) for x in h[57:61])))
