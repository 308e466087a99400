"""How far can the simulated Jitterbug travel in one 10 s episode under simple open-loop pulsed motor commands?
(The reference's fig-heatmap shows trained MuJoCo policies travelling 2+ m per episode.)"""
import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv
periods = [1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 50, 100, 10**6]
amps = [1.0, 0.6]
biases = [0.0, 0.4]
reps = 32
cfg = [(p, a, b) for p in periods for a in amps for b in biases]
n = len(cfg) * reps
P = np.repeat(np.array([c[0] for c in cfg]), reps); A = np.repeat(np.array([c[1] for c in cfg]), reps); B = np.repeat(np.array([c[2] for c in cfg]), reps)
env = JitterbugVecEnv(n, "move_from_origin", seed=0, auto_reset=False)
env.reset()
path = np.zeros(n); last = None
for t in range(999):
    u = np.clip(B + A * np.where((t // P) % 2 == 0, 1.0, -1.0), -1, 1).astype(np.float32)
    env.step(u)
q, v, _ = env.get_state()
d = np.hypot(q[:, 0], q[:, 1]); up = 1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)
for i, c in enumerate(cfg):
    s = slice(i * reps, (i + 1) * reps)
    print("half-period %7d steps amp %.1f bias %.1f : displacement after 10 s mean %.3f m max %.3f m, upright %.2f" % (c[0], c[1], c[2], d[s].mean(), d[s].max(), (up[s] > 0.9).mean()))
