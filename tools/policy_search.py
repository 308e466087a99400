"""Bound the locomotion discrepancy: can ANY simple policy on this simulator reach what the reference's trained agents
reach on real MuJoCo - >= 0.1 m/s along the target direction (TARGET_SPEED, reference jitterbug.py:58) and an episode return >~ 900 on
move_in_direction / move_from_origin (manuscript/ICRA2020/root.tex:279, fig-rl-perf)?

Cross-entropy search, on the GPU simulator, over two parametrised policy families evaluated closed loop on the device:

  kick    the reference's own bang-bang structure (heuristic_policies.py:28-56) with free parameters, on DE-NORMALISED inputs:
              s = +1 if motor_angle < off - kick,  -1 if motor_angle > off + kick,  else sign(motor_velocity);   action = clip(bias + amp * s)
          move_in_direction adds the reference's steering (:64-95, 120-136): |angle_to_target| > thr -> face policy gain * clip(3 angle / pi),
          else the kick policy with the "optimal orientation" offset (+-pi/2)
  pulse   open loop: action = clip(bias + amp * square(t; half-period T, duty))

    python tools/policy_search.py [--pop 256] [--reps 64] [--gens 20] [--quick]  > profiles/r04_policy_search.txt

Prints, per task and family: best episode return (mean over the envs of the best candidate, and the candidate's parameters), sustained
speed along the target direction (move_in_direction: mean of the velocity-in-target-frame observation over steps 200-1000; move_from_origin:
distance from the origin after 10 s / 10), share of upright robots; then the same search (shorter, warm-started) with the floor friction
mu in {0.5, 1, 2} and the contact time constant solref[0] in {0.01, 0.02, 0.04} (jb_set_model_params).  No parity claim either way.
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd import model
from jitterbug_amd.vec_env import JitterbugVecEnv

ap = argparse.ArgumentParser()
ap.add_argument("--pop", type=int, default=256)
ap.add_argument("--reps", type=int, default=64)
ap.add_argument("--gens", type=int, default=20)
ap.add_argument("--gens-sweep", type=int, default=8)
ap.add_argument("--quick", action="store_true", help="a smoke run: pop 32, reps 16, 3 generations, no parameter sweep")
ap.add_argument("--no-sweep", action="store_true")
args = ap.parse_args()
if args.quick:
    args.pop, args.reps, args.gens, args.no_sweep = 32, 16, 3, True
dev = torch.device("cuda", 0)
PI = math.pi

# parameter boxes (lo, hi) per family
KICK = [("amp", 0.0, 1.0), ("kick", 0.05, PI), ("off", -PI, PI), ("bias", -1.0, 1.0), ("thr", 0.05, 1.5), ("gain", 0.0, 1.0)]
PULSE = [("amp", 0.0, 1.0), ("half_period", 1.0, 60.0), ("duty", 0.1, 0.9), ("bias", -1.0, 1.0)]


def wrap(a):
    return torch.remainder(a + PI, 2 * PI) - PI


def make_policy(family, task, theta):
    """theta [N, d] (already in physical units) -> f(t, obs [N, D]) -> action [N]"""
    if family == "pulse":
        amp, T, duty, bias = theta[:, 0], theta[:, 1], theta[:, 2], theta[:, 3]

        def f(t, obs):
            ph = torch.remainder(torch.full_like(T, float(t)), 2 * T) / (2 * T)
            return torch.clamp(bias + amp * torch.where(ph < duty, 1.0, -1.0), -1, 1)
        return f
    amp, kick, off, bias, thr, gain = (theta[:, i] for i in range(6))

    def f(t, obs):
        ma = obs[:, 13] * PI                       # de-normalised motor angle (rad); the velocity only enters by its sign
        mv = obs[:, 14]
        o = off
        steer = None
        if task == "move_in_direction":
            ang = obs[:, 15] * PI
            # reference :120-136: walk sideways when the target lies to the side
            side_p = (ang > PI / 4) & (ang <= PI)
            side_n = (ang >= -PI) & (ang < -PI / 4)
            o = off + torch.where(side_p, PI / 2, 0.0) + torch.where(side_n, -PI / 2, 0.0)
            ang2 = torch.where(side_p, (ang.abs() - PI / 2).abs(), torch.where(side_n, -(ang.abs() - PI / 2).abs(), ang))
            steer = (ang2.abs() > thr, 0.9 * gain * torch.clamp(3 * ang2 / PI, -1, 1))
        d = wrap(ma - o)
        s = torch.where(d < -kick, 1.0, torch.where(d > kick, -1.0, torch.where(mv > 0, 1.0, -1.0)))
        a = torch.clamp(bias + amp * s, -1, 1)
        if steer is not None:
            a = torch.where(steer[0], steer[1], a)
        return a
    return f


def evaluate(env, family, task, theta, steps=1000):
    """One episode of every env under its candidate's policy.  Returns per-env (return, speed, upright share)."""
    n = env.num_envs
    D = env.obs_dim
    obs = torch.zeros((n, D), device=dev); rew = torch.zeros((n,), device=dev); done = torch.zeros((n,), device=dev, dtype=torch.uint8)
    act = torch.zeros((n,), device=dev)
    env.reset_device(None, obs.data_ptr())
    pol = make_policy(family, task, theta)
    ret = torch.zeros((n,), device=dev); vel = torch.zeros((n,), device=dev); up = torch.zeros((n,), device=dev)
    for t in range(steps - 1):                      # (the 1000th step would auto-reset: stop one short, like evaluate_policy's episode)
        act.copy_(pol(t, obs))
        env.step_device(act.data_ptr(), obs.data_ptr(), rew.data_ptr(), done.data_ptr())
        ret += rew
        if t >= 200:
            up += ((1 - 2 * (obs[:, 4] ** 2 + obs[:, 5] ** 2)) > 0.9).float()
            if task == "move_in_direction":
                vel += obs[:, 16]
    if task == "move_in_direction":
        speed = vel / (steps - 1 - 200)
    else:
        speed = torch.sqrt((obs[:, 0] * 2) ** 2 + (obs[:, 1] * 2) ** 2) / (steps * 0.01)
    return ret, speed, up / (steps - 1 - 200)


def cem(env, family, task, pop, reps, gens, init=None, log=None):
    box = PULSE if family == "pulse" else KICK
    d = len(box)
    lo = torch.tensor([b[1] for b in box], device=dev); hi = torch.tensor([b[2] for b in box], device=dev)
    mean = torch.full((d,), 0.5, device=dev) if init is None else init.clone()
    std = torch.full((d,), 0.3 if init is None else 0.12, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(0)
    best = None
    for gen in range(gens):
        u = torch.clamp(mean + std * torch.randn((pop, d), generator=g, device=dev), 0, 1)
        u[0] = mean                                  # the current mean is always a candidate
        theta = (lo + u * (hi - lo)).repeat_interleave(reps, 0)
        ret, speed, up = evaluate(env, family, task, theta)
        R = ret.view(pop, reps).mean(1); S = speed.view(pop, reps).mean(1); U = up.view(pop, reps).mean(1)
        order = torch.argsort(-R)
        elite = u[order[: max(4, pop // 8)]]
        mean, std = elite.mean(0), torch.clamp(elite.std(0), min=0.02)
        i = int(order[0])
        cand = dict(ret=float(R[i]), speed=float(S[i]), upright=float(U[i]), ret_p10=float(ret.view(pop, reps)[i].quantile(0.1)), ret_p90=float(ret.view(pop, reps)[i].quantile(0.9)),
                    solved=float((ret.view(pop, reps)[i] >= 900).float().mean()), u=u[i].clone(), theta={b[0]: float(lo[k] + u[i, k] * (hi[k] - lo[k])) for k, b in enumerate(box)},
                    best_speed_any=float(S.max()))
        if best is None or cand["ret"] > best["ret"]:
            best = cand
        if log is not None:
            log("    gen %2d: best return %6.1f (speed %.3f m/s, upright %.2f), population mean %6.1f, fastest candidate %.3f m/s" % (gen, cand["ret"], cand["speed"], cand["upright"], float(R.mean()), float(S.max())))
    return best


def fmt(b):
    return "return %6.1f (p10 %5.0f p90 %5.0f, solved>=900 %.2f)  speed %.3f m/s  upright %.2f  fastest candidate seen %.3f m/s | %s" % (
        b["ret"], b["ret_p10"], b["ret_p90"], b["solved"], b["speed"], b["upright"], b["best_speed_any"], " ".join("%s=%.3f" % kv for kv in b["theta"].items()))


t00 = time.time()
print("# tools/policy_search.py on MI355X: CEM, population %d x %d envs x 999 steps x %d generations per (task, family); nominal model first" % (args.pop, args.reps, args.gens))
print("# reference figures to compare with (real MuJoCo): TARGET_SPEED 0.1 m/s (jitterbug.py:58); trained agents / heuristic policies 'solve' the tasks (return >~ 900, root.tex:279)")
P0 = model.default_params()
n = args.pop * args.reps
results = {}
for task in ("move_in_direction", "move_from_origin"):
    env = JitterbugVecEnv(n, task, seed=0, variant="auto", auto_reset=False, stream=torch.cuda.current_stream(dev).cuda_stream)
    for family in ("kick", "pulse"):
        print("%s / %s  (kernel variant %s)" % (task, family, env.kernel_variant))
        b = cem(env, family, task, args.pop, args.reps, args.gens, log=print)
        results[(task, family)] = b
        print("  BEST  " + fmt(b))
        sys.stdout.flush()
    env.close()
if not args.no_sweep:
    print("# the same search (%d generations, warm-started from the nominal model's best kick policy) with other floor friction / contact time constants" % args.gens_sweep)
    pop, reps = max(32, args.pop // 2), max(16, args.reps // 2)
    for task in ("move_in_direction", "move_from_origin"):
        env = JitterbugVecEnv(pop * reps, task, seed=0, variant="auto", auto_reset=False, stream=torch.cuda.current_stream(dev).cuda_stream)
        for mu in (0.5, 1.0, 2.0):
            for tc in (0.01, 0.02, 0.04):
                P = P0.copy()
                P[model.P_FRICTION] = mu
                P[model.P_SOLREF] = tc
                env.set_model_params(P)
                b = cem(env, "kick", task, pop, reps, args.gens_sweep, init=results[(task, "kick")]["u"])
                print("%-18s mu %.1f solref %.2f : %s" % (task, mu, tc, fmt(b)))
                sys.stdout.flush()
        env.close()
print("# %.0f s" % (time.time() - t00))
