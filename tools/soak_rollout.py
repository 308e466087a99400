"""Soak: many consecutive fused episodes (jb_step_many_device, K = 1000) on BASELINE configs[2] and on a randomised batch; every env must stay
finite and physical, the solver cap must stay (nearly) untouched.   python tools/soak_rollout.py [episodes]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jitterbug_amd.vec_env import JitterbugVecEnv
E = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda", 0)
for label, n, kw, rnd, const in (("configs[2] uniform", 4096, {}, False, False), ("configs[2] motor flat out", 4096, {}, False, True), ("8192 randomised models", 8192, dict(per_env_model=True, variant="auto"), True, False)):
    env = JitterbugVecEnv(n, "move_from_origin", seed=1, **kw)
    if rnd:
        env.randomise_models(seed=3, return_params=False)
    g = torch.Generator(device=dev); g.manual_seed(5)
    tape = torch.rand((1000, n), generator=g, device=dev) * 2 - 1
    if const:
        tape.fill_(1.0)
    rew = torch.zeros((1000, n), device=dev); obs = torch.zeros((n, env.obs_dim), device=dev)
    env.reset_device()
    t0 = time.time(); ms = []
    for e in range(E):
        te = time.time()
        env.step_many_device(1000, tape.data_ptr(), rewards_ptr=rew.data_ptr(), obs_last_ptr=obs.data_ptr())
        env.synchronize()
        ms.append((time.time() - te))
        assert bool(torch.isfinite(rew).all()) and bool(torch.isfinite(obs).all()), (label, e)
    sc, ep, cap = env.counters()
    q, v, _ = env.get_state()
    try:
        resolved = env.solver_stats()
    except AttributeError:          # (an A/B library of an older ABI)
        resolved = -1
    print("%-28s %d episodes x %d envs: %.2f M env-steps/s (per episode ms/step min %.3f max %.3f), episodes %d, unconverged substeps %.0f (%.2e of the substeps), wave-substeps solved a second time with the line search %d (%.2e of the wave-substeps), non-finite steps %d, |quat| error %.1e, z in [%.4f, %.4f]"
          % (label, E, n, n * 1000 * E / (time.time() - t0) / 1e6, min(ms), max(ms), int(ep[0]), float(cap[cap < 1000].sum()), float(cap[cap < 1000].sum()) / (n * 1000.0 * E * 50), resolved, resolved / (n / 4 * 1000.0 * E * 50), int((cap >= 1000).sum()),
             float(np.abs(np.linalg.norm(q[:, 3:7], axis=1) - 1).max()), q[:, 2].min(), q[:, 2].max()))
    assert (cap < 1000).all()
    env.close()
