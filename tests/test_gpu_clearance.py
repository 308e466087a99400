"""GPU property test for the geom-geom contacts the simulator leaves out: over full-length rollouts of the HIP path
at BASELINE size, in the regimes where legs are loaded hardest, the minimum distance over every geom pair MuJoCo's filters would
test (oracle/jb_clearance.c, 160 pairs, exact GJK) must stay positive - then MuJoCo would not have generated a geom-geom contact
either, and floor-only collision is not an approximation on these trajectories.  Reference: jitterbug.xml:44-107 (every jitterbug
geom has the default contype/conaffinity), :115-116 (the target geoms are the only opt-outs)."""
import numpy as np
import pytest

from jitterbug_amd import model

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("regime", ["uniform", "flat_out_plus", "flat_out_minus", "augmented"])
def test_no_geom_pair_ever_touches(regime):
    from jitterbug_amd.vec_env import JitterbugVecEnv
    from oracle import oracle as O
    n, steps, every = 4096, 1000, 10
    P = model.default_params()
    env = JitterbugVecEnv(n, "move_to_pose", seed=8, time_limit=float("inf"), auto_reset=False)     # one 1000-step rollout, no reset at the end
    if regime == "augmented":
        # config 5: one perturbed model per env, generated on the device (leg ends move by sigma = 3 mm, the motor axis by sigma =
        # 1.5 / 2 / 1 mm), EVERY draw kept.  The reference's distribution produces robots whose eccentric mass cannot turn without
        # hitting a front leg (3.6 % of the draws; ~14 % come within 1 mm): that pair - mass ellipsoid against the upper-leg
        # cylinders - is simulated since round 3 (PAIR kernel), the motor-axis thread against the upper legs since round 5; both are left out of
        # the minimum here; every OTHER pair must stay clear.
        out = env.randomise_models(seed=11, min_mass_clearance=0.0)
        P = out["params"]
        touching = O.mass_sweep_clearance(P[:1024], 72) <= 1e-9
        print("augmented: %.1f %% of the drawn robots cannot turn their mass without touching a leg" % (100 * touching.mean()))
        assert (out["attempts"] == 1).all() and 0.01 < touching.mean() < 0.08
    env.reset()
    rng = np.random.default_rng(5)
    worst, worst_pair, max_hinge = np.inf, None, 0.0
    tipped = 0.0
    touched = np.zeros(n, bool)
    for t in range(steps):
        if regime in ("uniform", "augmented"):
            a = rng.uniform(-1, 1, size=n).astype(np.float32)
        else:
            a = np.full(n, 1.0 if regime == "flat_out_plus" else -1.0, dtype=np.float32)       # about half the robots tip over
        env.step(a)
        if t % every == every - 1 or t > steps - 20:
            q, _, _ = env.get_state()
            d, pairs = O.pair_clearance(P, q, skip_simulated=(regime == "augmented"))
            touched |= d <= 0.0
            i = int(d.argmin())
            if d[i] < worst:
                worst, worst_pair = float(d[i]), (int(pairs[i, 0]), int(pairs[i, 1]), t, i)
            max_hinge = max(max_hinge, float(np.abs(q[:, 7:15]).max()))
            tipped = float(((1 - 2 * (q[:, 4] ** 2 + q[:, 5] ** 2)) < 0.5).mean())
    env.close()
    n_touch = int(touched.sum())
    if regime == "augmented":
        print("augmented: envs in which a pair the simulator does NOT collide touched at some sample: %d of %d" % (n_touch, n))
    print("%s: min pair clearance %.3f mm (geoms %s), max |leg hinge| %.3f rad, tipped over at the end %.1f %%"
          % (regime, worst * 1e3, worst_pair, max_hinge, 100 * tipped))
    if regime == "augmented":
        # the pairs the simulator does not collide (round 5: the motor-axis thread against the upper legs is simulated too and left out of the
        # minimum with the mass pair): no unsimulated pair may touch in more than 0.02 % of the envs - 0 of 4096 here
        assert n_touch <= 0.0002 * n, (n_touch, worst_pair)
    else:
        assert worst > 0.0, (worst, worst_pair)
    if regime.startswith("flat_out"):
        assert tipped > 0.1                              # the regime really has robots lying on their legs


def test_pair_witness_fires_where_the_clearance_oracle_says():
    """a run-time witness for the 152 geom pairs the simulator does not collide (jb_pair_witness / JB_FLAG_PAIR_WITNESS,
    reference jitterbug.xml:44-107: every geom collides).  With sigmas TWICE the reference's some robots do push unsimulated pairs into each
    other: the witness must fire for exactly the envs oracle/jb_clearance.c's exact GJK says, warn the caller who widened the sigmas without
    it, and stay silent on the reference's own distribution."""
    import warnings
    from jitterbug_amd.vec_env import JitterbugVecEnv
    from oracle import oracle as O
    n = 2048
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        plain = JitterbugVecEnv(64, "move_to_pose", seed=1)
        plain.randomise_models(seed=3, sd_legs=(0.006, 0.006, 0.004))
        assert any("pair_witness" in str(x.message) for x in w), "widened sigmas without the witness must warn"
        plain.close()
    with warnings.catch_warnings():
        warnings.simplefilter("error")                      # with the witness on, no warning
        env = JitterbugVecEnv(n, "move_to_pose", seed=8, time_limit=float("inf"), auto_reset=False, pair_witness=True)
        out = env.randomise_models(seed=11, sd_legs=(0.006, 0.006, 0.004), sd_mass_pos=(0.003, 0.004, 0.002))
    P = out["params"]
    env.reset()
    rng = np.random.default_rng(5)
    fired_any = np.zeros(n, bool)
    agree = checked = 0
    for t in range(120):
        env.step(rng.uniform(-1, 1, size=n).astype(np.float32))
        if t % 20 == 19:
            d_w, pairs_w = env.pair_witness()
            q, _, _ = env.get_state()
            d_o, pairs_o = O.pair_clearance(P, q, skip_simulated=True)
            clear_cut = np.abs(d_o) > 1e-7                   # (a pair within 0.1 um of touching may fall either side in the float tables)
            assert np.array_equal((d_w <= 0)[clear_cut | (d_o <= 0)], (d_o <= 0)[clear_cut | (d_o <= 0)])
            np.testing.assert_allclose(d_w[d_o > 0], d_o[d_o > 0], atol=5e-8)
            far = d_o > 1e-6
            assert np.array_equal(pairs_w[far], pairs_o[far])
            fired_any |= d_w <= 0
            checked += n
    counts, dmin = env.pair_witness_counters()
    print("2 x sigmas: %d of %d envs had an unsimulated pair interpenetrating at a sampled state; counters fired in %d envs (every launch is watched), smallest clearance %.3f mm"
          % (fired_any.sum(), n, (counts > 0).sum(), 1e3 * dmin.min()))
    assert fired_any.sum() >= 5                              # the counter fires ...
    assert np.all(counts[fired_any] > 0) and np.all(dmin[fired_any] == 0)      # ... and the per-launch passes saw every env the samples saw
    env.close()
    # the reference's own distribution: nothing to report (4096 robots x 100 steps)
    ref = JitterbugVecEnv(4096, "move_to_pose", seed=8, pair_witness=True)
    ref.randomise_models(seed=11, return_params=False)
    ref.reset()
    for t in range(100):
        ref.step(rng.uniform(-1, 1, size=4096).astype(np.float32))
    counts, dmin = ref.pair_witness_counters()
    print("reference sigmas: %d envs with an overlap, smallest clearance of an unsimulated pair %.3f mm" % ((counts > 0).sum(), 1e3 * dmin.min()))
    assert (counts > 0).sum() == 0 and dmin.min() > 0
    ref.close()
