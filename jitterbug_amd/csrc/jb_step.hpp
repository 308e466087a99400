// jb_step.hpp — what a control step does AFTER its physics substeps, written once for device lanes and the host test harness:
// failure flag, reward, time limit, in-place episode reset, observation, and (rollouts) the heuristic policy's next action.
//
// Replaces, per environment, the tail of dm_control's control.Environment.step() as the reference drives it
// (reference jitterbug.py:84-90 -> Jitterbug.get_reward :891-925, get_observation :673-763, and on the step that reaches the time
// limit the VecEnv auto-reset of benchmarks/benchmark.py:146-185 -> initialize_episode :601-666); the policy is
// benchmarks/evaluate_policy.py:29-33's `action = policy(obs)`.
//
// A K-step launch (jb_api.hip step_body) calls this K times with the state, the step counter, the episode number and the target
// held in registers in between; tests/host_harness.cpp runs the same loop on the host (fp64) against the oracle.
#pragma once
#include "jb_sim.hpp"
#include "jb_task.hpp"

namespace jb {

// the value of a quantity that is replicated over the lanes of an env
JB_HD float lane0(float x) { return x; }
JB_HD double lane0(double x) { return x; }
#if !defined(__HIPCC__)
template <typename T> inline T lane0(const Quad<T>& x) { return x.v[0]; }
#endif

// per-env bookkeeping a K-step launch carries between control steps
template <typename R> struct EpisodeRegs {
    int step_count;
    uint32_t episode;
    R tgt[3];             // target x, y, yaw
};
struct TaskOpts {
    int task, step_limit, auto_reset, random_pose;
    uint64_t seed, env_global;        // RNG key and the env's GLOBAL index (results do not depend on how a batch is sharded)
};

template <typename V, typename R = typename lane_traits<V>::real>
JB_HD void core_from_lane_state(const LaneModel<V>& m, const LaneState<V>& s, const R (&tgt)[3], EnvCore<R>& e) {
    e.cx = lane0(m.c[LM_C0]); e.cy = lane0(m.c[LM_C0 + 1]); e.cz = lane0(m.c[LM_C0 + 2]);
    e.px = lane0(s.px); e.py = lane0(s.py); e.pz = lane0(s.pz); e.qw = lane0(s.qw); e.qx = lane0(s.qx); e.qy = lane0(s.qy); e.qz = lane0(s.qz);
    e.vx = lane0(s.vx); e.vy = lane0(s.vy); e.vz = lane0(s.vz); e.wx = lane0(s.wx); e.wy = lane0(s.wy); e.wz = lane0(s.wz);
    e.phi = lane0(s.phi); e.phid = lane0(s.phid);
    e.tx = tgt[0]; e.ty = tgt[1]; e.tpsi = tgt[2];
}
// the state an episode starts from (reference jitterbug.py:601-666: qpos0 with the drawn root orientation, everything else zero).
// The failure counter is not part of an episode's state: it stays.
template <typename V, typename R = typename lane_traits<V>::real>
JB_HD void lane_state_from_reset(const EnvCore<R>& e, LaneState<V>& s) {
    s.px = V(e.px); s.py = V(e.py); s.pz = V(e.pz); s.qw = V(e.qw); s.qx = V(e.qx); s.qy = V(e.qy); s.qz = V(e.qz);
    s.pz_lo = s.qw_lo = s.qx_lo = s.qy_lo = s.qz_lo = V(R(0));
    s.vx = s.vy = s.vz = s.wx = s.wy = s.wz = V(R(0)); s.phi = V(R(0)); s.phid = V(R(0)); s.turns = V(R(0));
    s.th1 = s.th2 = s.thd1 = s.thd2 = V(R(0));
#pragma unroll
    for (int i = 0; i < 3; i++) { s.wa[i] = V(R(0)); s.wl[i] = V(R(0)); }
    s.wj[0] = s.wj[1] = V(R(0)); s.wm = V(R(0));
}

// After the substeps of one control step: s is the new physics state.  Updates the failure counter, the step counter and - on the
// step that reaches the time limit, with auto_reset - replaces the state by the next episode's first (new target, episode + 1);
// returns the reward of the state the step ended in, the done flag and the observation row the caller hands out (the NEW
// episode's first observation after a reset: VecEnv semantics).
template <typename V, typename R = typename lane_traits<V>::real>
JB_HD void control_step_tail(const TaskOpts& t, const LaneModel<V>& m, LaneState<V>& s, EpisodeRegs<R>& er, R (&obs)[19], R& rew, bool& done) {
    {   // a control step that ends in a non-finite state is recorded in the failure counter (+1000; Newton cap hits add 1 each)
        const V chk = s.px + s.py + s.pz + s.qw + s.qx + s.qy + s.qz + s.vx + s.vy + s.vz + s.wx + s.wy + s.wz + s.phi + s.phid + s.th1 + s.th2 + s.thd1 + s.thd2;
        const auto bad = quad_sum_u(mbit(mnot(lt(vabs(chk), V(R(1e30))))));
        s.fail = sel(neq_u(bad, zero_u<V>()), s.fail + V(R(1000)), s.fail);
    }
    // (trailing mj_step1: derived quantities use the normalised quaternion - phase C leaves hi + lo normalised to ~1e-14)
    const R target_z = lane0(m.c[LM_TARGET_Z]);
    er.step_count = er.step_count + 1;
    EnvCore<R> e;
    core_from_lane_state<V>(m, s, er.tgt, e);
    rew = reward<R>(t.task, e, target_z);
    done = er.step_count >= t.step_limit;
    if (done && t.auto_reset) {
        episode_reset<R>(t.task, t.random_pose, t.seed, t.env_global, er.episode, lane0(m.c[LM_ROOT_Z0]), e);
        lane_state_from_reset<V>(e, s);
        er.step_count = 0;
        er.episode = er.episode + 1;
        er.tgt[0] = e.tx; er.tgt[1] = e.ty; er.tgt[2] = e.tpsi;
    }
    observe<R>(t.task, e, target_z, obs, 1);
}

}  // namespace jb
