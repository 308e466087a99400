// host_harness.cpp — TEST INFRASTRUCTURE.  Compiles the simulator source the HIP kernel runs
// (jitterbug_amd/csrc/jb_sim.hpp) for the host, with the 4 lanes of a quad emulated by jb::Quad<T>,
// so tests/ can check that math in fp64 and fp32 against the oracle without a GPU.
// It is never linked into, or loaded by, the product library.
#include <cstring>
#include <limits>
#include <thread>
#include <vector>

#include "../jitterbug_amd/csrc/jb_model_build.hpp"
#include "../jitterbug_amd/csrc/jb_step.hpp"

using namespace jb;

// NGROUPS = 1: the main lanes alone.  NGROUPS = 4: the wave layout of the 4-envs-per-wave kernel for ONE env - a main group and three
// helper groups, one host thread each, sharing the scratch and exchanging through jb_lane.hpp's HostWave (group_sum, row_transpose_sum,
// the rank-one pass on rows other groups built, the broadcast loop decisions): the same code paths the device takes with helper lanes.
static int g_offload = 1;      // lane group 1 as the main lanes' replica (SimOpts::offload), as the one-wave-per-SIMD kernels run it
extern "C" void jbh_set_offload(int on) { g_offload = on; }
static int g_aux = 1;          // aux bodies on lane groups 2 / 3 (SimOpts::aux)
extern "C" void jbh_set_aux(int on) { g_aux = on; }
static int g_spread = 1;       // spread contact sweeps (SimOpts::spread)
extern "C" void jbh_set_spread(int on) { g_spread = on; }
template <typename T>
static int run(const double* P, double* qpos, double* qvel, double ctrl, int nsub, int contacts, int max_newton, int implicit_damp, double* fail, int ngroups = 1, int rank_one = 1, int lean = 0, int pair = 0) {
    using V = Quad<T>;
    LaneModel<V> m;
    T tab[LM_TABLE];
    { int rc = build_packed_model<T>(P, tab); if (rc) return rc; }
    m.c.inv = tab; m.c.tab = tab + LM_INV; m.c.lean = lean != 0; m.c.preload();
    LaneState<V> s;
    s.px = V(T(qpos[0])); s.py = V(T(qpos[1])); s.pz = V(T(qpos[2]));
    s.qw = V(T(qpos[3])); s.qx = V(T(qpos[4])); s.qy = V(T(qpos[5])); s.qz = V(T(qpos[6]));
    // low-order words of the compensated position state (what jb_set_state imports): value - (double)(T)value
    auto lo = [](double x) { return V(T(x - (double)T(x))); };
    s.pz_lo = lo(qpos[2]); s.qw_lo = lo(qpos[3]); s.qx_lo = lo(qpos[4]); s.qy_lo = lo(qpos[5]); s.qz_lo = lo(qpos[6]);
    s.vx = V(T(qvel[0])); s.vy = V(T(qvel[1])); s.vz = V(T(qvel[2]));
    s.wx = V(T(qvel[3])); s.wy = V(T(qvel[4])); s.wz = V(T(qvel[5]));
    double ph = qpos[15], k = std::floor((ph + M_PI) / (2 * M_PI));
    s.phi = V(T(ph - k * 2 * M_PI)); s.turns = V(T(k)); s.phid = V(T(qvel[14]));
    s.th1 = V(T(qpos[7]), T(qpos[9]), T(qpos[11]), T(qpos[13]));
    s.th2 = V(T(qpos[8]), T(qpos[10]), T(qpos[12]), T(qpos[14]));
    s.thd1 = V(T(qvel[6]), T(qvel[8]), T(qvel[10]), T(qvel[12]));
    s.thd2 = V(T(qvel[7]), T(qvel[9]), T(qvel[11]), T(qvel[13]));
    for (int i = 0; i < 3; i++) { s.wa[i] = V(T(0)); s.wl[i] = V(T(0)); }
    s.wj[0] = s.wj[1] = V(T(0)); s.wm = V(T(0)); s.fail = V(T(0));
    SimOpts o; o.contacts = contacts; o.max_newton = max_newton; o.implicit_damp = implicit_damp; o.rank_one = rank_one; o.lean = lean; o.offload = (g_offload && !lean && ngroups >= 2) ? 1 : 0; o.prof = nullptr; o.hist = nullptr;
    o.spread = g_spread;
    o.aux = (g_aux && o.offload && ngroups == 4 && !pair) ? 1 : 0;          // groups 2 and 3 run phase A on the motor body and the root body (SimOpts::aux), like the ordinary device kernel
    T auxtab[LM_AUX];
    build_aux_block<T>(tab, auxtab);
    LaneModel<V> m_aux = m;
    m_aux.c.tab = auxtab; m_aux.c.tab_rare = tab + LM_INV; m_aux.c.preload();
    constexpr int SCMAX = SC_COUNT > SC_COUNT_LEAN_PAIR ? SC_COUNT : SC_COUNT_LEAN_PAIR;
    V scratch[SCMAX];                  // (the LEAN variant parks state / system / factorisation where the ordinary one has its reduction buffer and overflow candidates)
    V ovcbuf[4 * (NSLOT - ROW_K) + 9]; // LEAN: the candidates beyond the row cache live outside the scratch (global memory on the device), and behind them the thread pair contact's frame
    auto set_ovc = [&](LaneScratch<V>& sc) {
        sc.ovc = lean ? ovcbuf : scratch + SC_OVC; sc.ovc_stride = 1; sc.red_lds = !lean; sc.pd = lean ? SC_PD_LEAN : SC_PD;
        sc.pd2 = lean ? 4 * (NSLOT - ROW_K) : SC_PD2 - SC_OVC;
    };
    normalise_state(s);
    if (ngroups <= 1) {
        LaneScratch<V> sc; sc.p = scratch; sc.stride = 1; sc.grp = 0; sc.ngrp = 1; sc.gstride = 4; set_ovc(sc);
        for (int k = 0; k < SCMAX; k++) scratch[k] = V(std::numeric_limits<T>::quiet_NaN());
        if (lean) state_store(sc, s);
        scratch[sc.pd + 11] = V(T(0));         // the narrow phase starts cold in every control step
        for (int i = 0; i < nsub; i++) {
            // poison the scratch: a substep must not read anything it has not written itself (on the device LDS keeps whatever
            // the previous kernel left there); in the LEAN variant the parked state is the one thing that carries over
            for (int k = 0; k < (lean ? SC_LSTATE : SCMAX); k++) if (k < sc.pd + 9 || k > sc.pd + 11) scratch[k] = V(std::numeric_limits<T>::quiet_NaN());      // (the pair's warm start carries over, like on the device)
            for (auto& c : ovcbuf) c = V(std::numeric_limits<T>::quiet_NaN());
            if (pair) substep<V, true>(m, sc, s, V(T(ctrl)), o); else substep<V>(m, sc, s, V(T(ctrl)), o);
        }
        if (lean) state_load(sc, s);
    } else {
        HostWave wave;
        wave.ngrp = ngroups; wave.gstride = ngroups == 2 ? 32 : 16;            // 16: the lane distance between groups in the 4-envs-per-wave kernel (selects its transposed reduction); 32: the 8-envs-per-wave kernel's two groups
        auto body = [&](int g) {
            g_host_wave = &wave; g_host_grp = g;
            LaneScratch<V> sc; sc.p = scratch; sc.stride = 1; sc.grp = g; sc.ngrp = ngroups; sc.gstride = ngroups == 2 ? 32 : 16; set_ovc(sc);
            sc.aux_lane = o.aux && g >= 2;
            const LaneModel<V>& mg = sc.aux_lane ? m_aux : m;
            LaneState<V> hs = s;                            // helper lanes start from a harmless state, like the kernel's (replica and aux lanes: from the env's state)
            LaneState<V>& st = (g == 0) ? s : hs;
            if (g != 0 && !(o.offload && g == 1) && !sc.aux_lane) {
                hs.px = hs.py = hs.pz = V(T(0)); hs.qw = V(T(1)); hs.qx = hs.qy = hs.qz = V(T(0)); hs.vx = hs.vy = hs.vz = hs.wx = hs.wy = hs.wz = V(T(0));
                hs.pz_lo = hs.qw_lo = hs.qx_lo = hs.qy_lo = hs.qz_lo = V(T(0));
                hs.phi = hs.phid = hs.turns = V(T(0)); hs.th1 = hs.th2 = hs.thd1 = hs.thd2 = V(T(0));
            }
            if (g == 0) {
                for (int k = 0; k < SCMAX; k++) scratch[k] = V(std::numeric_limits<T>::quiet_NaN());
                if (lean) state_store(sc, st);
                if (o.offload) for (int k = 0; k < 56; k++) scratch[SC_ZERO + k] = V(T(0));       // written once per kernel on the device
            }
            for (int i = 0; i < nsub; i++) {
                wave.barrier();
                if (g == 0) { for (int k = 0; k < (lean ? SC_LSTATE : o.offload ? SC_ZERO : SCMAX); k++) if (i == 0 || k < sc.pd + 9 || k > sc.pd + 11) scratch[k] = V(std::numeric_limits<T>::quiet_NaN()); if (i == 0) scratch[sc.pd + 11] = V(T(0)); for (auto& c : ovcbuf) c = V(std::numeric_limits<T>::quiet_NaN()); }
                wave.barrier();
                if (pair) substep<V, true>(mg, sc, st, V(T(ctrl)), o); else substep<V>(mg, sc, st, V(T(ctrl)), o);
            }
            if (g == 0 && lean) state_load(sc, st);
            g_host_wave = nullptr;
        };
        std::vector<std::thread> th;
        for (int g = 1; g < ngroups; g++) th.emplace_back(body, g);
        body(0);
        for (auto& t : th) t.join();
    }
    // replicated quantities must agree across the quad
    for (int l = 1; l < 4; l++) if (s.px.v[l] != s.px.v[0] || s.qw.v[l] != s.qw.v[0] || s.wz.v[l] != s.wz.v[0] || s.phid.v[l] != s.phid.v[0]) return -100;
    T n = std::sqrt(s.qw.v[0] * s.qw.v[0] + s.qx.v[0] * s.qx.v[0] + s.qy.v[0] * s.qy.v[0] + s.qz.v[0] * s.qz.v[0]);
    qpos[0] = s.px.v[0]; qpos[1] = s.py.v[0]; qpos[2] = (double)s.pz.v[0] + (double)s.pz_lo.v[0];
    qpos[3] = ((double)s.qw.v[0] + (double)s.qw_lo.v[0]) / n; qpos[4] = ((double)s.qx.v[0] + (double)s.qx_lo.v[0]) / n;
    qpos[5] = ((double)s.qy.v[0] + (double)s.qy_lo.v[0]) / n; qpos[6] = ((double)s.qz.v[0] + (double)s.qz_lo.v[0]) / n;
    qvel[0] = s.vx.v[0]; qvel[1] = s.vy.v[0]; qvel[2] = s.vz.v[0]; qvel[3] = s.wx.v[0]; qvel[4] = s.wy.v[0]; qvel[5] = s.wz.v[0];
    for (int l = 0; l < 4; l++) { qpos[7 + 2 * l] = s.th1.v[l]; qpos[8 + 2 * l] = s.th2.v[l]; qvel[6 + 2 * l] = s.thd1.v[l]; qvel[7 + 2 * l] = s.thd2.v[l]; }
    qpos[15] = (double)s.phi.v[0] + 2 * M_PI * (double)s.turns.v[0]; qvel[14] = s.phid.v[0];
    if (fail) *fail = s.fail.v[0];
    return 0;
}

extern "C" int jbh_step(const double* P, double* qpos, double* qvel, double ctrl, int nsub, int contacts, int max_newton, int implicit_damp,
                        int use_float, double* fail) {
    return use_float ? run<float>(P, qpos, qvel, ctrl, nsub, contacts, max_newton, implicit_damp, fail)
                     : run<double>(P, qpos, qvel, ctrl, nsub, contacts, max_newton, implicit_damp, fail);
}
// the same with ngroups lane groups (1 or 4) and the rank-one Newton passes on or off
extern "C" int jbh_step_groups(const double* P, double* qpos, double* qvel, double ctrl, int nsub, int contacts, int max_newton, int use_float, int ngroups, int rank_one, double* fail) {
    if (ngroups != 1 && ngroups != 2 && ngroups != 4) return -101;
    return use_float ? run<float>(P, qpos, qvel, ctrl, nsub, contacts, max_newton, 1, fail, ngroups, rank_one)
                     : run<double>(P, qpos, qvel, ctrl, nsub, contacts, max_newton, 1, fail, ngroups, rank_one);
}
// ... and in the LEAN variant (state / system / factorisation parked in the scratch, constants never preloaded)
extern "C" int jbh_step_lean(const double* P, double* qpos, double* qvel, double ctrl, int nsub, int contacts, int max_newton, int use_float, int ngroups, int rank_one, double* fail) {
    if (ngroups != 1 && ngroups != 2 && ngroups != 4) return -101;
    return use_float ? run<float>(P, qpos, qvel, ctrl, nsub, contacts, max_newton, 1, fail, ngroups, rank_one, 1)
                     : run<double>(P, qpos, qvel, ctrl, nsub, contacts, max_newton, 1, fail, ngroups, rank_one, 1);
}
// ... and with the geom-geom pair contact (mass ellipsoid against the upper-leg cylinders): the PAIR instantiation of the substep
extern "C" int jbh_step_pair(const double* P, double* qpos, double* qvel, double ctrl, int nsub, int contacts, int max_newton, int use_float, int ngroups, int rank_one, double* fail) {
    if (ngroups != 1 && ngroups != 2 && ngroups != 4) return -101;
    return use_float ? run<float>(P, qpos, qvel, ctrl, nsub, contacts, max_newton, 1, fail, ngroups, rank_one, 0, 1)
                     : run<double>(P, qpos, qvel, ctrl, nsub, contacts, max_newton, 1, fail, ngroups, rank_one, 0, 1);
}
// ... and PAIR in the LEAN layout (parked state / system / factorisation incl. the cross term's share, pair frame behind them)
extern "C" int jbh_step_pair_lean(const double* P, double* qpos, double* qvel, double ctrl, int nsub, int contacts, int max_newton, int use_float, int ngroups, int rank_one, double* fail) {
    if (ngroups != 1 && ngroups != 2 && ngroups != 4) return -101;
    return use_float ? run<float>(P, qpos, qvel, ctrl, nsub, contacts, max_newton, 1, fail, ngroups, rank_one, 1, 1)
                     : run<double>(P, qpos, qvel, ctrl, nsub, contacts, max_newton, 1, fail, ngroups, rank_one, 1, 1);
}
// ---- K control steps in ONE call, the loop of the fused rollout kernel (jb_api.hip step_body): the state, the step counter, the episode
// number and the target stay in the lane variables between the control steps; every step = normalise + nsub substeps +
// control_step_tail (jb_step.hpp: failure flag, reward, time limit, in-place reset, observation) and, with actions == NULL, the heuristic
// policy evaluated on the observation just produced.  ngroups = 1 (main lanes) or 4 (main + replica + two helper groups, one thread each).
// io: qpos/qvel/target in and out; rows_out [K, D+2] = [obs | reward | done]; counters = [step_count, episode] in and out.
template <typename T>
static int rollout(const double* P, double* qpos, double* qvel, double* target, int* counters, int K, const double* actions, int task, int nsub, int step_limit, int auto_reset,
                   int random_pose, unsigned long long seed, unsigned long long env_global, const double* policy_params, int ngroups, double* rows_out) {
    using V = Quad<T>;
    LaneModel<V> m;
    T tab[LM_TABLE];
    { int rc = build_packed_model<T>(P, tab); if (rc) return rc; }
    m.c.inv = tab; m.c.tab = tab + LM_INV; m.c.lean = false; m.c.preload();
    LaneState<V> s0;
    s0.px = V(T(qpos[0])); s0.py = V(T(qpos[1])); s0.pz = V(T(qpos[2]));
    s0.qw = V(T(qpos[3])); s0.qx = V(T(qpos[4])); s0.qy = V(T(qpos[5])); s0.qz = V(T(qpos[6]));
    auto lo = [](double x) { return V(T(x - (double)T(x))); };
    s0.pz_lo = lo(qpos[2]); s0.qw_lo = lo(qpos[3]); s0.qx_lo = lo(qpos[4]); s0.qy_lo = lo(qpos[5]); s0.qz_lo = lo(qpos[6]);
    s0.vx = V(T(qvel[0])); s0.vy = V(T(qvel[1])); s0.vz = V(T(qvel[2]));
    s0.wx = V(T(qvel[3])); s0.wy = V(T(qvel[4])); s0.wz = V(T(qvel[5]));
    double ph = qpos[15], kk = std::floor((ph + M_PI) / (2 * M_PI));
    s0.phi = V(T(ph - kk * 2 * M_PI)); s0.turns = V(T(kk)); s0.phid = V(T(qvel[14]));
    s0.th1 = V(T(qpos[7]), T(qpos[9]), T(qpos[11]), T(qpos[13]));
    s0.th2 = V(T(qpos[8]), T(qpos[10]), T(qpos[12]), T(qpos[14]));
    s0.thd1 = V(T(qvel[6]), T(qvel[8]), T(qvel[10]), T(qvel[12]));
    s0.thd2 = V(T(qvel[7]), T(qvel[9]), T(qvel[11]), T(qvel[13]));
    for (int i = 0; i < 3; i++) { s0.wa[i] = V(T(0)); s0.wl[i] = V(T(0)); }
    s0.wj[0] = s0.wj[1] = V(T(0)); s0.wm = V(T(0)); s0.fail = V(T(0));
    SimOpts o; o.contacts = 1; o.max_newton = 12; o.implicit_damp = 1; o.rank_one = 1; o.lean = 0; o.offload = (g_offload && ngroups >= 2) ? 1 : 0; o.prof = nullptr; o.hist = nullptr;
    o.spread = g_spread;
    o.aux = (g_aux && o.offload && ngroups == 4) ? 1 : 0;
    T auxtab[LM_AUX];
    build_aux_block<T>(tab, auxtab);
    LaneModel<V> m_aux = m;
    m_aux.c.tab = auxtab; m_aux.c.tab_rare = tab + LM_INV; m_aux.c.preload();
    V scratch[SC_COUNT];
    const int D = obs_dim(task);
    TaskOpts topt; topt.task = task; topt.step_limit = step_limit; topt.auto_reset = auto_reset; topt.random_pose = random_pose; topt.seed = seed; topt.env_global = env_global;
    PolicyParams<T> pp = default_policy_params<T>();
    if (policy_params) { pp.kick_angle = T(policy_params[0]); pp.speed = T(policy_params[1]); pp.angle_threshold = T(policy_params[2]); }
    LaneState<V> s_final;
    EpisodeRegs<T> er_final;
    HostWave wave;
    wave.ngrp = ngroups; wave.gstride = 16;
    auto body = [&](int g) {
        if (ngroups > 1) { g_host_wave = &wave; g_host_grp = g; }
        LaneScratch<V> sc; sc.p = scratch; sc.stride = 1; sc.grp = g; sc.ngrp = ngroups; sc.gstride = ngroups > 1 ? 16 : 4;
        sc.ovc = scratch + SC_OVC; sc.ovc_stride = 1; sc.red_lds = true; sc.pd = SC_PD; sc.pd2 = SC_PD2 - SC_OVC;
        sc.aux_lane = o.aux && g >= 2;
        const LaneModel<V>& mg = sc.aux_lane ? m_aux : m;
        const bool rep = g == 0 || (o.offload && g == 1) || sc.aux_lane;          // the lanes that hold the env's state (main, replica, aux)
        LaneState<V> s = s0;
        if (!rep) {
            s.px = s.py = s.pz = V(T(0)); s.qw = V(T(1)); s.qx = s.qy = s.qz = V(T(0)); s.vx = s.vy = s.vz = s.wx = s.wy = s.wz = V(T(0));
            s.pz_lo = s.qw_lo = s.qx_lo = s.qy_lo = s.qz_lo = V(T(0));
            s.phi = s.phid = s.turns = V(T(0)); s.th1 = s.th2 = s.thd1 = s.thd2 = V(T(0));
        }
        EpisodeRegs<T> er;
        er.step_count = counters[0]; er.episode = (uint32_t)counters[1];
        er.tgt[0] = T(target[0]); er.tgt[1] = T(target[1]); er.tgt[2] = T(target[2]);
        if (g == 0) {
            for (int k = 0; k < SC_COUNT; k++) scratch[k] = V(std::numeric_limits<T>::quiet_NaN());
            if (o.offload) for (int k = 0; k < 56; k++) scratch[SC_ZERO + k] = V(T(0));
        }
        T ctrl_next = T(0);
        if (!actions && rep) {
            T obs0[19];
            EnvCore<T> e0;
            core_from_lane_state<V>(mg, s, er.tgt, e0);
            observe<T>(task, e0, lane0(mg.c[LM_TARGET_Z]), obs0, 1);
            ctrl_next = heuristic_policy<T>(task, obs0, 1, pp);
        }
        for (int k = 0; k < K; k++) {
            T ctrl = ctrl_next;
            if (actions && rep) ctrl = T(actions[k]);
            if (rep) normalise_state(s);
            for (int i = 0; i < nsub; i++) {
                if (ngroups > 1) wave.barrier();
                if (g == 0) for (int q = 0; q < (o.offload ? SC_ZERO : SC_COUNT); q++) scratch[q] = V(std::numeric_limits<T>::quiet_NaN());
                if (ngroups > 1) wave.barrier();
                substep<V>(mg, sc, s, V(ctrl), o);
            }
            if (!rep) continue;
            T obs[19], rew;
            bool done;
            control_step_tail<V>(topt, mg, s, er, obs, rew, done);
            if (!actions) ctrl_next = heuristic_policy<T>(task, obs, 1, pp);
            if (g == 0) {
                double* row = rows_out + (size_t)k * (D + 2);
                for (int j = 0; j < D; j++) row[j] = (double)obs[j];
                row[D] = (double)rew; row[D + 1] = done ? 1.0 : 0.0;
            }
        }
        if (g == 0) { s_final = s; er_final = er; }
        g_host_wave = nullptr;
    };
    if (ngroups <= 1) body(0);
    else {
        std::vector<std::thread> th;
        for (int g = 1; g < ngroups; g++) th.emplace_back(body, g);
        body(0);
        for (auto& t : th) t.join();
    }
    const LaneState<V>& s = s_final;
    T nq = std::sqrt(s.qw.v[0] * s.qw.v[0] + s.qx.v[0] * s.qx.v[0] + s.qy.v[0] * s.qy.v[0] + s.qz.v[0] * s.qz.v[0]);
    qpos[0] = s.px.v[0]; qpos[1] = s.py.v[0]; qpos[2] = (double)s.pz.v[0] + (double)s.pz_lo.v[0];
    qpos[3] = ((double)s.qw.v[0] + (double)s.qw_lo.v[0]) / nq; qpos[4] = ((double)s.qx.v[0] + (double)s.qx_lo.v[0]) / nq;
    qpos[5] = ((double)s.qy.v[0] + (double)s.qy_lo.v[0]) / nq; qpos[6] = ((double)s.qz.v[0] + (double)s.qz_lo.v[0]) / nq;
    qvel[0] = s.vx.v[0]; qvel[1] = s.vy.v[0]; qvel[2] = s.vz.v[0]; qvel[3] = s.wx.v[0]; qvel[4] = s.wy.v[0]; qvel[5] = s.wz.v[0];
    for (int l = 0; l < 4; l++) { qpos[7 + 2 * l] = s.th1.v[l]; qpos[8 + 2 * l] = s.th2.v[l]; qvel[6 + 2 * l] = s.thd1.v[l]; qvel[7 + 2 * l] = s.thd2.v[l]; }
    qpos[15] = (double)s.phi.v[0] + 2 * M_PI * (double)s.turns.v[0]; qvel[14] = s.phid.v[0];
    target[0] = er_final.tgt[0]; target[1] = er_final.tgt[1]; target[2] = er_final.tgt[2];
    counters[0] = er_final.step_count; counters[1] = (int)er_final.episode;
    return 0;
}
extern "C" int jbh_rollout(const double* P, double* qpos, double* qvel, double* target, int* counters, int K, const double* actions, int task, int nsub, int step_limit,
                           int auto_reset, int random_pose, unsigned long long seed, unsigned long long env_global, const double* policy_params, int ngroups, int use_float, double* rows_out) {
    if (ngroups != 1 && ngroups != 4) return -101;
    return use_float ? rollout<float>(P, qpos, qvel, target, counters, K, actions, task, nsub, step_limit, auto_reset, random_pose, seed, env_global, policy_params, ngroups, rows_out)
                     : rollout<double>(P, qpos, qvel, target, counters, K, actions, task, nsub, step_limit, auto_reset, random_pose, seed, env_global, policy_params, ngroups, rows_out);
}
// ONE substep from a captured fp32 record (jb_sim.hpp SimOpts::capture: the substep's entry state with its warm start, 64 floats), main lanes +
// three helper groups; returns the failure counter's increment.  trace = 1 prints the line-searched iteration.
extern "C" int jbh_substep_record(const double* P, const float* rec, int max_newton, int ngroups, int trace, double* fail_out) {
    using T = float;
    using V = Quad<T>;
    LaneModel<V> m;
    T tab[LM_TABLE];
    { int rc = build_packed_model<T>(P, tab); if (rc) return rc; }
    m.c.inv = tab; m.c.tab = tab + LM_INV; m.c.lean = false; m.c.preload();
    LaneState<V> s0;
    const T ctrl = rec[0];
    s0.px = V(rec[1]); s0.py = V(rec[2]); s0.pz = V(rec[3]); s0.qw = V(rec[4]); s0.qx = V(rec[5]); s0.qy = V(rec[6]); s0.qz = V(rec[7]);
    s0.pz_lo = V(rec[8]); s0.qw_lo = V(rec[9]); s0.qx_lo = V(rec[10]); s0.qy_lo = V(rec[11]); s0.qz_lo = V(rec[12]);
    s0.vx = V(rec[13]); s0.vy = V(rec[14]); s0.vz = V(rec[15]); s0.wx = V(rec[16]); s0.wy = V(rec[17]); s0.wz = V(rec[18]);
    s0.phi = V(rec[19]); s0.phid = V(rec[20]); s0.turns = V(rec[21]);
    for (int i = 0; i < 3; i++) { s0.wa[i] = V(rec[22 + i]); s0.wl[i] = V(rec[25 + i]); }
    s0.wm = V(rec[28]);
    const float* l = rec + 32;
    s0.th1 = V(l[0], l[6], l[12], l[18]); s0.th2 = V(l[1], l[7], l[13], l[19]); s0.thd1 = V(l[2], l[8], l[14], l[20]); s0.thd2 = V(l[3], l[9], l[15], l[21]);
    s0.wj[0] = V(l[4], l[10], l[16], l[22]); s0.wj[1] = V(l[5], l[11], l[17], l[23]);
    s0.fail = V(T(0));
    SimOpts o; o.contacts = 1; o.max_newton = max_newton; o.implicit_damp = 1; o.rank_one = 1; o.lean = 0; o.offload = (g_offload && ngroups >= 2) ? 1 : 0; o.prof = nullptr; o.hist = nullptr;
    o.spread = g_spread;
    o.aux = (g_aux && o.offload && ngroups == 4) ? 1 : 0;
    T auxtab[LM_AUX];
    build_aux_block<T>(tab, auxtab);
    LaneModel<V> m_aux = m;
    m_aux.c.tab = auxtab; m_aux.c.tab_rare = tab + LM_INV; m_aux.c.preload();
    V scratch[SC_COUNT];
    HostWave wave;
    wave.ngrp = ngroups; wave.gstride = 16;
    LaneState<V> s_final;
    g_ls_trace = trace;
    auto body = [&](int g) {
        if (ngroups > 1) { g_host_wave = &wave; g_host_grp = g; }
        LaneScratch<V> sc; sc.p = scratch; sc.stride = 1; sc.grp = g; sc.ngrp = ngroups; sc.gstride = ngroups > 1 ? 16 : 4;
        sc.ovc = scratch + SC_OVC; sc.ovc_stride = 1; sc.red_lds = true; sc.pd = SC_PD; sc.pd2 = SC_PD2 - SC_OVC;
        sc.aux_lane = o.aux && g >= 2;
        const bool rep = g == 0 || (o.offload && g == 1) || sc.aux_lane;
        LaneState<V> s = s0;
        if (!rep) {
            s.px = s.py = s.pz = V(T(0)); s.qw = V(T(1)); s.qx = s.qy = s.qz = V(T(0)); s.vx = s.vy = s.vz = s.wx = s.wy = s.wz = V(T(0));
            s.pz_lo = s.qw_lo = s.qx_lo = s.qy_lo = s.qz_lo = V(T(0));
            s.phi = s.phid = s.turns = V(T(0)); s.th1 = s.th2 = s.thd1 = s.thd2 = V(T(0));
        }
        if (g == 0) {
            for (int k = 0; k < SC_COUNT; k++) scratch[k] = V(std::numeric_limits<T>::quiet_NaN());
            if (o.offload) for (int k = 0; k < 56; k++) scratch[SC_ZERO + k] = V(T(0));
        }
        if (ngroups > 1) wave.barrier();
        substep<V>(sc.aux_lane ? m_aux : m, sc, s, V(ctrl), o);
        if (g == 0) s_final = s;
        g_host_wave = nullptr;
    };
    if (ngroups <= 1) body(0);
    else {
        std::vector<std::thread> th;
        for (int g = 1; g < ngroups; g++) th.emplace_back(body, g);
        body(0);
        for (auto& t : th) t.join();
    }
    g_ls_trace = 0;
    if (fail_out) *fail_out = s_final.fail.v[0];
    return 0;
}
// the line-searched second solve (jb_sim.hpp newton_phase<LS = true>): [substeps solved a second time, outer passes of those solves, passes whose
// line search shortened the step, second solves that ended at NEWTON_LS_CAP]
extern "C" void jbh_ls_stats(long* out, int reset) { for (int i = 0; i < 4; i++) { out[i] = g_ls_stats[i]; if (reset) g_ls_stats[i] = 0; } }
extern "C" void jbh_pair_narrow_stats(long* out, int reset) { out[0] = g_pair_narrow_stats[0]; out[1] = g_pair_narrow_stats[1]; if (reset) g_pair_narrow_stats[0] = g_pair_narrow_stats[1] = 0; }
extern "C" int jbh_lm_count(void) { return LM_COUNT; }
// the per-leg constant table (LM_COUNT doubles) for inspection by tests / tools
extern "C" int jbh_lane_table(const double* P, int leg, double* out) { return build_lane_model<double>(P, leg, out); }

// ---- jb_witness.hpp (the run-time witness for the geom pairs the simulator does not collide) on the host: one configuration, from the
// float constant table the device kernel reads -> smallest distance over the unsimulated pairs and the geoms that attain it
#include "../jitterbug_amd/csrc/jb_witness.hpp"
extern "C" int jbh_witness(const double* P, const double* qpos, int* pair, double* out) {
    static thread_local float tab[LM_TABLE];
    const int rc = build_packed_model<float>(P, tab);
    if (rc) return rc;
    double th1[4], th2[4];
    for (int l = 0; l < 4; l++) { th1[l] = qpos[7 + 2 * l]; th2[l] = qpos[8 + 2 * l]; }
    *out = witness_clearance<float>(tab, th1, th2, qpos[15], pair);
    return 0;
}

// ---- jb_device_guard.hpp against a recording stub of hipGetDevice / hipSetDevice (the library instantiates it with the real ones)
#include "../jitterbug_amd/csrc/jb_device_guard.hpp"
namespace {
struct StubDeviceApi {
    static int cur, fail_set, n_set, last_set;
    static int get(int* d) { *d = cur; return 0; }
    static int set(int d) { n_set++; last_set = d; if (fail_set) return 1; cur = d; return 0; }
};
int StubDeviceApi::cur = 0, StubDeviceApi::fail_set = 0, StubDeviceApi::n_set = 0, StubDeviceApi::last_set = -1;
}
// One "entry point" on a handle of device `target` while the caller's current device is `cur` (early_return: the entry point fails
// half way, like a JB_HIP return).  out = [rc of enter, device current INSIDE the entry point, device current AFTER it, hipSetDevice calls]
extern "C" void jbh_device_guard_probe(int cur, int target, int fail_set, int early_return, int* out) {
    StubDeviceApi::cur = cur; StubDeviceApi::fail_set = fail_set; StubDeviceApi::n_set = 0; StubDeviceApi::last_set = -1;
    out[1] = -1;
    auto entry = [&]() -> int {
        jb::DeviceGuard<StubDeviceApi> g;
        const int rc = g.enter(target);
        if (rc) return rc;
        out[1] = StubDeviceApi::cur;
        if (early_return) return -3;
        return 0;
    };
    out[0] = entry();
    out[2] = StubDeviceApi::cur; out[3] = StubDeviceApi::n_set;
}

// ---- the pair narrow phase of jb_sim.hpp (mass ellipsoid against the upper cylinder of `leg`) on a posed model, fp64 or fp32, for the
// comparison with the oracle's pair_geometric: inputs are the two geoms in WORLD coordinates, out = [dist, n(3), pos(3)]
extern "C" void jbh_pair_narrow(const double* ce, const double* Re, const double* sz, const double* cc, const double* ua, double rad, double half, int use_float, double* out) {
    auto run1 = [&](auto tag) {
        using T = decltype(tag);
        Vec3<T> c_e = v3<T>(T(ce[0]), T(ce[1]), T(ce[2])), s_z = v3<T>(T(sz[0]), T(sz[1]), T(sz[2])), c_c = v3<T>(T(cc[0]), T(cc[1]), T(cc[2])), u_a = v3<T>(T(ua[0]), T(ua[1]), T(ua[2]));
        Mat3<T> R;
        for (int i = 0; i < 9; i++) R.m[i] = T(Re[i]);
        T dist; Vec3<T> n, pos;
        pair_narrow<T>(c_e, R, s_z, c_c, u_a, T(rad), T(half), dist, n, pos);
        out[0] = dist; out[1] = n.x; out[2] = n.y; out[3] = n.z; out[4] = pos.x; out[5] = pos.y; out[6] = pos.z;
    };
    if (use_float) run1(float(0)); else run1(double(0));
}
