// jb_task.hpp — per-environment task layer: episode reset, observation packing, reward.
// Scalar templates (T = float on the device).  Restates the Python of the reference task:
//   reset       reference jitterbug.py:601-666   (RNG: Philox4x32-10 counter streams, see below)
//   accessors   reference jitterbug.py:180-317
//   observation reference jitterbug.py:673-763, normalisation tables :324-372
//   reward      reference jitterbug.py:840-925 + dm_control rewards.tolerance (gaussian / cosine / linear)
#pragma once
#include "jb_lane.hpp"

// The task layer is inlined into several kernels (reset, observe, step, policy).  Its arithmetic is compiled WITHOUT multiply-add
// contraction, so that an observation, a reward or a policy decision has the same bits whichever kernel computed it (the compiler
// fuses a*b+c differently from one inlining context to the next): the in-kernel policy of a fused rollout must see exactly the row
// a separate jb_policy_device launch would read.
#if defined(__clang__)
#define JB_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define JB_NO_CONTRACT
#endif

namespace jb {

enum Task : int { TASK_MOVE_FROM_ORIGIN = 0, TASK_FACE_DIRECTION = 1, TASK_MOVE_IN_DIRECTION = 2, TASK_MOVE_TO_POSITION = 3, TASK_MOVE_TO_POSE = 4 };
JB_HD int obs_dim(int task) { return task == 0 ? 15 : task == 1 ? 16 : task == 2 ? 19 : task == 3 ? 18 : 19; }

// ---- Philox4x32-10 (Salmon et al. 2011); counter = (env_lo, env_hi, episode, stream), key = seed
JB_HD void philox4x32(uint64_t seed, uint64_t env, uint32_t episode, uint32_t stream, uint32_t (&out)[4]) {
    uint32_t c0 = (uint32_t)env, c1 = (uint32_t)(env >> 32), c2 = episode, c3 = stream;
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
template <typename T> JB_HD T u01(uint32_t x) { return T(x >> 8) * T(1.0 / 16777216.0); }   // [0,1) on a 24-bit grid: exact in fp32

template <typename T> struct EnvCore {        // the part of the state the task layer reads / writes
    T px, py, pz, qw, qx, qy, qz;             // root pose
    T vx, vy, vz, wx, wy, wz;                 // root velocity (world linear, body angular)
    T phi, phid;                              // motor angle (wrapped or not) and rate
    T tx, ty, tpsi;                           // target x, y, yaw
    T cx, cy, cz;                             // the root body's OWN centre of mass in root coordinates (model constant): the
                                              // point the reference's framelinvel sensor measures at, see vel_in_target
};

// reference jitterbug.py:601-666.  Draw order: angle, radius, yaw, then (random_pose) rotation angle, axis x, axis y.
// Returns the root quaternion and the target; everything else resets to qpos0 / zero (the model constants cx, cy, cz are not touched).
template <typename T>
JB_HD void episode_reset(int task, int random_pose, uint64_t seed, uint64_t env, uint32_t episode, T root_z0, EnvCore<T>& e) {
    JB_NO_CONTRACT
    uint32_t r0[4], r1[4];
    philox4x32(seed, env, episode, 0u, r0);
    philox4x32(seed, env, episode, 1u, r1);
    const T TWO_PI = T(6.283185307179586);
    T angle = u01<T>(r0[0]) * TWO_PI;                     // :609
    T radius = T(0.05) + u01<T>(r0[1]) * T(0.15);         // :610
    T yaw = u01<T>(r0[2]) * TWO_PI;                       // :611
    e.px = T(0); e.py = T(0); e.pz = root_z0;
    e.qw = T(1); e.qx = e.qy = e.qz = T(0);
    e.vx = e.vy = e.vz = e.wx = e.wy = e.wz = T(0);
    e.phi = T(0); e.phid = T(0);
    e.tx = e.ty = e.tpsi = T(0);
    if (task == TASK_FACE_DIRECTION || task == TASK_MOVE_IN_DIRECTION) e.tpsi = yaw;                                  // :618-630
    else if (task == TASK_MOVE_TO_POSITION) { e.tx = radius * vcos_b(angle); e.ty = radius * vsin_b(angle); }             // :632-639
    else if (task == TASK_MOVE_TO_POSE) { e.tx = radius * vcos_b(angle); e.ty = radius * vsin_b(angle); e.tpsi = yaw; }   // :641-648
    if (random_pose) {                                                                                                // :653-664
        T th = u01<T>(r0[3]) * TWO_PI;
        T ax = u01<T>(r1[0]) * T(0.05) - T(0.025), ay = u01<T>(r1[1]) * T(0.05) - T(0.025);
        T inv = T(1) / vsqrt(ax * ax + ay * ay + T(1));
        T sh = vsin_b(T(0.5) * th);
        e.qw = vcos_b(T(0.5) * th); e.qx = sh * ax * inv; e.qy = sh * ay * inv; e.qz = sh * inv;
    }
}

template <typename T> JB_HD T wrap_pi(T a) {             // (-pi, pi]   reference jitterbug.py:235-238, 313-316
    JB_NO_CONTRACT
    const T PI = T(3.141592653589793), TWO_PI = T(6.283185307179586);
    T k = vfloor((PI - a) / TWO_PI);                     // a + 2 pi k in (-pi, pi]
    a = a + k * TWO_PI;
    if (a > PI) a -= TWO_PI;                             // guard the rounding of the floor argument
    if (a <= -PI) a += TWO_PI;
    return a;
}
JB_HD float vatan2(float y, float x) { return atan2f(y, x); }
JB_HD double vatan2(double y, double x) { return atan2(y, x); }
JB_HD float vexp(float x) { return expf(x); }
JB_HD double vexp(double x) { return exp(x); }

// relative yaw from the Jitterbug heading to the target  (reference :192-208, 262-273, 305-317)
template <typename T> JB_HD T angle_to_target(const EnvCore<T>& e) {
    JB_NO_CONTRACT
    T R00 = e.qw * e.qw + e.qx * e.qx - e.qy * e.qy - e.qz * e.qz;
    T R10 = T(2) * (e.qx * e.qy + e.qw * e.qz);
    T yaw = vatan2(R10, R00) - T(1.5707963267948966);
    // target quat (cos psi/2, 0, 0, sin psi/2) -> its yaw by the same formula
    T c = vcos_b(T(0.5) * e.tpsi), s = vsin_b(T(0.5) * e.tpsi);
    T tyaw = vatan2(T(2) * c * s, c * c - s * s);
    return wrap_pi(tyaw - yaw);
}
// target position in the Jitterbug frame  R^T (t - p)   (reference :275-290)
template <typename T> JB_HD void target_in_body(const EnvCore<T>& e, T target_z, T (&o)[3]) {
    JB_NO_CONTRACT
    T dx = e.tx - e.px, dy = e.ty - e.py, dz = target_z - e.pz;
    T w = e.qw, x = e.qx, y = e.qy, z = e.qz;
    T R00 = w * w + x * x - y * y - z * z, R01 = T(2) * (x * y - w * z), R02 = T(2) * (x * z + w * y);
    T R10 = T(2) * (x * y + w * z), R11 = w * w - x * x + y * y - z * z, R12 = T(2) * (y * z - w * x);
    T R20 = T(2) * (x * z - w * y), R21 = T(2) * (y * z + w * x), R22 = w * w - x * x - y * y + z * z;
    o[0] = R00 * dx + R10 * dy + R20 * dz;
    o[1] = R01 * dx + R11 * dy + R21 * dz;
    o[2] = R02 * dx + R12 * dy + R22 * dz;
}
// The reference's sensor `jitterbug_framelinvel` (jitterbug.xml:121, objtype="body"): MuJoCo's mj_objectVelocity measures an
// mjOBJ_BODY object at the body's INERTIAL frame (xipos: the root body's own centre of mass), world axes - only "xbody" means
// the joint frame.  So the sensor reads  v + R (w_body x c0),  not qvel[0:3].
template <typename T> JB_HD void framelinvel(const EnvCore<T>& e, T (&o)[3]) {
    JB_NO_CONTRACT
    T lx = e.wy * e.cz - e.wz * e.cy, ly = e.wz * e.cx - e.wx * e.cz, lz = e.wx * e.cy - e.wy * e.cx;
    T w = e.qw, x = e.qx, y = e.qy, z = e.qz;
    o[0] = e.vx + (w * w + x * x - y * y - z * z) * lx + T(2) * (x * y - w * z) * ly + T(2) * (x * z + w * y) * lz;
    o[1] = e.vy + T(2) * (x * y + w * z) * lx + (w * w - x * x + y * y - z * z) * ly + T(2) * (y * z - w * x) * lz;
    o[2] = e.vz + T(2) * (x * z - w * y) * lx + T(2) * (y * z + w * x) * ly + (w * w - x * x - y * y + z * z) * lz;
}
// Jitterbug linear velocity (the sensor above) in the target frame   (reference :292-303)
template <typename T> JB_HD void vel_in_target(const EnvCore<T>& e, T (&o)[3]) {
    JB_NO_CONTRACT
    T c = vcos_b(e.tpsi), s = vsin_b(e.tpsi);
    T v[3];
    framelinvel(e, v);
    o[0] = c * v[0] + s * v[1]; o[1] = -s * v[0] + c * v[1]; o[2] = v[2];
}

// reference jitterbug.py:673-763: 15 common entries then the task's extras, in dict order; _norm :668-671
template <typename T> JB_HD void observe(int task, const EnvCore<T>& e, T target_z, T* obs, int stride) {
    JB_NO_CONTRACT
    const T PI = T(3.141592653589793);
    obs[0 * stride] = e.px * T(0.5); obs[1 * stride] = e.py * T(0.5); obs[2 * stride] = e.pz * T(20) - T(1);
    obs[3 * stride] = e.qw; obs[4 * stride] = e.qx; obs[5 * stride] = e.qy; obs[6 * stride] = e.qz;
    obs[7 * stride] = e.vx; obs[8 * stride] = e.vy; obs[9 * stride] = e.vz;
    obs[10 * stride] = e.wx * T(1.0 / 35); obs[11 * stride] = e.wy * T(1.0 / 35); obs[12 * stride] = e.wz * T(1.0 / 35);
    obs[13 * stride] = wrap_pi(e.phi + T(1.5707963267948966)) / PI;      // :222-239
    obs[14 * stride] = e.phid * T(1.0 / 180);
    // the task's extras: worked out into plain values and stored by ONE set of stores (every caller passes room for 19 entries; the ones beyond
    // the task's width are zeros nobody reads) - stores of their own in every branch end as a choice between addresses, which keeps the
    // row in private memory
    T t3[3];
    T x15 = T(0), x16 = T(0), x17 = T(0), x18 = T(0);
    if (task == TASK_FACE_DIRECTION) {
        x15 = angle_to_target(e) / PI;
    } else if (task == TASK_MOVE_IN_DIRECTION) {
        x15 = angle_to_target(e) / PI;
        vel_in_target(e, t3);
        x16 = t3[0]; x17 = t3[1]; x18 = t3[2];
    } else if (task == TASK_MOVE_TO_POSITION) {
        target_in_body(e, target_z, t3);
        x15 = t3[0] * T(1.0 / 3); x16 = t3[1] * T(1.0 / 3); x17 = t3[2] * T(10);
    } else if (task == TASK_MOVE_TO_POSE) {
        target_in_body(e, target_z, t3);
        x15 = t3[0] * T(1.0 / 3); x16 = t3[1] * T(1.0 / 3); x17 = t3[2] * T(10);
        x18 = angle_to_target(e) / PI;
    }
    obs[15 * stride] = x15; obs[16 * stride] = x16; obs[17 * stride] = x17; obs[18 * stride] = x18;
}

// reference jitterbug.py:840-925.  tolerance() sigmoids with the reference's arguments folded in:
//   position  gaussian, bounds (0,0), margin .05, value_at_margin .1   -> 0.1^((d/.05)^2)
//   upright   gaussian, bounds (1,1), margin .5                        -> 0.1^(((1-Rzz)/.5)^2), 1 when Rzz == 1
//   heading   cosine,   bounds (0,0), margin pi/2, value_at_margin 0   -> (1+cos(2 dpsi))/2 for |dpsi| < pi/2 else 0
//   velocity  linear,   bounds (.1,inf), margin .1, value_at_margin 0  -> clamp(v/.1, 0, 1)
// the four reward terms on their own (reference :840-889: heading_reward, velocity_reward, position_reward, upright_reward),
// out = [P, H, V, U]; reward() below combines the ones its task uses
template <typename T> JB_HD void reward_terms(const EnvCore<T>& e, T target_z, T (&out)[4]) {
    JB_NO_CONTRACT
    const T LN01 = T(-2.302585092994046);
    T Rzz = e.qw * e.qw - e.qx * e.qx - e.qy * e.qy + e.qz * e.qz;
    T du = vabs(T(1) - Rzz) * T(2);
    out[3] = (Rzz == T(1)) ? T(1) : vexp(LN01 * du * du);
    T t3[3];
    target_in_body(e, target_z, t3);
    T d = vsqrt(t3[0] * t3[0] + t3[1] * t3[1] + t3[2] * t3[2]), dn = d * T(20);
    out[0] = (d == T(0)) ? T(1) : vexp(LN01 * dn * dn);
    T a = angle_to_target(e), x = vabs(a) / T(1.5707963267948966);
    out[1] = (a == T(0)) ? T(1) : (x < T(1) ? T(0.5) * (T(1) + vcos_b(T(3.141592653589793) * x)) : T(0));
    vel_in_target(e, t3);
    T v = t3[0];
    out[2] = (v >= T(0.1)) ? T(1) : ((T(0.1) - v) * T(10) < T(1) ? T(1) - (T(0.1) - v) * T(10) : T(0));
}
template <typename T> JB_HD T reward(int task, const EnvCore<T>& e, T target_z) {
    JB_NO_CONTRACT
    const T LN01 = T(-2.302585092994046);     // ln 0.1
    T Rzz = e.qw * e.qw - e.qx * e.qx - e.qy * e.qy + e.qz * e.qz;
    T du = vabs(T(1) - Rzz) * T(2);
    T U = (Rzz == T(1)) ? T(1) : vexp(LN01 * du * du);
    T r = T(0);
    if (task == TASK_MOVE_FROM_ORIGIN || task == TASK_MOVE_TO_POSITION || task == TASK_MOVE_TO_POSE) {
        T t3[3];
        target_in_body(e, target_z, t3);
        T d = vsqrt(t3[0] * t3[0] + t3[1] * t3[1] + t3[2] * t3[2]);
        T dn = d * T(20);
        T P = (d == T(0)) ? T(1) : vexp(LN01 * dn * dn);
        r = (task == TASK_MOVE_FROM_ORIGIN) ? (T(1) - P) : P;
    }
    if (task == TASK_FACE_DIRECTION || task == TASK_MOVE_TO_POSE) {
        T a = angle_to_target(e);
        T x = vabs(a) / T(1.5707963267948966);
        T H = (a == T(0)) ? T(1) : (x < T(1) ? T(0.5) * (T(1) + vcos_b(T(3.141592653589793) * x)) : T(0));
        r = (task == TASK_FACE_DIRECTION) ? H : r * H;
    }
    if (task == TASK_MOVE_IN_DIRECTION) {
        T t3[3];
        vel_in_target(e, t3);
        T v = t3[0];
        r = (v >= T(0.1)) ? T(1) : ((T(0.1) - v) * T(10) < T(1) ? T(1) - (T(0.1) - v) * T(10) : T(0));
    }
    return r * U;
}


// ---- heuristic bang-bang policies on a flat (normalised) observation row; reference heuristic_policies.py:6-136.
// They read the normalised observation entries and compare them with thresholds in radians, exactly like the reference.
template <typename T> struct PolicyParams { T kick_angle, speed, angle_threshold; };     // reference keyword arguments :28, :64, :81, :98
template <typename T> JB_HD PolicyParams<T> default_policy_params() { PolicyParams<T> p; p.kick_angle = T(0.7853981633974483); p.speed = T(0.3); p.angle_threshold = T(0.3490658503988659); return p; }
template <typename T> JB_HD T policy_face(T angle) {                                  // :6-25
    JB_NO_CONTRACT
    T v = T(3) * angle / T(3.141592653589793);
    v = v > T(1) ? T(1) : (v < T(-1) ? T(-1) : v);
    return T(0.9) * v;
}
template <typename T> JB_HD T policy_forward(T motor_angle, T motor_vel, T offset, const PolicyParams<T>& pp) {  // :28-56
    JB_NO_CONTRACT
    if (motor_angle < offset - pp.kick_angle) return pp.speed;
    if (motor_angle > offset + pp.kick_angle) return -pp.speed;
    return motor_vel > T(0) ? pp.speed : -pp.speed;
}
template <typename T> JB_HD T heuristic_policy(int task, const T* obs, int stride, const PolicyParams<T>& pp) {
    JB_NO_CONTRACT
    const T PI = T(3.141592653589793), Q = T(0.7853981633974483), H = T(1.5707963267948966), THR = pp.angle_threshold;
    const T ma = obs[13 * stride], mv = obs[14 * stride];
    if (task == TASK_MOVE_FROM_ORIGIN) return policy_forward(ma, mv, T(0), pp);       // :59-61
    if (task == TASK_FACE_DIRECTION) return policy_face(obs[15 * stride]);
    if (task == TASK_MOVE_IN_DIRECTION || task == TASK_MOVE_TO_POSITION) {            // :64-95, 120-136
        T ang = (task == TASK_MOVE_IN_DIRECTION) ? obs[15 * stride] : vatan2(obs[15 * stride], -obs[16 * stride]);
        T off = T(0);
        if (ang > Q && ang <= PI) { off = H; ang = vabs(vabs(ang) - H); }
        else if (ang >= -PI && ang < -Q) { off = -H; ang = -vabs(vabs(ang) - H); }
        return vabs(ang) > THR ? policy_face(ang) : policy_forward(ma, mv, off, pp);
    }
    T dx = obs[15 * stride], dy = obs[16 * stride];                                   // move_to_pose :98-118
    T ang = vatan2(dx, -dy);
    if (vabs(ang) > THR) return policy_face(ang);
    if (vsqrt(dx * dx + dy * dy) > T(0.01)) return policy_forward(ma, mv, T(0), pp);
    return policy_face(obs[18 * stride]);
}

}  // namespace jb
