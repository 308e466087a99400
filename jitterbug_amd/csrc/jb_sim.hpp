// jb_sim.hpp — the Jitterbug substep, written once for device lanes and the host test harness.
//
// Replaces, for one physics substep of one environment, what the reference runs
// inside MuJoCo 2.0 through dm_control's Physics.step() (reference call chain:
// jitterbug.py:84-90 control.Environment -> 50 x mj_step2/mj_step1 on
// jitterbug.xml).  This is a from-scratch formulation specialised to the
// Jitterbug topology (free root + four 2-hinge legs + one motor hinge):
//
//   * everything is expressed in the ROOT BODY FRAME about the root origin, so
//     leg kinematics depend only on that leg's two hinge angles;
//   * joint-space inertia by composite rigid bodies:  M = [Arr Br; Br^T C] with
//     Arr 6x6 (whole robot as one rigid body), per-branch coupling Br (6x2 per
//     leg, 6x1 motor) and per-branch blocks C (2x2 / 1x1) — the star topology
//     makes the joint block block-diagonal;
//   * bias forces by a Newton-Euler pass with the root acceleration pinned to
//     MuJoCo's qacc=0 convention (world-frame linear acceleration of the root
//     origin = 0), gravity folded in as a fictitious root acceleration;
//   * linear solves by eliminating each branch (2x2) onto the root: a 6x6 Schur
//     complement, Cholesky-factored redundantly by the four lanes of a quad;
//   * soft contacts exactly as MuJoCo poses them (pyramidal cone, solref/solimp
//     impedance, diagApprox regulariser) and solved in the PRIMAL like MuJoCo's
//     Newton solver:  H = M + sum_c B_c^T W_c B_c  has the same star sparsity as
//     M, so each Newton iteration is one more Schur solve; iterations stop when
//     the active set repeats (the minimiser of a piecewise quadratic is then
//     exact);
//   * semi-implicit Euler with implicit joint damping (M + h diag(b)).
//
// Lane mapping: 4 lanes per environment, lane l owns leg l; see jb_lane.hpp.
// Unknown ordering used internally: y = [alpha(3) (root angular accel, body
// frame) ; a(3) = R^T d/dt(v_world) ; leg joint accels (2, lane-private) ;
// motor accel (1, replicated)].
#pragma once
#include <type_traits>
#if !defined(__HIPCC__)
#include <cstdio>      // (host harness: trace of the line-searched solve)
#endif

#include "jb_lane.hpp"

#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
#define JB_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#define JB_ASM_MARK(txt) asm volatile("; " txt)
#else
#define JB_ASM_MARK(txt) ((void)0)
#define JB_NO_DEVICE_PROF 1
#define JB_SCHED_FENCE() ((void)0)
#endif

#if defined(JB_WAVE_STATS) && !defined(JB_NO_DEVICE_PROF)
#define JB_PROF_T0() unsigned long long _pt = __builtin_amdgcn_s_memtime()
#define JB_PROF_ADD(o, i) do { unsigned long long _n = __builtin_amdgcn_s_memtime(); if ((o).prof) (o).prof[i] += _n - _pt; _pt = _n; } while (0)
#else
#define JB_PROF_T0() ((void)0)
#define JB_PROF_ADD(o, i) ((void)0)
#endif

#ifndef JB_ROW_K
#define JB_ROW_K 9          // cached contact rows per substep (SC_ROWS); a build parameter for A/B measurements and for the tests of the beyond-the-cache path
#endif

namespace jb {

// ----------------------------------------------------------------------------- lane model table
// Index map of the per-lane constant table.  Filled by jb_build_lane_model()
// (host) from the compiled parameter table of include/jitterbug_model.h.
enum LM : int {
    // ---------------- used every substep
    LM_H = 0, LM_GRAV = 1 /*3*/, LM_KK = 4, LM_BB = 5, LM_IMP_D0 = 6, LM_IMP_DW = 7, LM_IMP_IW /*1/width*/ = 8, LM_IMP_MID = 9, LM_IMP_IMID /*1/mid*/ = 10,
    LM_MU = 11, LM_FR2 = 12, LM_GEAR = 13, LM_GAIN = 14, LM_BIAS = 15 /*3*/, LM_CTRL_LO = 18, LM_CTRL_HI = 19, LM_MTOT = 20,
    // root body
    LM_M0 = 21, LM_C0 = 22 /*3*/, LM_I0 = 25 /*6*/, LM_TRAN0 = 31,
    // motor ("mass") body
    LM_MM = 32, LM_AM = 33 /*3*/, LM_EM = 36 /*3*/, LM_DCM = 39 /*3: com - anchor*/, LM_IM = 42 /*6*/, LM_TRANM = 48,
    // own leg
    LM_A1 = 49 /*3*/, LM_E1 = 52 /*3*/, LM_DA2 = 55 /*3: knee anchor - a1*/, LM_E2 = 58 /*3*/, LM_DC1 = 61 /*3*/, LM_I1 = 64 /*6*/, LM_M1 = 70,
    LM_DC2 = 71 /*3: com2 - knee anchor*/, LM_I2 = 74 /*6*/, LM_M2 = 80, LM_K1 = 81, LM_B1 = 82, LM_K2 = 83, LM_B2 = 84, LM_TRAN1 = 85, LM_TRAN2 = 86,
    LM_DFOOT = 87 /*3*/, LM_FOOT_R = 90,
    LM_LC_D = 91 /*3: lower cylinder centre - knee anchor*/, LM_LC_AX = 94 /*3*/, LM_LC_XA = 97 /*3*/, LM_LC_R = 100, LM_LC_H = 101,
    LM_TARGET_Z = 102, LM_ROOT_Z0 = 103, LM_IMP_I1MID = 104 /* 1/(1-mid) */,
    // broadphase spheres (root reference coordinates) for the rarely touching geoms of this lane:
    //   LEG: sphere around upper cylinder + knee tip;  BX: two oriented boxes around the lane's root-body geoms
    //   (motor-body geoms: one axis-aligned cube centred on the motor axis, so it does not move with the motor angle)
    LM_BS_LEG_C = 105 /*3*/, LM_BS_LEG_R = 108,
    //   BODY: one sphere around all root / motor-body geoms of this lane - the cheap first test; the exact one (BX) runs only
    //   when some lane's sphere reaches the floor
    LM_BS_BODY_C = 109 /*3*/, LM_BS_BODY_R = 112,
    LM_HOT = 113,
    // ---------------- read only on the rare paths (from the table in LDS)
    LM_BX = LM_HOT /*2 x 15: centre(3), axes(3x3, unit), half sizes(3) + margin: boxes bounding the lane's root/motor-body geoms*/,
    LM_UC_D = LM_BX + 30 /*3: upper cylinder centre - a1*/, LM_UC_AX = LM_UC_D + 3 /*3*/, LM_UC_XA = LM_UC_D + 6 /*3*/, LM_UC_R = LM_UC_D + 9, LM_UC_H = LM_UC_D + 10,
    LM_DTIP = LM_UC_D + 11 /*3*/, LM_TIP_R = LM_UC_D + 14,
    // lane-assigned geoms of the root / motor body (lane 0: coreBody1 box, lane 1: coreBody2 box,
    // lane 2: screw1 cylinder + screw2 ellipsoid, lane 3: threadMass cylinder + mass ellipsoid on the motor body)
    LM_XB_EN = LM_UC_D + 15, LM_XB_C = LM_XB_EN + 1 /*3*/, LM_XB_R = LM_XB_EN + 4 /*9*/, LM_XB_S = LM_XB_EN + 13 /*3*/,
    LM_XC_EN = LM_XB_EN + 16, LM_XC_C = LM_XC_EN + 1 /*3*/, LM_XC_AX = LM_XC_EN + 4 /*3*/, LM_XC_XA = LM_XC_EN + 7 /*3*/, LM_XC_R = LM_XC_EN + 10, LM_XC_H = LM_XC_EN + 11,
    LM_XE_EN = LM_XC_EN + 12, LM_XE_C = LM_XE_EN + 1 /*3*/, LM_XE_R = LM_XE_EN + 4 /*9*/, LM_XE_S = LM_XE_EN + 13 /*3*/,
    LM_X_ONM = LM_XE_EN + 16 /*1: the lane's cylinder+ellipsoid sit on the motor body*/,
    // the eccentric-mass ellipsoid (motor body) for EVERY lane: its contact with the lane's own upper-leg cylinder (PAIR kernels: the
    // one geom-geom pair that touches on randomised models)
    LM_PE_C = LM_X_ONM + 4 /*3: centre*/, LM_PE_R = LM_PE_C + 3 /*9: rotation, columns = semi-axes directions*/, LM_PE_S = LM_PE_R + 9 /*3: semi-axes*/,
    LM_PE_IS = LM_PE_S + 3 /*3: 1 / (semi-axis + cylinder radius + 0.2 mm): the broad phase tests the leg's axis against this inflated ellipsoid; x <= 0: no pair.
                             The SIGN of the y entry is a flag of its own: negative = this lane's upper leg can come near the motor-axis thread (LM_PT below)*/,
    // the motor-axis thread (geom 20, a cylinder on the motor body) for EVERY lane: its contact with the lane's own upper-leg cylinder (PAIR kernels: the second
    // geom-geom pair randomised models can bring together; read only by lanes whose LM_PE_IS flag is set - 0.2 % of the reference's draws)
    LM_PT_C = LM_PE_IS + 3 /*3: centre*/, LM_PT_AX = LM_PT_C + 3 /*3: unit axis*/, LM_PT_R = LM_PT_AX + 3, LM_PT_H = LM_PT_R + 1,
    LM_COUNT = LM_PT_H + 1,

    // Entries [0, LM_INV) (global options, root body, motor body) are the same for the 4 lanes of an env and are stored
    // once; the rest is stored per lane.  Packed table: [LM_INV] then [LM_COUNT - LM_INV][4]  = LM_TABLE floats per env.
    LM_INV = 49,
    LM_TABLE = LM_INV + 4 * (LM_COUNT - LM_INV),
    LM_TABLE_BASE = LM_INV + 4 * (LM_PE_C - LM_INV),      // the table without the pair contact's entries (its tail): what the kernels without that contact stage into LDS
    // Split mode (LEAN kernel with one model per env; LaneConsts below): what stays resident in LDS per env - the lane-invariant prefix, the
    // pair contact's ellipsoid (one geom: the same in the four lanes) and the five per-lane entries that are read after phase A
    LM_SPLIT_NRES = 5 /* LM_B1, LM_B2, LM_TARGET_Z, LM_ROOT_Z0, LM_IMP_I1MID */,
    LM_SPLIT_RES = LM_INV + 15 + 4 * LM_SPLIT_NRES,
    // ... and the overlay: per lane, the entries phase A reads - three runs of the table: all of [LM_INV, LM_HOT), 16 entries from LM_UC_D (the
    // upper-leg cylinder, the knee tip and one more), 4 entries from LM_PE_IS - 1 (the pair's broad-phase scale) - re-staged from the env's
    // table in global memory at the start of every substep into scratch that is dead during phase A
    LM_SPLIT_OVL = (LM_HOT - LM_INV) + 16 + 4
};
JB_HD constexpr int lm_split_res_lane(int i) { return i == LM_B1 ? 0 : i == LM_B2 ? 1 : i == LM_TARGET_Z ? 2 : i == LM_ROOT_Z0 ? 3 : i == LM_IMP_I1MID ? 4 : -1; }
JB_HD constexpr int lm_split_res_entry(int r) { return r == 0 ? LM_B1 : r == 1 ? LM_B2 : r == 2 ? LM_TARGET_Z : r == 3 ? LM_ROOT_Z0 : LM_IMP_I1MID; }
// overlay slot of a per-lane entry (-1: not in the overlay)
JB_HD constexpr int lm_split_slot(int i) {
    return (i >= LM_INV && i < LM_HOT) ? i - LM_INV : (i >= LM_UC_D && i < LM_UC_D + 16) ? (LM_HOT - LM_INV) + i - LM_UC_D
         : (i >= LM_PE_IS - 1 && i < LM_PE_IS + 3) ? (LM_HOT - LM_INV) + 16 + i - (LM_PE_IS - 1) : -1;
}
constexpr int LM_SPLIT_STRIDE = 16;      // the overlay is laid out like the scratch of a four-env wave (the only shape split tables are launched in)
JB_HD constexpr int lm_offset(int i, int leg) { return i < LM_INV ? i : LM_INV + 4 * (i - LM_INV) + leg; }

// The packed constant table (LM_TABLE floats, see LM_INV) lives in LDS on the device, one copy per workgroup (shared
// model) or per env (per-env models).  Constants are read where they are used instead of being pinned in registers for
// the whole control step: `m.c[LM_X]` is one ds_read_b32 with an immediate offset.
// inv: base of the env's table; tab: device = lane part already offset by the lane's leg, host Quad = the lane part.
JB_HD float lane_from4(const float* p, float*) { return p[0]; }
JB_HD double lane_from4(const double* p, double*) { return p[0]; }
JB_HD float lane_bcast(const float* p, float*) { return p[0]; }
JB_HD double lane_bcast(const double* p, double*) { return p[0]; }
#if !defined(__HIPCC__)
template <typename T> inline Quad<T> lane_from4(const T* p, Quad<T>*) { return Quad<T>(p[0], p[1], p[2], p[3]); }
template <typename T> inline Quad<T> lane_bcast(const T* p, Quad<T>*) { return Quad<T>(p[0]); }
#endif
template <typename V> struct LaneConsts {
    const typename lane_traits<V>::real* inv;
    const typename lane_traits<V>::real* tab;
    const typename lane_traits<V>::real* tab_rare = nullptr;     // per-lane entries >= LM_HOT, addressed like `tab`: the same block for ordinary lanes; the AUX lanes (SimOpts::aux)
                                                                 // take their hot entries from an aux block and everything else from the leg they mirror
    // The entries used every substep are read from the table ONCE per kernel (preload()) and then live in registers: with
    // one wave per SIMD the kernel owns all 512 registers of its lanes, and the allocator parks these long-lived values in
    // the accumulation half, one v_accvgpr_read away - no LDS round trip, no s_waitcnt in the middle of the dynamics.
    V hot[LM_HOT];
    bool lean = false;              // LEAN kernel variant (two waves per SIMD): nothing is preloaded, every constant is read from LDS where it is used
    V tran0, tran1, tran2, tranm;   // the four per-body-level entries that are picked by a run-time level: kept out of the
                                    // array so that the pick is a select of values, never an indexed access (which would
                                    // force the whole array into scratch memory)
    JB_HD V tran_of(int level) const { return level == 2 ? tran2 : level == 1 ? tran1 : level == 0 ? tran0 : tranm; }
    // Split mode (LEAN kernel with one model per env: four 3 KB tables do not fit next to the scratch of eight waves per CU).  Resident in LDS
    // per env (inv, LM_SPLIT_RES floats): the lane-invariant prefix, the pair contact's ellipsoid, and the five per-lane entries that are read
    // after phase A.  Everything else phase A reads comes from the OVERLAY: scratch entries [SC_SYS, SC_SYS + LM_SPLIT_OVL) of the lane - dead
    // from the final pass of one substep to the end of phase A of the next - re-staged from the env's table in global memory at the start
    // of every substep (jb_api.hip restage_overlay).  The rest (all-geom path, broad-phase boxes) is read from global memory where it is used.
    const typename lane_traits<V>::real* cold = nullptr;
    const typename lane_traits<V>::real* ovl = nullptr;          // the lane's overlay column (stride LM_SPLIT_STRIDE)
    const typename lane_traits<V>::real* ovl_src = nullptr;      // restage: this lane's source in global memory (the env's table, its leg, its share of the slots)
    typename lane_traits<V>::real* ovl_dst = nullptr;            // ... and its destination in LDS
    bool split = false;
    JB_HD V table(int i) const {
        if (i < LM_INV) return lane_bcast(inv + i, (V*)nullptr);
        if (!split) return lane_from4((i < LM_HOT || !tab_rare ? tab : tab_rare) + 4 * (i - LM_INV), (V*)nullptr);
        const int r = lm_split_res_lane(i);
        if (r >= 0) return lane_bcast(tab + 4 * r, (V*)nullptr);                                  // (split: tab = the lane's resident entries)
        if (i >= LM_PE_C && i < LM_PE_C + 15) return lane_bcast(inv + LM_INV + (i - LM_PE_C), (V*)nullptr);
        const int sl = lm_split_slot(i);
        if (sl >= 0) return lane_bcast(ovl + sl * LM_SPLIT_STRIDE, (V*)nullptr);
        return lane_bcast(cold + 4 * (i - LM_INV), (V*)nullptr);
    }
    JB_HD void preload() {
        if (!lean) {
#pragma unroll
            for (int i = 0; i < LM_HOT; i++) hot[i] = table(i);
        }
        tran0 = table(LM_TRAN0); tran1 = table(LM_TRAN1); tran2 = table(LM_TRAN2); tranm = table(LM_TRANM);
    }
    JB_HD V operator[](int i) const { return (i < LM_HOT && !lean) ? hot[i] : table(i); }
};
// The AUX block (SimOpts::aux): the hot per-lane entries [LM_INV, LM_HOT) of the four lanes of a helper quad that runs phase A for the two bodies
// every leg lane would otherwise repeat - lane 0: the MOTOR body as "an upper leg whose lower leg has no mass" (hinge em at am, COM offset dcm,
// inertia Im, mass mm; its angle is the motor's), lane 1: the ROOT body's own mass likewise (no hinge: axis 0, COM c0, inertia I0, mass m0),
// lanes 2, 3: nothing.  Everything outside the body entries [LM_A1, LM_B2] is the mirrored leg's own value (the contact-row weights TRAN1 /
// TRAN2 that the lanes use as helper lanes in phase B, the task constants the control-step tail reads).  tab: a packed table (LM_TABLE layout).
template <typename R> JB_HD R aux_entry(const R* tab, int i, int lane) {
    if (i < LM_A1 || i > LM_B2) return tab[LM_INV + 4 * (i - LM_INV) + lane];
    if (lane == 0) {
        if (i >= LM_A1 && i < LM_A1 + 3) return tab[LM_AM + (i - LM_A1)];
        if (i >= LM_E1 && i < LM_E1 + 3) return tab[LM_EM + (i - LM_E1)];
        if (i >= LM_DC1 && i < LM_DC1 + 3) return tab[LM_DCM + (i - LM_DC1)];
        if (i >= LM_I1 && i < LM_I1 + 6) return tab[LM_IM + (i - LM_I1)];
        if (i == LM_M1) return tab[LM_MM];
    } else if (lane == 1) {
        if (i >= LM_DC1 && i < LM_DC1 + 3) return tab[LM_C0 + (i - LM_DC1)];
        if (i >= LM_I1 && i < LM_I1 + 6) return tab[LM_I0 + (i - LM_I1)];
        if (i == LM_M1) return tab[LM_M0];
    }
    return R(0);
}
constexpr int LM_AUX = 4 * (LM_HOT - LM_INV);        // floats of an aux block, laid out like the per-lane part of a table: [LM_HOT - LM_INV][4]
template <typename R> JB_HD void build_aux_block(const R* tab, R* aux) {
    for (int k = 0; k < LM_AUX; k++) aux[k] = aux_entry<R>(tab, LM_INV + (k >> 2), k & 3);
}
template <typename V> struct LaneModel { LaneConsts<V> c; };
template <typename V> JB_HD V ldc(const LaneModel<V>& m, int i) { return m.c[i]; }


template <typename V> struct Vec3 { V x, y, z; };
template <typename V> JB_HD Vec3<V> v3(const V& x, const V& y, const V& z) { Vec3<V> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename V> JB_HD Vec3<V> ldv3(const LaneModel<V>& m, int i) { return v3(m.c[i], m.c[i + 1], m.c[i + 2]); }
template <typename V> JB_HD Vec3<V> ldc3(const LaneModel<V>& m, int i) { return v3(ldc(m, i), ldc(m, i + 1), ldc(m, i + 2)); }
template <typename V> JB_HD Vec3<V> operator+(const Vec3<V>& a, const Vec3<V>& b) { return v3<V>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename V> JB_HD Vec3<V> operator-(const Vec3<V>& a, const Vec3<V>& b) { return v3<V>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename V> JB_HD Vec3<V> operator-(const Vec3<V>& a) { return v3<V>(-a.x, -a.y, -a.z); }
template <typename V> JB_HD Vec3<V> operator*(const Vec3<V>& a, const V& s) { return v3<V>(a.x * s, a.y * s, a.z * s); }
template <typename V> JB_HD V dot(const Vec3<V>& a, const Vec3<V>& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
template <typename V> JB_HD Vec3<V> cross(const Vec3<V>& a, const Vec3<V>& b) {
    return v3<V>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// acc + a0*b0 + a1*b1 + a2*b2 as a chain of three fused multiply-adds
template <typename V> JB_HD V fma3(const V& acc, const V& a0, const V& b0, const V& a1, const V& b1, const V& a2, const V& b2) {
    V t = acc + a0 * b0;
    t = t + a1 * b1;
    return t + a2 * b2;
}
template <typename V, typename MKT> JB_HD Vec3<V> sel_v3(const MKT& k, const Vec3<V>& a, const Vec3<V>& b) { return v3<V>(sel(k, a.x, b.x), sel(k, a.y, b.y), sel(k, a.z, b.z)); }
template <typename V> JB_HD typename lane_traits<V>::uint zero_u() { return mbit(lt(V(1), V(0))); }
template <typename V> JB_HD Vec3<V> qsum(const Vec3<V>& a) { return v3<V>(quad_sum(a.x), quad_sum(a.y), quad_sum(a.z)); }

// 3x3 general matrix (row major) and symmetric 3x3 (xx yy zz xy xz yz)
template <typename V> struct Mat3 { V m[9]; };
template <typename V> struct Sym3 { V xx, yy, zz, xy, xz, yz; };
template <typename V> JB_HD Vec3<V> mul(const Mat3<V>& R, const Vec3<V>& v) {
    return v3<V>(R.m[0] * v.x + R.m[1] * v.y + R.m[2] * v.z, R.m[3] * v.x + R.m[4] * v.y + R.m[5] * v.z, R.m[6] * v.x + R.m[7] * v.y + R.m[8] * v.z);
}
template <typename V> JB_HD Vec3<V> mulT(const Mat3<V>& R, const Vec3<V>& v) {
    return v3<V>(R.m[0] * v.x + R.m[3] * v.y + R.m[6] * v.z, R.m[1] * v.x + R.m[4] * v.y + R.m[7] * v.z, R.m[2] * v.x + R.m[5] * v.y + R.m[8] * v.z);
}
template <typename V> JB_HD Mat3<V> mul(const Mat3<V>& A, const Mat3<V>& B) {
    Mat3<V> C;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) C.m[3 * i + j] = A.m[3 * i] * B.m[j] + A.m[3 * i + 1] * B.m[3 + j] + A.m[3 * i + 2] * B.m[6 + j];
    return C;
}
template <typename V> JB_HD Vec3<V> mul(const Sym3<V>& S, const Vec3<V>& v) {
    return v3<V>(S.xx * v.x + S.xy * v.y + S.xz * v.z, S.xy * v.x + S.yy * v.y + S.yz * v.z, S.xz * v.x + S.yz * v.y + S.zz * v.z);
}
template <typename V> JB_HD Sym3<V> ldsym(const LaneModel<V>& m, int i) {
    Sym3<V> s; s.xx = m.c[i]; s.yy = m.c[i + 1]; s.zz = m.c[i + 2]; s.xy = m.c[i + 3]; s.xz = m.c[i + 4]; s.yz = m.c[i + 5]; return s;
}
template <typename V> JB_HD Sym3<V> operator+(const Sym3<V>& a, const Sym3<V>& b) {
    Sym3<V> s; s.xx = a.xx + b.xx; s.yy = a.yy + b.yy; s.zz = a.zz + b.zz; s.xy = a.xy + b.xy; s.xz = a.xz + b.xz; s.yz = a.yz + b.yz; return s;
}
template <typename V> JB_HD Sym3<V> qsum(const Sym3<V>& a) {
    Sym3<V> s; s.xx = quad_sum(a.xx); s.yy = quad_sum(a.yy); s.zz = quad_sum(a.zz); s.xy = quad_sum(a.xy); s.xz = quad_sum(a.xz); s.yz = quad_sum(a.yz); return s;
}
// R S R^T for symmetric S
template <typename V> JB_HD Sym3<V> rotate(const Mat3<V>& R, const Sym3<V>& S) {
    V t[9];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        t[3 * i + 0] = R.m[3 * i] * S.xx + R.m[3 * i + 1] * S.xy + R.m[3 * i + 2] * S.xz;
        t[3 * i + 1] = R.m[3 * i] * S.xy + R.m[3 * i + 1] * S.yy + R.m[3 * i + 2] * S.yz;
        t[3 * i + 2] = R.m[3 * i] * S.xz + R.m[3 * i + 1] * S.yz + R.m[3 * i + 2] * S.zz;
    }
    Sym3<V> o;
    o.xx = t[0] * R.m[0] + t[1] * R.m[1] + t[2] * R.m[2];
    o.yy = t[3] * R.m[3] + t[4] * R.m[4] + t[5] * R.m[5];
    o.zz = t[6] * R.m[6] + t[7] * R.m[7] + t[8] * R.m[8];
    o.xy = t[0] * R.m[3] + t[1] * R.m[4] + t[2] * R.m[5];
    o.xz = t[0] * R.m[6] + t[1] * R.m[7] + t[2] * R.m[8];
    o.yz = t[3] * R.m[6] + t[4] * R.m[7] + t[5] * R.m[8];
    return o;
}
// inertia about the origin of a body with COM inertia I, mass m, COM c:  I + m (|c|^2 1 - c c^T)
template <typename V> JB_HD Sym3<V> about_origin(const Sym3<V>& I, const V& m, const Vec3<V>& c) {
    Sym3<V> o;
    V cx2 = c.x * c.x, cy2 = c.y * c.y, cz2 = c.z * c.z;
    o.xx = I.xx + m * (cy2 + cz2); o.yy = I.yy + m * (cx2 + cz2); o.zz = I.zz + m * (cx2 + cy2);
    o.xy = I.xy - m * c.x * c.y; o.xz = I.xz - m * c.x * c.z; o.yz = I.yz - m * c.y * c.z;
    return o;
}
// Rodrigues rotation about unit axis e with (sin, cos)
template <typename V> JB_HD Mat3<V> rodrigues(const Vec3<V>& e, const V& s, const V& c) {
    V v = V(1) - c;
    Mat3<V> R;
    R.m[0] = c + e.x * e.x * v;       R.m[1] = e.x * e.y * v - e.z * s; R.m[2] = e.x * e.z * v + e.y * s;
    R.m[3] = e.y * e.x * v + e.z * s; R.m[4] = c + e.y * e.y * v;       R.m[5] = e.y * e.z * v - e.x * s;
    R.m[6] = e.z * e.x * v - e.y * s; R.m[7] = e.z * e.y * v + e.x * s; R.m[8] = c + e.z * e.z * v;
    return R;
}
template <typename V> JB_HD Mat3<V> quat2mat(const V& w, const V& x, const V& y, const V& z) {
    Mat3<V> R;
    R.m[0] = w * w + x * x - y * y - z * z; R.m[1] = V(2) * (x * y - w * z);         R.m[2] = V(2) * (x * z + w * y);
    R.m[3] = V(2) * (x * y + w * z);         R.m[4] = w * w - x * x + y * y - z * z; R.m[5] = V(2) * (y * z - w * x);
    R.m[6] = V(2) * (x * z - w * y);         R.m[7] = V(2) * (y * z + w * x);         R.m[8] = w * w - x * x - y * y + z * z;
    return R;
}

// ---- two bodies per lane (jb_lane.hpp Pk2): pack two vectors / matrices into one of pairs, take the halves out again, use one for both halves.
// On the device taking a half out is free (a pair IS two registers) and so is `both` (the packed instructions read either half of any pair).
template <typename V> JB_HD Vec3<Pk2<V>> pk(const Vec3<V>& a, const Vec3<V>& b) { return v3<Pk2<V>>(Pk2<V>(a.x, b.x), Pk2<V>(a.y, b.y), Pk2<V>(a.z, b.z)); }
template <typename V> JB_HD Vec3<Pk2<V>> both(const Vec3<V>& a) { return pk(a, a); }
template <typename V> JB_HD Vec3<V> lo(const Vec3<Pk2<V>>& w) { return v3<V>(pk_lo(w.x), pk_lo(w.y), pk_lo(w.z)); }
template <typename V> JB_HD Vec3<V> hi(const Vec3<Pk2<V>>& w) { return v3<V>(pk_hi(w.x), pk_hi(w.y), pk_hi(w.z)); }
template <typename V> JB_HD Mat3<Pk2<V>> pk(const Mat3<V>& a, const Mat3<V>& b) {
    Mat3<Pk2<V>> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.m[i] = Pk2<V>(a.m[i], b.m[i]);
    return r;
}
template <typename V> JB_HD Mat3<Pk2<V>> both(const Mat3<V>& a) { return pk(a, a); }
template <typename V> JB_HD Mat3<V> lo(const Mat3<Pk2<V>>& w) {
    Mat3<V> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.m[i] = pk_lo(w.m[i]);
    return r;
}
template <typename V> JB_HD Mat3<V> hi(const Mat3<Pk2<V>>& w) {
    Mat3<V> r;
#pragma unroll
    for (int i = 0; i < 9; i++) r.m[i] = pk_hi(w.m[i]);
    return r;
}
template <typename V> JB_HD Sym3<Pk2<V>> pk(const Sym3<V>& a, const Sym3<V>& b) {
    Sym3<Pk2<V>> s;
    s.xx = Pk2<V>(a.xx, b.xx); s.yy = Pk2<V>(a.yy, b.yy); s.zz = Pk2<V>(a.zz, b.zz); s.xy = Pk2<V>(a.xy, b.xy); s.xz = Pk2<V>(a.xz, b.xz); s.yz = Pk2<V>(a.yz, b.yz);
    return s;
}
template <typename V> JB_HD Sym3<V> lo(const Sym3<Pk2<V>>& w) { Sym3<V> s; s.xx = pk_lo(w.xx); s.yy = pk_lo(w.yy); s.zz = pk_lo(w.zz); s.xy = pk_lo(w.xy); s.xz = pk_lo(w.xz); s.yz = pk_lo(w.yz); return s; }
template <typename V> JB_HD Sym3<V> hi(const Sym3<Pk2<V>>& w) { Sym3<V> s; s.xx = pk_hi(w.xx); s.yy = pk_hi(w.yy); s.zz = pk_hi(w.zz); s.xy = pk_hi(w.xy); s.xz = pk_hi(w.xz); s.yz = pk_hi(w.yz); return s; }
// centripetal term  w x (w x r) = w (w.r) - r |w|^2  (9 flops + a shared |w|^2 instead of two cross products)
template <typename V> JB_HD Vec3<V> wxwx(const Vec3<V>& ww, const V& ww2, const Vec3<V>& r) {
    V d = dot(ww, r);
    return v3<V>(ww.x * d - r.x * ww2, ww.y * d - r.y * ww2, ww.z * d - r.z * ww2);
}

// ----------------------------------------------------------------------------- lane state
template <typename V> struct LaneState {
    // replicated in the 4 lanes of a quad
    V px, py, pz, qw, qx, qy, qz;        // root pose (world)
    V pz_lo, qw_lo, qx_lo, qy_lo, qz_lo; // low-order words of the height and the quaternion: these five are integrated with
                                         // compensated sums (phase C), so that 50 substeps of fp32 rounding do not move a foot
                                         // across the floor plane (contact activation is the model's one discontinuity)
    V vx, vy, vz;                        // root linear velocity (world)
    V wx, wy, wz;                        // root angular velocity (body frame)
    V phi, phid, turns;                  // motor angle wrapped to [-pi, pi), rate, whole turns
    // lane private: own leg
    V th1, th2, thd1, thd2;
    // warm start of the contact solve: last acceleration [alpha(3), world linear(3)], own leg (2), motor
    V wa[3], wl[3], wj[2], wm;
    V fail;                              // >0: the Newton iteration hit its cap in some substep
#ifdef JB_WAVE_STATS
    V st_fast, st_checks;
    V st_xtra, st_sweeps, st_contact, st_slots;    // diagnostic build only: substeps on the rare path, Newton sweeps, substeps with contact, live slots summed over contact substeps
#endif
};

// ----------------------------------------------------------------------------- per-lane scratch
// Values that one group of lanes produces and another consumes later in the substep (contact-frame direction data, contact
// candidates, cached contact rows, the Newton iterate, the reduction hand-over) live in a per-lane scratch: on the device
// this is LDS with element i of lane L at lds[i*stride + L] (conflict free); on the host a plain array.  Helper lanes use
// the addresses of the main lane they mirror.  (Lane-private long-lived values - model constants, the joint-space system -
// are kept in registers instead, see LaneConsts / StarSys.)
constexpr int NSLOT = 30, SLOT_THREAD = 28, SLOT_MASS = 29;     // PAIR kernels - 28: motor-axis thread against the lane's upper leg, 29: mass ellipsoid against it
constexpr int ROW_F = 13;            // floats per cached contact: x(3), w (the candidate's effective distance until the row is built, then the weight D), jsh(3), j7(3), ahat(3)
constexpr int ROW_K = JB_ROW_K;      // cached live slots per substep: a leg lying on the floor has 8-9 live (10 measured +1.4 % / +3.4 % over 8 for uniform / flat-out actions; 9 is what lets eight
                                     // LEAN waves share a CU's 160 KB of LDS); the contacts
                                     // beyond them keep only their candidate (sc.ovc) and their rows are recomputed in registers in every pass
enum SC : int {
    SC_DD = 0 /*24: contact-frame directions d_k (3x3), own-leg hinge data e1 a1 e2 a2 (4x3), d_k.u (3)*/,
    SC_Y = 24 /*9: current Newton iterate [yr(6), yl(2), ym], shared with the helper groups*/,
    SC_ST = SC_Y + 9 /*6: w(3), thd1, thd2, phid for the helper groups*/,
    SC_OWN = SC_ST + 6 /*1: bitmask of the leg slots (0-9) in which THIS lane's leg has a contact, as a number (spread sweeps)*/,
    SC_ROWS = SC_OWN + 1 /*ROW_K x ROW_F: one entry per LIVE candidate slot of this substep, in slot order (entry = rank of the slot among the
                   live ones).  Phase A writes position (3, root coords rel. root origin) and effective distance (the real distance when the
                   candidate is a contact, +1 otherwise) of a live slot straight into its entry; the row build adds the y-independent
                   rest.  Slots: 0 foot, 1-4 lower-leg cylinder | 5-8 upper cylinder, 9 knee tip | root body: 10-13 lane cylinder, 14 lane
                   ellipsoid, 15-22 lane box vertices | motor body: 23-26 lane cylinder, 27 lane ellipsoid | 28: the mass ellipsoid against
                   the lane's upper-leg cylinder (the geom-geom pair; PAIR kernels only)*/,
    SC_VAR = SC_ROWS + ROW_K * ROW_F /*from here on the layout depends on the kernel variant*/,
    // ---- one wave per SIMD (the ordinary and the PAIR kernels)
    SC_RED = SC_VAR /*56: hand-over of the group reduction's totals to the main lanes*/,
    SC_OVC = SC_RED + 56 /*(NSLOT - ROW_K) x 4: candidates of the live slots beyond the row cache (sc.ovc points here)*/,
    SC_PD = SC_OVC + 4 * (NSLOT - ROW_K) /*9 + 3: contact frame (n, t1, t2) of the pair contact, root coordinates; then the narrow phase's warm start
                   for the next substep of the same control step: axis parameter, multiplier, valid flag*/,
    SC_PD2 = SC_PD + 12 /*9: contact frame of the thread pair contact (slot 28), stored NEGATED - see substep_impl (sc.pd2 points here; LEAN: global memory)*/,
    SC_ZERO = SC_PD2 + 9 /*56 zeros, written once per kernel: what the replica group reads where the main lanes read the reduction's totals (SimOpts::offload)*/,
    SC_COUNT = SC_ZERO + 56,
    // ---- LEAN kernel variant (two waves per SIMD: 20 KB of LDS per wave): long-lived values that the one-wave-per-SIMD kernel keeps in
    // registers are parked here between the phases that use them; no reduction hand-over (the totals are combined in registers), the
    // candidates beyond the row cache live in global memory (sc.ovc)
    SC_SYS = SC_VAR /*41: joint-space system (the root block without its structural zeros)*/, SC_FAC = SC_SYS + 41 /*43: kept factorisation*/, SC_LSTATE = SC_FAC + 43 /*36: lane state*/,
    SC_COUNT_LEAN = SC_LSTATE + 36,
    // LEAN + PAIR (one model per env on the two-waves-per-SIMD kernel): the cross term's share of the parked factorisation, the pair's frame
    SC_CX_LEAN = SC_COUNT_LEAN /*2*/, SC_PD_LEAN = SC_CX_LEAN + 2 /*9 + 3*/, SC_COUNT_LEAN_PAIR = SC_PD_LEAN + 12
};
template <typename V> struct LaneScratch {
    V* p;
    int stride;            // device: number of MAIN lanes of the wave (4 x envs per wave); host: 1
    // Helper groups: when a wave holds fewer than 16 envs its spare lanes form ngrp-1 helper groups that mirror the main
    // lanes (same env, same leg, same scratch addresses) and take a share of the live contact slots in every pass.
    int grp, ngrp, gstride;   // group of this lane (0 = main), number of groups (1, 2 or 4), lane distance between groups
    V* ovc;                   // candidates of the live slots beyond the row cache: [i * ovc_stride], LDS (SC_OVC) or, in the LEAN variant, global memory
    int ovc_stride;
    int pd;                   // where the pair contact's frame lives (SC_PD, or SC_PD_LEAN in the LEAN layout)
    int pd2;                  // the thread pair contact's frame (9 values): sc.ovc[(pd2 + i) * ovc_stride] - behind the overflow candidates, in LDS (SC_PD2 - SC_OVC) or, in the LEAN variant, in global memory (rare path: 0.2 % of the robots)
    bool aux_lane = false;    // this lane is an AUX lane (SimOpts::aux): it runs phase A on a body of its own and must not write the scratch of the leg it mirrors
    bool red_lds;             // the group reduction hands its totals over through SC_RED (false: combined in registers, the LEAN variant has no room for the buffer)
    JB_HD V ld(int i) const { return p[i * stride]; }
    JB_HD void st(int i, const V& v) const { p[i * stride] = v; }
    JB_HD V ldv(const typename lane_traits<V>::uint& i) const { return ld_gather(p, stride, i); }      // per-lane index
    JB_HD Vec3<V> ld3(int i) const { return v3<V>(ld(i), ld(i + 1), ld(i + 2)); }
    JB_HD void st3(int i, const Vec3<V>& v) const { st(i, v.x); st(i + 1, v.y); st(i + 2, v.z); }
};
// sum over the helper groups (every group ends with the same total)
// Butterfly over the group index, HIGH bit first: the order is part of the result's bits, and it is the same for every
// envs-per-wave variant that has 4 groups.
template <typename V> JB_HD V group_sum(const LaneScratch<V>& sc, V x) {
    if (sc.ngrp == 4) { x = xor_sum(x, 2 * sc.gstride, false); x = xor_sum(x, sc.gstride, true); }
    else if (sc.ngrp == 2) x = xor_sum(x, sc.gstride, false);
    return x;
}
template <typename V> JB_HD typename lane_traits<V>::uint group_sum_u(const LaneScratch<V>& sc, typename lane_traits<V>::uint x) {
    if (sc.ngrp == 4) { x = xor_sum_u(x, 2 * sc.gstride, false); x = xor_sum_u(x, sc.gstride, true); }
    else if (sc.ngrp == 2) x = xor_sum_u(x, sc.gstride, false);
    return x;
}
// Line search (newton_phase<LS = true>): where the main lanes leave the direction (9 floats) and the trial step for the sweeps of every lane
// group - the reduction hand-over, or in the LEAN layout the head of the kept factorisation (dead there: the line-searched iteration makes
// no rank-one passes).
template <typename V> JB_HD int ls_mailbox(const LaneScratch<V>& sc) { return sc.red_lds ? (int)SC_RED : (int)SC_FAC; }
JB_HD constexpr int tri(int i, int j) { return i >= j ? i * (i + 1) / 2 + j : j * (j + 1) / 2 + i; }

// ---- packed layouts of the solver (jb_lane.hpp Pk2: one packed instruction for two values)
// A symmetric 6x6 block (the root block of M, of the Newton system, its Schur complement and Cholesky factor): the LOWER triangle, stored
// by COLUMNS in pairs of ROWS - pair (ip, j) holds rows 2 ip and 2 ip + 1 of column j, for 2 ip + 1 >= j: 12 pairs.  Three low halves lie
// above the diagonal (rows 0 / 2 / 4 of columns 1 / 3 / 5): they carry the mirrored element or junk and are never read as part of the
// triangle.  Everything the solver does to such a block is "pair of rows op one column element": rank-3 updates from contact rows, the
// Schur complement's updates, right-looking Cholesky and the forward substitution all run on pairs, no horizontal sums.
// A 6-vector is three pairs of rows; the 6 x 2 leg coupling is [row pair][column].
JB_HD constexpr int s6base(int j) { return j == 0 ? 0 : j == 1 ? 3 : j == 2 ? 6 : j == 3 ? 8 : j == 4 ? 10 : 11; }
JB_HD constexpr int s6p(int ip, int j) { return s6base(j) + ip - j / 2; }
template <typename V> struct Sym6P { Pk2<V> p[12]; };
template <typename V> JB_HD V pk_half(const Pk2<V>& w, int h) { return h ? pk_hi(w) : pk_lo(w); }
template <typename V> JB_HD Pk2<V> pk_with(const Pk2<V>& w, int h, const V& x) { return h ? Pk2<V>(pk_lo(w), x) : Pk2<V>(x, pk_hi(w)); }
template <typename V> JB_HD V s6get(const Sym6P<V>& S, int i, int j) { return i >= j ? pk_half(S.p[s6p(i >> 1, j)], i & 1) : pk_half(S.p[s6p(j >> 1, i)], j & 1); }
template <typename V> JB_HD void s6set(Sym6P<V>& S, int i, int j, const V& x) { S.p[s6p(i >> 1, j)] = pk_with(S.p[s6p(i >> 1, j)], i & 1, x); }      // i >= j
template <typename V> JB_HD V v6get(const Pk2<V> (&v)[3], int i) { return pk_half(v[i >> 1], i & 1); }
template <typename V> JB_HD void v6set(Pk2<V> (&v)[3], int i, const V& x) { v[i >> 1] = pk_with(v[i >> 1], i & 1, x); }
template <typename V> JB_HD V b6get(const Pk2<V> (&B)[3][2], int i, int c) { return pk_half(B[i >> 1][c], i & 1); }
template <typename V, typename MKT> JB_HD Pk2<V> selw(const MKT& k, const Pk2<V>& a, const Pk2<V>& b) { return Pk2<V>(sel(k, pk_lo(a), pk_lo(b)), sel(k, pk_hi(a), pk_hi(b))); }
template <typename V> JB_HD Pk2<V> qsum2(const Pk2<V>& a) { return Pk2<V>(quad_sum(pk_lo(a)), quad_sum(pk_hi(a))); }
// ... of a pair whose low half lies above the diagonal (left as it is)
template <typename V> JB_HD Pk2<V> qsum_hi(const Pk2<V>& a) { return Pk2<V>(pk_lo(a), quad_sum(pk_hi(a))); }
JB_HD constexpr bool s6_junk_lo(int ip, int j) { return 2 * ip < j; }

// Accumulator of the contact terms of the Newton system for the lane (everything here is ADDED to M / tau,
// which stay in the scratch)
template <typename V> struct NewtonAcc {
    Sym6P<V> A;              // lane-private additive part of the root block
    Pk2<V> B[3][2];          // additive part of the leg coupling
    V C11, C12, C22;
    V X;                     // PAIR kernels: the own shoulder - motor cross term of the pair contact (mass against the own upper leg)
    Pk2<V> Bm[3]; V Cm;      // lane-private additive part of the motor branch
    Pk2<V> rr[3]; V rl[2], rm;      // additive rhs parts
    typename lane_traits<V>::uint bw0, bw1;   // active-set records of the leg slots: 5 bits (4 pyramid edges + valid) each, exact; bw0: slots 0-4, bw1: slots 5-9
    typename lane_traits<V>::uint xh;      // polynomial hash of the records of the rarely-evaluated slots (lane-private, never summed across lanes)
    V ls_g, ls_h, ls_a;      // line search (sweeps of mode 3): sums over this lane's contacts and pyramid edges of D min(0, r + alpha s) s, of D s^2 over the
                             // edges active at alpha, and of |D r s| over the edges active at alpha (the size of the terms phi' is a difference of)
};
// direction and step of the line search, handed to the sweeps of mode 3
template <typename V> struct LineDir { V dr[6], dl[2], dm, alpha; };
template <typename V> JB_HD void acc_clear(NewtonAcc<V>& acc) {
    using W = Pk2<V>;
#pragma unroll
    for (int i = 0; i < 12; i++) acc.A.p[i] = W(0);
#pragma unroll
    for (int i = 0; i < 3; i++) { acc.B[i][0] = W(0); acc.B[i][1] = W(0); acc.Bm[i] = W(0); acc.rr[i] = W(0); }
    acc.C11 = V(0); acc.C12 = V(0); acc.C22 = V(0); acc.Cm = V(0); acc.X = V(0);
    acc.rl[0] = V(0); acc.rl[1] = V(0); acc.rm = V(0);
    acc.bw0 = zero_u<V>(); acc.bw1 = zero_u<V>(); acc.xh = zero_u<V>();
}

// the 52 additive values of the accumulator as a flat list (and back); PAIR kernels carry X as value 52 (callers pad to 56)
template <typename V> JB_HD void acc_pack(const NewtonAcc<V>& a, V* v) {
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) v[tri(i, j)] = s6get(a.A, i, j);
#pragma unroll
    for (int i = 0; i < 6; i++) { v[21 + 2 * i] = b6get(a.B, i, 0); v[22 + 2 * i] = b6get(a.B, i, 1); v[36 + i] = v6get(a.Bm, i); v[43 + i] = v6get(a.rr, i); }
    v[33] = a.C11; v[34] = a.C12; v[35] = a.C22; v[42] = a.Cm; v[49] = a.rl[0]; v[50] = a.rl[1]; v[51] = a.rm;
}
template <typename V> JB_HD void acc_unpack(const V* v, NewtonAcc<V>& a) {
    using W = Pk2<V>;
    // (the halves above the diagonal take the mirrored element: defined values, never read)
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
        for (int ip = j / 2; ip < 3; ip++) a.A.p[s6p(ip, j)] = W(v[tri(2 * ip, j)], v[tri(2 * ip + 1, j)]);
#pragma unroll
    for (int ip = 0; ip < 3; ip++) {
        a.B[ip][0] = W(v[21 + 4 * ip], v[23 + 4 * ip]); a.B[ip][1] = W(v[22 + 4 * ip], v[24 + 4 * ip]);
        a.Bm[ip] = W(v[36 + 2 * ip], v[37 + 2 * ip]); a.rr[ip] = W(v[43 + 2 * ip], v[44 + 2 * ip]);
    }
    a.C11 = v[33]; a.C12 = v[34]; a.C22 = v[35]; a.Cm = v[42]; a.rl[0] = v[49]; a.rl[1] = v[50]; a.rm = v[51];
}

// The joint-space system of the substep, M = [A B Bm; B^T C 0; Bm^T 0 Cm] and tau.  Only the main lanes ever read it
// (solves, final pass), so it lives in registers - in practice in the otherwise unused accumulation registers, one move
// away - instead of making ~130 LDS round trips per substep.
template <typename V> struct StarSys {
    Sym6P<V> A;           // root block
    Pk2<V> B[3][2];       // own leg coupling, [row pair][column: shoulder, knee]
    V C[3];               // own leg block: 11, 12, 22
    Pk2<V> Bm[3]; V Cm;   // motor coupling and block
    Pk2<V> tr[3]; V tl[2], tm;   // applied + bias forces: root, own leg, motor
};

// Factorisation of H = M + contact terms (+ hb on the leg diagonal): the leg 2x2 and motor 1x1 blocks inverted, the 6x6 Schur
// complement Cholesky-factored (reciprocal pivots on the diagonal).  Kept by the caller between Newton passes: a pass
// whose active set differs from the factored one by a single pyramid edge is a rank-one update of this factorisation.
template <typename V> struct StarFactor {
    Sym6P<V> S;
    Pk2<V> B[3][2];
    V i11, i12, i22;
    Pk2<V> bm[3]; V icm;
    V cx0, cx1;     // PAIR kernels: C^-1 [X, 0]^T, the leg block's answer to the shoulder - motor cross term
};
// WITH_ACC = false: no contact terms (the final pass): the factorisation of M + hb alone; `acc` is not read.
// PAIR: the pair contact (eccentric mass against the own upper leg) couples the lane's shoulder with the motor, H[shoulder, motor] = X.
// Eliminating the legs first turns that into a correction of the motor branch, and the rest of the elimination is unchanged:
//     bm' = bm - sum_legs B C^-1 x,    cm' = cm - sum_legs x^T C^-1 x,    rm' = rm - sum_legs x^T C^-1 r_leg,    x = [X, 0]^T,
//     y_leg = C^-1 (r_leg - B^T y_root) - (C^-1 x) y_motor.
template <typename V, bool WITH_ACC = true, bool PAIR = false>
JB_HD void star_factor(const StarSys<V>& M, const NewtonAcc<V>& acc, const V& hb1, const V& hb2, StarFactor<V>& F) {
    using W = Pk2<V>;
#pragma unroll
    for (int ip = 0; ip < 3; ip++) {
        F.B[ip][0] = WITH_ACC ? M.B[ip][0] + acc.B[ip][0] : M.B[ip][0];
        F.B[ip][1] = WITH_ACC ? M.B[ip][1] + acc.B[ip][1] : M.B[ip][1];
    }
    V C11 = M.C[0] + hb1, C12 = M.C[1], C22 = M.C[2] + hb2;
    if (WITH_ACC) { C11 = M.C[0] + acc.C11 + hb1; C12 = M.C[1] + acc.C12; C22 = M.C[2] + acc.C22 + hb2; }
    V idet = vrcp(C11 * C22 - C12 * C12);
    F.i11 = C22 * idet; F.i12 = -C12 * idet; F.i22 = C11 * idet;
    Sym6P<V>& S = F.S;
    W g[3][2];          // B C^-1, row pairs
#pragma unroll
    for (int ip = 0; ip < 3; ip++) {
        g[ip][0] = F.B[ip][0] * W(F.i11) + F.B[ip][1] * W(F.i12);
        g[ip][1] = F.B[ip][0] * W(F.i12) + F.B[ip][1] * W(F.i22);
    }
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const V bj0 = b6get(F.B, j, 0), bj1 = b6get(F.B, j, 1);
#pragma unroll
        for (int ip = j / 2; ip < 3; ip++) {
            const W leg = g[ip][0] * W(bj0) + g[ip][1] * W(bj1);
            const int k = s6p(ip, j);
            if (WITH_ACC) { const W t = acc.A.p[k] - leg; S.p[k] = (s6_junk_lo(ip, j) ? qsum_hi(t) : qsum2(t)) + M.A.p[k]; }
            else S.p[k] = M.A.p[k] - (s6_junk_lo(ip, j) ? qsum_hi(leg) : qsum2(leg));
        }
    }
    // motor branch
    V cm = M.Cm;
#pragma unroll
    for (int ip = 0; ip < 3; ip++) F.bm[ip] = M.Bm[ip];
    if (WITH_ACC) {
#pragma unroll
        for (int ip = 0; ip < 3; ip++) F.bm[ip] = F.bm[ip] + qsum2(acc.Bm[ip]);
        cm = cm + quad_sum(acc.Cm);
    }
    F.cx0 = V(0); F.cx1 = V(0);
    if (WITH_ACC && PAIR) {
        F.cx0 = F.i11 * acc.X; F.cx1 = F.i12 * acc.X;
#pragma unroll
        for (int ip = 0; ip < 3; ip++) F.bm[ip] = F.bm[ip] - qsum2(F.B[ip][0] * W(F.cx0) + F.B[ip][1] * W(F.cx1));
        cm = cm - quad_sum(acc.X * F.cx0);
    }
    F.icm = vrcp(cm);
    {
        W gm[3];
#pragma unroll
        for (int ip = 0; ip < 3; ip++) gm[ip] = F.bm[ip] * W(F.icm);
#pragma unroll
        for (int j = 0; j < 6; j++) {
            const V bmj = v6get(F.bm, j);
#pragma unroll
            for (int ip = j / 2; ip < 3; ip++) S.p[s6p(ip, j)] = S.p[s6p(ip, j)] - gm[ip] * W(bmj);
        }
    }
    // Cholesky S = L L^T in place, right-looking: column j is finished (reciprocal pivot on the diagonal, the rest scaled), then taken off
    // the columns to its right - pairs of rows times one element of column j.  Every element sees the subtractions of the left-looking form
    // in the same order.
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const int jp = j >> 1;
        const V id = vrsqrt(s6get(S, j, j));
        if ((j & 1) == 0) S.p[s6p(jp, j)] = W(id, pk_hi(S.p[s6p(jp, j)]) * id);
        else S.p[s6p(jp, j)] = W(pk_lo(S.p[s6p(jp, j)]), id);
#pragma unroll
        for (int ip = jp + 1; ip < 3; ip++) S.p[s6p(ip, j)] = S.p[s6p(ip, j)] * W(id);
#pragma unroll
        for (int c = j + 1; c < 6; c++) {
            const V lcj = s6get(S, c, j);
#pragma unroll
            for (int ip = c / 2; ip < 3; ip++) S.p[s6p(ip, c)] = S.p[s6p(ip, c)] - S.p[s6p(ip, j)] * W(lcj);
        }
    }
}
// H y = rhs with the factorisation above.  rr: lane-private parts of the root rhs (summed over the quad), tr: replicated
// root rhs (added once), rl0/rl1: own leg rhs, rmt: total motor rhs (replicated).
template <typename V, bool PAIR = false>
JB_HD void star_subst(const StarFactor<V>& F, const Pk2<V> (&rr)[3], const Pk2<V> (&tr)[3], const V& rl0, const V& rl1, const V& rmt_in, Pk2<V> (&yr)[3], V (&yl)[2], V& ym) {
    using W = Pk2<V>;
    const Sym6P<V>& S = F.S;
    W r[3];
    V rmt = rmt_in;
    if (PAIR) rmt = rmt_in - quad_sum(F.cx0 * rl0 + F.cx1 * rl1);
#pragma unroll
    for (int ip = 0; ip < 3; ip++) {
        const W g0 = F.B[ip][0] * W(F.i11) + F.B[ip][1] * W(F.i12), g1 = F.B[ip][0] * W(F.i12) + F.B[ip][1] * W(F.i22);
        r[ip] = qsum2(rr[ip] - (g0 * W(rl0) + g1 * W(rl1))) + tr[ip];
    }
#pragma unroll
    for (int ip = 0; ip < 3; ip++) r[ip] = r[ip] - (F.bm[ip] * W(F.icm)) * W(rmt);
    // L z = r, column by column: z_j leaves, and column j times z_j comes off the rows below (the pair that holds row j itself goes along:
    // its slot is not read again)
    V y[6];
#pragma unroll
    for (int j = 0; j < 6; j++) {
        y[j] = v6get(r, j) * s6get(S, j, j);
#pragma unroll
        for (int ip = (j + 1) / 2; ip < 3; ip++) r[ip] = r[ip] - S.p[s6p(ip, j)] * W(y[j]);
    }
    // L^T y = z
#pragma unroll
    for (int i = 5; i >= 0; i--) {
        V t = y[i];
#pragma unroll
        for (int k = i + 1; k < 6; k++) t = t - s6get(S, k, i) * y[k];
        y[i] = t * s6get(S, i, i);
    }
#pragma unroll
    for (int ip = 0; ip < 3; ip++) yr[ip] = W(y[2 * ip], y[2 * ip + 1]);
    // back-substitute the branches
    W u0 = F.B[0][0] * yr[0], u1 = F.B[0][1] * yr[0], um = F.bm[0] * yr[0];
#pragma unroll
    for (int ip = 1; ip < 3; ip++) { u0 = u0 + F.B[ip][0] * yr[ip]; u1 = u1 + F.B[ip][1] * yr[ip]; um = um + F.bm[ip] * yr[ip]; }
    const V t0 = rl0 - (pk_lo(u0) + pk_hi(u0)), t1 = rl1 - (pk_lo(u1) + pk_hi(u1)), tmm = rmt - (pk_lo(um) + pk_hi(um));
    ym = tmm * F.icm;
    yl[0] = F.i11 * t0 + F.i12 * t1;
    yl[1] = F.i12 * t0 + F.i22 * t1;
    if (PAIR) { yl[0] = yl[0] - F.cx0 * ym; yl[1] = yl[1] - F.cx1 * ym; }
}
// Solve  [A B Bm; B^T C 0; Bm^T 0 Cm] y = rhs  where the matrix is M + the contact terms in `acc` (+ hb1/hb2 on the leg
// diagonal: implicit joint damping) and rhs = tau + acc.r*.
//   A, tau_root, Bm, Cm : replicated in the 4 lanes, added ONCE;  acc.A, acc.rr, acc.Bm/Cm/rm: lane-private
//   parts that are summed over the quad;  B, C, tau_leg: lane-private leg branch.
// The own leg (2x2) and the motor (1x1) are eliminated onto the 6 root dofs; the 6x6 Schur complement is
// Cholesky-factored redundantly by the 4 lanes.
template <typename V, bool PAIR = false>
JB_HD void star_solve(const StarSys<V>& M, const NewtonAcc<V>& acc, const V& hb1, const V& hb2, StarFactor<V>& F, Pk2<V> (&yr)[3], V (&yl)[2], V& ym) {
    star_factor<V, true, PAIR>(M, acc, hb1, hb2, F);
    star_subst<V, PAIR>(F, acc.rr, M.tr, M.tl[0] + acc.rl[0], M.tl[1] + acc.rl[1], M.tm + quad_sum(acc.rm), yr, yl, ym);
}

// ----------------------------------------------------------------------------- contacts
// MuJoCo solimp impedance d(r) (5-parameter form)
template <typename V> JB_HD V impedance(const LaneModel<V>& m, const V& dist) {
    V x = vmin(vabs(dist) * m.c[LM_IMP_IW], V(1));
    V mid = m.c[LM_IMP_MID];
    // power = 2 (MuJoCo default); other powers are rejected by the host when the table is built
    V ya = x * x * m.c[LM_IMP_IMID];
    V omx = V(1) - x;
    V yb = V(1) - omx * omx * m.c[LM_IMP_I1MID];
    V y = sel(lt(x, mid), ya, yb);
    return m.c[LM_IMP_D0] + y * (m.c[LM_IMP_DW] - m.c[LM_IMP_D0]);
}


// ---- contact rows.  Everything about a contact that does not depend on the iterate y is computed ONCE per substep
// and kept in the scratch row cache (19 floats per live slot): the three rows of B without their shared linear part
// [x cross d_k (3), J_sh, J_7] for k = n, t1, t2, the reference accelerations ahat (3) and the weight D (0 for a
// lane without this contact).  A Newton pass then costs one batch of LDS reads and ~30 FMAs per live slot to get the
// residuals, plus the rank-3 update when it accumulates.
// level: 0 root body, 1 upper leg (shoulder only), 2 lower leg (shoulder+knee), 3 motor body; column 7 of a row belongs
// to the knee (level 2) or to the motor (level 3).
// Direction data in the scratch (SC_DD): the three contact-frame directions d_k, the own leg's hinge axes / anchors and d_k.u.
//   J_sh(x,d) = (d x e1).(x - a1) = d.(e1 x (x - a1)),   J_kn(x,d) = d.(e2 x (x - a2)),   J_m(x,d) = d.(em x (x - am)):
// one cross product per hinge and contact, then a dot product per direction.
// 4: the pair contact - upper leg (shoulder column) against the motor body (motor column), no root columns
JB_HD constexpr int slot_level(int slot) { return slot < 5 ? 2 : slot < 10 ? 1 : slot < 23 ? 0 : slot < 28 ? 3 : 4; }

// the y-independent part of one contact: position, weight, the joint columns and reference accelerations of its three directions
template <typename V> struct RowVals { Vec3<V> x; V D; V jsh[3], j7[3], ah[3]; };

// The pair contact's rows: relative motion of the upper leg (geom2's body: + shoulder column) and the motor body (geom1's: - motor
// column) at the contact point along the pair's own frame (sc.pd); the root columns cancel (contact_apply masks them).
template <typename V>
JB_HD void pair_row_values(const LaneModel<V>& m, const LaneScratch<V>& sc, const Vec3<V>& x, const V& dist, RowVals<V>& r, const bool second = false) {
    const V thd1 = sc.ld(SC_ST + 3), phid = sc.ld(SC_ST + 5);
    const V mu = m.c[LM_MU];
    const auto valid = lt(dist, V(0));
    const V imp = impedance(m, dist);
    const V invD = (V(1) - imp) * ((m.c.tran_of(1) + m.c.tran_of(3)) * ((V(1) + m.c[LM_FR2]) * (V(2) * mu * mu)));
    r.x = x;
    r.D = sel(valid, imp * vrcp(invD), V(0));
    const Vec3<V> p1 = cross(sc.ld3(SC_DD + 9), x - sc.ld3(SC_DD + 12));      // e1 x (x - a1)
    const Vec3<V> pm = cross(ldv3(m, LM_EM), x - ldv3(m, LM_AM));              // em x (x - am)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const Vec3<V> d = second ? v3<V>(sc.ovc[(sc.pd2 + 3 * k) * sc.ovc_stride], sc.ovc[(sc.pd2 + 3 * k + 1) * sc.ovc_stride], sc.ovc[(sc.pd2 + 3 * k + 2) * sc.ovc_stride]) : sc.ld3(sc.pd + 3 * k);
        r.jsh[k] = dot(d, p1); r.j7[k] = -dot(d, pm);
        r.ah[k] = -m.c[LM_BB] * (r.jsh[k] * thd1 + r.j7[k] * phid);
        if (k == 0) r.ah[k] = r.ah[k] - m.c[LM_KK] * imp * dist;
    }
}

// position x and effective distance of a candidate -> its row values
template <typename V, bool PAIR = false>
JB_HD void row_values(const LaneModel<V>& m, const LaneScratch<V>& sc, bool xtra, int slot, const Vec3<V>& x, const V& dist, RowVals<V>& r) {
    if (PAIR && slot >= SLOT_THREAD) { pair_row_values<V>(m, sc, x, dist, r, slot == SLOT_THREAD); return; }
    const Vec3<V> w = sc.ld3(SC_ST);
    const V thd1 = sc.ld(SC_ST + 3), thd2 = sc.ld(SC_ST + 4), phid = sc.ld(SC_ST + 5);
    const int level = slot_level(slot);
    const V f_sh = V((level == 1 || level == 2) ? 1.0f : 0.0f), f_kn = V(level == 2 ? 1.0f : 0.0f), f_m = V(level == 3 ? 1.0f : 0.0f);
    const V tran = m.c.tran_of(level);
    const V mu = m.c[LM_MU];
    const auto valid = lt(dist, V(0));
    V imp = impedance(m, dist);
    // weight of a pyramid edge  D = 1 / (2 mu^2 R),  R = (1 - d)/d * tran * (1 + mu^2)   (MuJoCo's diagApprox regulariser)
    const V invD = (V(1) - imp) * (tran * ((V(1) + m.c[LM_FR2]) * (V(2) * mu * mu)));
    r.x = x;
    r.D = sel(valid, imp * vrcp(invD), V(0));
    const V jdot = f_kn * thd2 + f_m * phid;                    // the rate column 7 multiplies
    const Vec3<V> p1 = cross(sc.ld3(SC_DD + 9), x - sc.ld3(SC_DD + 12)), p2 = cross(sc.ld3(SC_DD + 15), x - sc.ld3(SC_DD + 18));
    Vec3<V> pm = v3<V>(V(0), V(0), V(0));
    if (xtra) pm = cross(ldv3(m, LM_EM), x - ldv3(m, LM_AM));
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const Vec3<V> d = sc.ld3(SC_DD + 3 * k);
        const Vec3<V> ang = cross(x, d);
        V jsh = f_sh * dot(d, p1);
        V j7 = f_kn * dot(d, p2);
        if (xtra) j7 = j7 + f_m * dot(d, pm);
        V vel = dot(ang, w) + sc.ld(SC_DD + 21 + k) + jsh * thd1 + j7 * jdot;
        V ah = -m.c[LM_BB] * vel;
        if (k == 0) ah = ah - m.c[LM_KK] * imp * dist;
        r.jsh[k] = jsh; r.j7[k] = j7; r.ah[k] = ah;
    }
}
// a cached entry: phase A left position and effective distance in it; the build adds the rest (the weight takes the distance's place)
template <typename V, bool PAIR = false>
JB_HD void contact_rows_build(const LaneModel<V>& m, const LaneScratch<V>& sc, bool xtra, int slot, int entry) {
    const int e0 = SC_ROWS + ROW_F * entry;
    RowVals<V> r;
    row_values<V, PAIR>(m, sc, xtra, slot, sc.ld3(e0), sc.ld(e0 + 3), r);
    sc.st(e0 + 3, r.D);
#pragma unroll
    for (int k = 0; k < 3; k++) { sc.st(e0 + 4 + k, r.jsh[k]); sc.st(e0 + 7 + k, r.j7[k]); sc.st(e0 + 10 + k, r.ah[k]); }
}
template <typename V> JB_HD void row_load(const LaneScratch<V>& sc, int entry, RowVals<V>& r) {
    const int e0 = SC_ROWS + ROW_F * entry;
    r.x = sc.ld3(e0); r.D = sc.ld(e0 + 3);
#pragma unroll
    for (int k = 0; k < 3; k++) { r.jsh[k] = sc.ld(e0 + 4 + k); r.j7[k] = sc.ld(e0 + 7 + k); r.ah[k] = sc.ld(e0 + 10 + k); }
}

// the share of one contact (residuals rho at y, slopes sg along the direction; weight D) in phi'(alpha), phi''(alpha) of the line search
template <typename V>
JB_HD void line_terms(const V& D, const V& mu, const V (&rho)[3], const V (&sg)[3], const V& alpha, NewtonAcc<V>& acc) {
    const V mr1 = mu * rho[1], mr2 = mu * rho[2], ms1 = mu * sg[1], ms2 = mu * sg[2];
    const V r[4] = {rho[0] + mr1, rho[0] - mr1, rho[0] + mr2, rho[0] - mr2}, sv[4] = {sg[0] + ms1, sg[0] - ms1, sg[0] + ms2, sg[0] - ms2};
#pragma unroll
    for (int e = 0; e < 4; e++) {
        const V ra = r[e] + alpha * sv[e];
        const auto on = lt(ra, V(0));
        const V t = D * ra * sv[e];
        acc.ls_g = acc.ls_g + sel(on, t, V(0));
        acc.ls_h = acc.ls_h + sel(on, D * sv[e] * sv[e], V(0));
        acc.ls_a = acc.ls_a + sel(on, vabs(t), V(0));
    }
}

// A contact's three rows of B (normal, two tangents) as pairs of COLUMNS: (ang.x, ang.y) (ang.z, d.x) (d.y, d.z) (J_shoulder, J_7) - what the
// packed accumulation below multiplies with single elements of W B.
template <typename V> struct RowPairs { Pk2<V> b[3][4]; V ah[3]; };
template <typename V, bool PAIR = false>
JB_HD void row_pairs(const RowVals<V>& rv, const Vec3<V> (&dk)[3], const V& lin, const V (&jsh)[3], const V (&j7)[3], const V (&ah)[3], RowPairs<V>& R) {
    using W = Pk2<V>;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const Vec3<V> ang = cross(rv.x, dk[k]);    // (recomputed from the position: 6 instructions against 3 LDS reads and 9 floats of row cache)
        R.b[k][0] = W(ang.x, ang.y); R.b[k][1] = W(ang.z, dk[k].x); R.b[k][2] = W(dk[k].y, dk[k].z); R.b[k][3] = W(jsh[k], j7[k]);
        if (PAIR) {
#pragma unroll
            for (int q = 0; q < 3; q++) R.b[k][q] = R.b[k][q] * W(lin);
        }
        R.ah[k] = ah[k];
    }
}
// residuals rho_k = B_k y - ahat_k of the three rows at the iterate (root part as row pairs, then the two joint columns)
template <typename V> JB_HD void row_residuals(const RowPairs<V>& R, const Pk2<V> (&yr)[3], const Pk2<V>& yj, V (&rho)[3]) {
#pragma unroll
    for (int k = 0; k < 3; k++) {
        Pk2<V> t = R.b[k][0] * yr[0];
        t = t + R.b[k][1] * yr[1]; t = t + R.b[k][2] * yr[2]; t = t + R.b[k][3] * yj;
        rho[k] = (pk_lo(t) - R.ah[k]) + pk_hi(t);
    }
}
// W B for the active set (weights of the pyramid edges folded into a 3 x 3 weight matrix), column pairs like B
template <typename V> struct WeightedRows { Pk2<V> wb[3][4]; V wa[3]; };
template <typename V>
JB_HD void weighted_rows(const RowPairs<V>& R, const V& D, const V& mu, const V& f1, const V& f2, const V& f3, const V& f4, WeightedRows<V>& Q) {
    using W = Pk2<V>;
    const V Wnn = D * (f1 + f2 + f3 + f4), Wn1 = D * mu * (f1 - f2), Wn2 = D * mu * (f3 - f4), W11 = D * mu * mu * (f1 + f2), W22 = D * mu * mu * (f3 + f4);
    Q.wa[0] = Wnn * R.ah[0] + Wn1 * R.ah[1] + Wn2 * R.ah[2];
    Q.wa[1] = Wn1 * R.ah[0] + W11 * R.ah[1];
    Q.wa[2] = Wn2 * R.ah[0] + W22 * R.ah[2];
#pragma unroll
    for (int q = 0; q < 4; q++) {
        Q.wb[0][q] = W(Wnn) * R.b[0][q] + W(Wn1) * R.b[1][q] + W(Wn2) * R.b[2][q];
        Q.wb[1][q] = W(Wn1) * R.b[0][q] + W(W11) * R.b[1][q];
        Q.wb[2][q] = W(Wn2) * R.b[0][q] + W(W22) * R.b[2][q];
    }
}
// element c (a column of B, 0-7) of row k
template <typename V> JB_HD V wb_get(const WeightedRows<V>& Q, int k, int c) { return pk_half(Q.wb[k][c >> 1], c & 1); }
template <typename V> JB_HD V rb_get(const RowPairs<V>& R, int k, int c) { return pk_half(R.b[k][c >> 1], c & 1); }
// the root block's share  A += B^T W B,  rr += B^T W ahat : pairs of rows of B^T (= column pairs of B) times single elements of W B
template <typename V> JB_HD void acc_root(const RowPairs<V>& R, const WeightedRows<V>& Q, NewtonAcc<V>& acc) {
    using W = Pk2<V>;
#pragma unroll
    for (int j = 0; j < 6; j++) {
        const W w0 = W(wb_get(Q, 0, j)), w1 = W(wb_get(Q, 1, j)), w2 = W(wb_get(Q, 2, j));
#pragma unroll
        for (int ip = j / 2; ip < 3; ip++) acc.A.p[s6p(ip, j)] = fma3(acc.A.p[s6p(ip, j)], R.b[0][ip], w0, R.b[1][ip], w1, R.b[2][ip], w2);
    }
#pragma unroll
    for (int ip = 0; ip < 3; ip++) acc.rr[ip] = fma3(acc.rr[ip], R.b[0][ip], W(Q.wa[0]), R.b[1][ip], W(Q.wa[1]), R.b[2][ip], W(Q.wa[2]));
}
// init + sum_k B_k[c] (W B)_k[row pair ip]: what column c of B adds to a coupling block's row pair (and to a single element, to a right-hand side)
template <typename V> JB_HD Pk2<V> col_times_wb(const RowPairs<V>& R, const WeightedRows<V>& Q, int c, int ip, const Pk2<V>& init) {
    using W = Pk2<V>;
    return fma3(init, W(rb_get(R, 0, c)), Q.wb[0][ip], W(rb_get(R, 1, c)), Q.wb[1][ip], W(rb_get(R, 2, c)), Q.wb[2][ip]);
}
template <typename V> JB_HD V col_times_col(const RowPairs<V>& R, const WeightedRows<V>& Q, int c, int d, const V& init) {
    return fma3(init, rb_get(R, 0, c), wb_get(Q, 0, d), rb_get(R, 1, c), wb_get(Q, 1, d), rb_get(R, 2, c), wb_get(Q, 2, d));
}
template <typename V> JB_HD V col_times_wa(const RowPairs<V>& R, const WeightedRows<V>& Q, int c, const V& init) {
    return fma3(init, rb_get(R, 0, c), Q.wa[0], rb_get(R, 1, c), Q.wa[1], rb_get(R, 2, c), Q.wa[2]);
}

// One cached contact against the iterate y.  mode 0: accumulate the Newton matrix / rhs terms for the active set at y;
// mode 2: only record the active set (the cheap convergence check); mode 3 (LS instantiations only): the contact's share of the
// line search's phi' and phi'' at y + ld->alpha * (ld's direction).
template <typename V, bool PAIR = false, bool LS = false>
JB_HD void contact_apply(const RowVals<V>& rv, const Vec3<V> (&dk)[3], const V& mu, int slot, bool lane_on, int mode,
                         const Pk2<V> (&yr)[3], const V (&yl)[2], const V& ym, NewtonAcc<V>& acc, const LineDir<V>* ld = nullptr) {
    using U = typename lane_traits<V>::uint;
    using W = Pk2<V>;
    const int level = slot_level(slot);
    const bool is_pair = PAIR && level == 4;
    const bool has_sh = (level == 1 || level == 2 || is_pair), has_kn = (level == 2), has_m = (level == 3 || is_pair);
    const V lin = V(is_pair ? 0.0f : 1.0f);        // the pair contact has no root columns (the cached angular part is zero, the shared linear part is masked)
    V rho[3];
    const V D = lane_on ? rv.D : V(0);             // a lane without a slot in this round contributes nothing
    const V y7 = has_kn ? yl[1] : ym;              // levels 0/1 have a zero column 7
    RowPairs<V> R;
    row_pairs<V, PAIR>(rv, dk, lin, rv.jsh, rv.j7, rv.ah, R);
    row_residuals<V>(R, yr, W(yl[0], y7), rho);
    const auto valid = gt(D, V(0));
    if (LS && mode == 3) {
        const V d7 = has_kn ? ld->dl[1] : ld->dm;
        const W dq[3] = {W(ld->dr[0], ld->dr[1]), W(ld->dr[2], ld->dr[3]), W(ld->dr[4], ld->dr[5])};
        V sg[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            W t = R.b[k][0] * dq[0];
            t = t + R.b[k][1] * dq[1]; t = t + R.b[k][2] * dq[2]; t = t + R.b[k][3] * W(ld->dl[0], d7);
            sg[k] = pk_lo(t) + pk_hi(t);
        }
        line_terms<V>(D, mu, rho, sg, ld->alpha, acc);
        return;
    }
    // pyramid edges  r = rho_n +- mu rho_t
    V mr1 = mu * rho[1], mr2 = mu * rho[2];
    auto a1 = lt(rho[0] + mr1, V(0)), a2 = lt(rho[0] - mr1, V(0)), a3 = lt(rho[0] + mr2, V(0)), a4 = lt(rho[0] - mr2, V(0));
    {   // active-set record: exact 5-bit fields for the always-evaluated slots 0..4, a lane-private hash for the rest
        U bits = (mbit(a1) + mbit(a2) * 2u + mbit(a3) * 4u + mbit(a4) * 8u + 16u);
        if (slot < 5) acc.bw0 = acc.bw0 + selu(valid, bits, zero_u<V>()) * (1u << (5 * slot));
        else if (slot < 10) acc.bw1 = acc.bw1 + selu(valid, bits, zero_u<V>()) * (1u << (5 * (slot - 5)));
        else {
            // (the lane groups' records are ADDED up - contact_sweep: a per-slot odd multiplier keeps a change in one group's slot from
            //  cancelling against the opposite change in another group's, which a plain sum of the 5-bit fields would allow)
            const unsigned sk = (((unsigned)slot * 0x9E3779B1u) ^ (((unsigned)slot * 0x85EBCA6Bu) >> 13)) | 1u;
            acc.xh = acc.xh * 0x9E3779B1u + selu(valid, bits, zero_u<V>() + 7u) * sk;
        }
    }
    if (mode == 2) return;
    V f1 = sel(a1, V(1), V(0)), f2 = sel(a2, V(1), V(0)), f3 = sel(a3, V(1), V(0)), f4 = sel(a4, V(1), V(0));
    WeightedRows<V> Q;
    weighted_rows<V>(R, D, mu, f1, f2, f3, f4, Q);
    acc_root<V>(R, Q, acc);
    if (has_sh) {
#pragma unroll
        for (int ip = 0; ip < 3; ip++) acc.B[ip][0] = col_times_wb<V>(R, Q, 6, ip, acc.B[ip][0]);
        acc.C11 = col_times_col<V>(R, Q, 6, 6, acc.C11);
        acc.rl[0] = col_times_wa<V>(R, Q, 6, acc.rl[0]);
    }
    if (has_kn) {
#pragma unroll
        for (int ip = 0; ip < 3; ip++) acc.B[ip][1] = col_times_wb<V>(R, Q, 7, ip, acc.B[ip][1]);
        acc.C12 = col_times_col<V>(R, Q, 6, 7, acc.C12);
        acc.C22 = col_times_col<V>(R, Q, 7, 7, acc.C22);
        acc.rl[1] = col_times_wa<V>(R, Q, 7, acc.rl[1]);
    }
    if (is_pair) acc.X = col_times_col<V>(R, Q, 6, 7, acc.X);
    if (has_m) {
#pragma unroll
        for (int ip = 0; ip < 3; ip++) acc.Bm[ip] = col_times_wb<V>(R, Q, 7, ip, acc.Bm[ip]);
        acc.Cm = col_times_col<V>(R, Q, 7, 7, acc.Cm);
        acc.rm = col_times_wa<V>(R, Q, 7, acc.rm);
    }
}

// ----------------------------------------------------------------------------- spread sweeps
// A robot lying on a leg has 8-9 live slots in ONE leg: in a sweep of the ordinary kind the four lanes of that leg (one per lane group)
// work through two or three slots each, one per round, while the twelve lanes of the other legs have next to nothing to do.  When a
// wave's plan has more than one round, the LEG slots (0-9, cached rows) are swept in spread mode instead: inside every lane group the
// four lanes of an env share out the group's slots of ALL FOUR legs - a lane keeps the first contact of its own leg and a lane whose
// own leg has none adopts a further contact of another leg of its env.  What a lane accumulates for the root block is summed over the
// quad anyway; what it accumulates for a leg block (and the active-set record) is routed to the lane of that leg by a quad-level
// segmented sum.  The assignment is a function of the env's own contacts only, and an env that never has two contacts in one lane
// keeps the very bits of the ordinary sweep (every item stays in the lane and group it had, the routed sums add zeros): results remain
// independent of an env's wave-mates.
template <typename V> struct SpreadItem {
    typename lane_traits<V>::uint src, slot;      // leg (position in the quad) and slot of the contact this lane works on
    typename lane_traits<V>::mask valid;
    typename lane_traits<V>::mask more;           // the env has contacts left for a further round
};
// round r of the assignment.  own: this lane's live leg slots; mg: the slots (live, cached, of this lane's group) that are spread.
template <typename V>
JB_HD SpreadItem<V> spread_assign(const typename lane_traits<V>::uint& own, unsigned mg, int round) {
    using U = typename lane_traits<V>::uint;
    using MK = typename lane_traits<V>::mask;
    const U me = quad_lane_id((const V*)nullptr);
    const U Z = zero_u<V>();
    U mk[4], c[4], e[4];
    mk[0] = and_u(quad_bcast_u<0>(own), mg); mk[1] = and_u(quad_bcast_u<1>(own), mg); mk[2] = and_u(quad_bcast_u<2>(own), mg); mk[3] = and_u(quad_bcast_u<3>(own), mg);
    U F = Z, f = Z, T = Z;        // free lanes of the quad, free lanes below this one, contacts beyond a leg's first
#pragma unroll
    for (int j = 0; j < 4; j++) {
        c[j] = popc_u(mk[j]);
        const MK fr = eq_u(c[j], 0u);
        e[j] = selu(fr, Z, sub_u(c[j], Z + 1u));
        F = F + mbit(fr);
        f = f + mbit(mand(fr, lt_u(Z + (unsigned)j, me)));
        T = T + e[j];
    }
    const MK me0 = eq_u(me, 0u), me1 = eq_u(me, 1u), me2 = eq_u(me, 2u);
    const U c_me = selu(me0, c[0], selu(me1, c[1], selu(me2, c[2], c[3])));
    const MK own_first = mand(neq_u(c_me, Z), round == 0 ? lt(V(0), V(1)) : lt(V(1), V(0)));
    // the overflow contact this lane would take: round 0 - the free lanes in order; later rounds - every lane
    const U o = round == 0 ? f : F + (unsigned)(4 * (round - 1)) + me;
    const MK has_over = mand(lt_u(o, T), mnot(own_first));
    U src = me, k = Z, base = Z;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const MK take = mand(mand(has_over, mnot(lt_u(o, base))), lt_u(o, base + e[j]));
        src = selu(take, Z + (unsigned)j, src);
        k = selu(take, sub_u(o, base) + 1u, k);
        base = base + e[j];
    }
    const U msrc = selu(eq_u(src, 0u), mk[0], selu(eq_u(src, 1u), mk[1], selu(eq_u(src, 2u), mk[2], mk[3])));
    const U t1 = clear_low_u(msrc), t2 = clear_low_u(t1), t3 = clear_low_u(t2), t4 = clear_low_u(t3);
    const U tk = selu(eq_u(k, 0u), msrc, selu(eq_u(k, 1u), t1, selu(eq_u(k, 2u), t2, selu(eq_u(k, 3u), t3, t4))));
    SpreadItem<V> it;
    it.valid = mor(own_first, has_over);
    it.src = selu(it.valid, src, me);
    it.slot = ctz_u(tk);
    it.more = lt_u(F + (unsigned)(4 * round), T);
    return it;
}
// sum over the quad of the values whose tag names this lane: out[l] = sum_j [tag_j == l] t_j, associated ((own + next) + second next) + third
template <typename V> JB_HD V quad_seg_sum(const V& t, const V& m0, const V& m1, const V& m2, const V& m3) {
    V r = t * m0;
    r = r + quad_rot<1>(t) * m1;
    r = r + quad_rot<2>(t) * m2;
    return r + quad_rot<3>(t) * m3;
}
template <typename V> JB_HD typename lane_traits<V>::uint quad_seg_sum_u(const typename lane_traits<V>::uint& t, const typename lane_traits<V>::mask& k0, const typename lane_traits<V>::mask& k1,
                                                                         const typename lane_traits<V>::mask& k2, const typename lane_traits<V>::mask& k3) {
    const auto Z = zero_u<V>();
    return selu(k0, t, Z) + selu(k1, quad_rot_u<1>(t), Z) + selu(k2, quad_rot_u<2>(t), Z) + selu(k3, quad_rot_u<3>(t), Z);
}

// One cached contact of a LEG slot (0-9) of leg `it.src` of this lane's env against the iterate (spread sweeps).  The arithmetic of
// contact_apply for a lower-leg slot: an upper-leg slot's cached knee column is zero, so its extra terms add zeros.  yl: the joint part
// of the iterate of leg it.src.
template <typename V, bool LS = false>
JB_HD void contact_apply_leg(const RowVals<V>& rv, const Vec3<V> (&dk)[3], const V& mu, const SpreadItem<V>& it, int mode,
                             const Pk2<V> (&yr)[3], const V (&yl)[2], NewtonAcc<V>& acc, const LineDir<V>* ld = nullptr, const V* dls = nullptr) {
    using U = typename lane_traits<V>::uint;
    using MK = typename lane_traits<V>::mask;
    using W = Pk2<V>;
    V rho[3];
    const V D = sel(it.valid, rv.D, V(0));
    RowPairs<V> R;
    {
        // (a lane without a contact this round reads the first live slot's entry of its own leg: position and distance are there, the
        //  rest of a row is only built for contacts - selects, not a zero weight, keep whatever the entry holds out of the sums)
        V jsh[3], j7[3], ah[3];
#pragma unroll
        for (int k = 0; k < 3; k++) { jsh[k] = sel(it.valid, rv.jsh[k], V(0)); j7[k] = sel(it.valid, rv.j7[k], V(0)); ah[k] = sel(it.valid, rv.ah[k], V(0)); }
        row_pairs<V, false>(rv, dk, V(1), jsh, j7, ah, R);
    }
    row_residuals<V>(R, yr, W(yl[0], yl[1]), rho);
    const MK valid = gt(D, V(0));
    if (LS && mode == 3) {      // line search (see contact_apply); dls: the joint part of the direction for leg it.src
        const W dq[3] = {W(ld->dr[0], ld->dr[1]), W(ld->dr[2], ld->dr[3]), W(ld->dr[4], ld->dr[5])};
        V sg[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            W t = R.b[k][0] * dq[0];
            t = t + R.b[k][1] * dq[1]; t = t + R.b[k][2] * dq[2]; t = t + R.b[k][3] * W(dls[0], dls[1]);
            sg[k] = pk_lo(t) + pk_hi(t);
        }
        line_terms<V>(D, mu, rho, sg, ld->alpha, acc);
        return;
    }
    V mr1 = mu * rho[1], mr2 = mu * rho[2];
    auto a1 = lt(rho[0] + mr1, V(0)), a2 = lt(rho[0] - mr1, V(0)), a3 = lt(rho[0] + mr2, V(0)), a4 = lt(rho[0] - mr2, V(0));
    // whose leg is it?  tags of the quad's four lanes against this lane's position
    const U me = quad_lane_id((const V*)nullptr);
    const MK k0 = eq_u(it.src, me), k1 = eq_u(quad_rot_u<1>(it.src), me), k2 = eq_u(quad_rot_u<2>(it.src), me), k3 = eq_u(quad_rot_u<3>(it.src), me);
    {   // the record goes to the lane of the contact's leg
        const U bits = selu(valid, mbit(a1) + mbit(a2) * 2u + mbit(a3) * 4u + mbit(a4) * 8u + 16u, zero_u<V>());
        const MK low = lt_u(it.slot, 5u);
        const U sh = selu(low, it.slot, sub_u(it.slot, zero_u<V>() + 5u)) * 5u;
        const U w = shl_u(bits, sh);
        acc.bw0 = acc.bw0 + quad_seg_sum_u<V>(selu(low, w, zero_u<V>()), k0, k1, k2, k3);
        acc.bw1 = acc.bw1 + quad_seg_sum_u<V>(selu(low, zero_u<V>(), w), k0, k1, k2, k3);
    }
    if (mode == 2) return;
    V f1 = sel(a1, V(1), V(0)), f2 = sel(a2, V(1), V(0)), f3 = sel(a3, V(1), V(0)), f4 = sel(a4, V(1), V(0));
    WeightedRows<V> Q;
    weighted_rows<V>(R, D, mu, f1, f2, f3, f4, Q);
    acc_root<V>(R, Q, acc);
    // the leg block of leg it.src: computed here, added in the lane of that leg
    const V m0 = sel(k0, V(1), V(0)), m1 = sel(k1, V(1), V(0)), m2 = sel(k2, V(1), V(0)), m3 = sel(k3, V(1), V(0));
    const V Z = V(0);
    auto seg2 = [&](const W& t) { return W(quad_seg_sum(pk_lo(t), m0, m1, m2, m3), quad_seg_sum(pk_hi(t), m0, m1, m2, m3)); };
#pragma unroll
    for (int ip = 0; ip < 3; ip++) {
        acc.B[ip][0] = acc.B[ip][0] + seg2(col_times_wb<V>(R, Q, 6, ip, W(0)));
        acc.B[ip][1] = acc.B[ip][1] + seg2(col_times_wb<V>(R, Q, 7, ip, W(0)));
    }
    acc.C11 = acc.C11 + quad_seg_sum(col_times_col<V>(R, Q, 6, 6, Z), m0, m1, m2, m3);
    acc.C12 = acc.C12 + quad_seg_sum(col_times_col<V>(R, Q, 6, 7, Z), m0, m1, m2, m3);
    acc.C22 = acc.C22 + quad_seg_sum(col_times_col<V>(R, Q, 7, 7, Z), m0, m1, m2, m3);
    acc.rl[0] = acc.rl[0] + quad_seg_sum(col_times_wa<V>(R, Q, 6, Z), m0, m1, m2, m3);
    acc.rl[1] = acc.rl[1] + quad_seg_sum(col_times_wa<V>(R, Q, 7, Z), m0, m1, m2, m3);
}

// Rank-one Newton pass.  The active set at y differs from the factored one by ONE pyramid edge e of one cached contact of
// this env's legs (weight D, reference acceleration a): H' = H + s D e e^T, rhs' = rhs + s D a e with s = +1 (edge switched on) or
// -1 (off), hence by Sherman-Morrison
//     y' = y + z * s D (a - e.y) / (1 + s D e.z),      z = H^-1 e   (one substitution with the kept factorisation).
// Only the lane whose leg carries the contact has a non-zero e; the leg slots sit on the lower leg (0-4: columns 6 root dofs,
// shoulder, knee) or the upper leg (5-9: the cached knee column is zero).  entry / is_flip / plus / tan2 / on: flip_decode of
// this lane's records (cached row of the contact, which edge, switched on or off).
template <typename V, bool PAIR = false>
JB_HD void rank_one_pass(const LaneScratch<V>& sc, const StarFactor<V>& F, const Vec3<V> (&dk)[3], const V& mu,
                         const typename lane_traits<V>::uint& entry, const typename lane_traits<V>::mask& is_flip, const typename lane_traits<V>::mask& plus,
                         const typename lane_traits<V>::mask& tan2, const typename lane_traits<V>::mask& on,
                         const Pk2<V> (&yr)[3], const V (&yl)[2], const V& ym, Pk2<V> (&nyr)[3], V (&nyl)[2], V& nym) {
    using U = typename lane_traits<V>::uint;
    using W = Pk2<V>;
    const U e0 = entry * (unsigned)ROW_F + (unsigned)SC_ROWS;
    const V sg = sel(plus, mu, -mu);                    // e = B_n + sg * B_t
    V e[8];
    const Vec3<V> x = v3<V>(sc.ldv(e0), sc.ldv(e0 + 1u), sc.ldv(e0 + 2u));
    const Vec3<V> dt = v3<V>(dk[0].x + sg * sel(tan2, dk[2].x, dk[1].x), dk[0].y + sg * sel(tan2, dk[2].y, dk[1].y), dk[0].z + sg * sel(tan2, dk[2].z, dk[1].z));
    const Vec3<V> an = cross(x, dk[0]), at = sel_v3(tan2, cross(x, dk[2]), cross(x, dk[1]));
    e[0] = an.x + sg * at.x; e[1] = an.y + sg * at.y; e[2] = an.z + sg * at.z;
    e[3] = dt.x; e[4] = dt.y; e[5] = dt.z;
    e[6] = sc.ldv(e0 + 4u) + sg * sel(tan2, sc.ldv(e0 + 6u), sc.ldv(e0 + 5u));
    e[7] = sc.ldv(e0 + 7u) + sg * sel(tan2, sc.ldv(e0 + 9u), sc.ldv(e0 + 8u));
    const V ah = sc.ldv(e0 + 10u) + sg * sel(tan2, sc.ldv(e0 + 12u), sc.ldv(e0 + 11u));
    const V sD = sel(is_flip, sel(on, sc.ldv(e0 + 3u), -sc.ldv(e0 + 3u)), V(0));
#pragma unroll
    for (int i = 0; i < 8; i++) e[i] = sel(is_flip, e[i], V(0));
    W ep[3], zero3[3], zr[3];
    V zl[2], zm;
#pragma unroll
    for (int ip = 0; ip < 3; ip++) { ep[ip] = W(e[2 * ip], e[2 * ip + 1]); zero3[ip] = W(0); }
    star_subst<V, PAIR>(F, ep, zero3, e[6], e[7], V(0), zr, zl, zm);
    W ty = ep[0] * yr[0], tz = ep[0] * zr[0];
#pragma unroll
    for (int ip = 1; ip < 3; ip++) { ty = ty + ep[ip] * yr[ip]; tz = tz + ep[ip] * zr[ip]; }
    const V ety = (e[6] * yl[0] + e[7] * yl[1]) + (pk_lo(ty) + pk_hi(ty)), etz = (e[6] * zl[0] + e[7] * zl[1]) + (pk_lo(tz) + pk_hi(tz));
    // (selects, not products with a zero weight: a lane without the flip may hold anything in these temporaries)
    const V num = quad_sum(sel(is_flip, sD * (ah - ety), V(0))), den = V(1) + quad_sum(sel(is_flip, sD * etz, V(0)));
    const V c = num * vrcp(den);
#pragma unroll
    for (int ip = 0; ip < 3; ip++) nyr[ip] = yr[ip] + W(c) * zr[ip];
    nyl[0] = yl[0] + c * zl[0]; nyl[1] = yl[1] + c * zl[1]; nym = ym + c * zm;
}

// j-th set bit of a mask (j < popcount(mask))
JB_HD int nth_set_bit(unsigned mask, int j) {
    for (int k = 0; k < j; k++) mask &= mask - 1u;
    return __builtin_ctz(mask);
}

// Static slot -> helper-group map.  It depends on the slot index ONLY, so the order in which an env's contact terms are
// summed never depends on what the other envs of its wave are doing (results stay bit-identical for any batch split).
// The common multi-contact case - foot + the three lower-leg cylinder points (slots 0,1,3,4) - lands on four groups.
JB_HD constexpr int slot_group(int slot, int ngroups) {
    // Leg slots 0-9 are placed by hand from the live-slot statistics of real rollouts (tools/wave_stats.py): a leg lying on
    // the floor has foot, lower cylinder points 0/2/3, upper cylinder points 0/2/3 and the knee tip live, which this map
    // spreads as two per group (4 groups) or four per group (2 groups).
    constexpr int leg4[10] = {0, 1, 0, 2, 3, 1, 0, 2, 3, 0};
    constexpr int leg2[10] = {0, 1, 0, 0, 1, 1, 0, 0, 1, 0};
    return ngroups <= 1 ? 0 : slot < 10 ? (ngroups == 4 ? leg4[slot] : leg2[slot]) : ((slot - 10) & (ngroups - 1));
}
template <int G, int NG> struct GroupMask {
    static constexpr unsigned make() { unsigned mk = 0; for (int sl = 0; sl < NSLOT; sl++) if (slot_group(sl, NG) == G) mk |= 1u << sl; return mk; }
    static constexpr unsigned value = make();
};
// compile-time masks, selected by the (per-lane) group index
JB_HD unsigned group_mask(int g, int ngroups) {
    if (ngroups == 4) return g == 0 ? GroupMask<0, 4>::value : g == 1 ? GroupMask<1, 4>::value : g == 2 ? GroupMask<2, 4>::value : GroupMask<3, 4>::value;
    if (ngroups == 2) return g == 0 ? GroupMask<0, 2>::value : GroupMask<1, 2>::value;
    return (1u << NSLOT) - 1u;
}

// How the live slots of a substep are shared out: group g works through the live slots of ITS static subset, one per round.
// With a single group (or fewer than two live slots) only the main lanes work and nothing has to be reduced.
struct SlotPlan {
    unsigned live;       // wave-uniform bitmask of live slots
    unsigned mine;       // the live slots of this lane's group
    bool grouped;        // helper groups take part
    int ngroups, rounds;
};
template <typename V> JB_HD SlotPlan make_slot_plan(const LaneScratch<V>& sc, unsigned live_slots) {
    SlotPlan p;
    p.live = live_slots;
    const int count = __builtin_popcount(live_slots);
    p.grouped = sc.ngrp > 1 && count >= 2;
    p.ngroups = p.grouped ? sc.ngrp : 1;
    p.rounds = 0;
    for (int g = 0; g < p.ngroups; g++) {
        const int c = __builtin_popcount(live_slots & group_mask(g, p.ngroups));
        p.rounds = c > p.rounds ? c : p.rounds;
    }
    p.mine = live_slots & group_mask(p.grouped ? sc.grp : 0, p.ngroups);
    return p;
}
// the slot group g takes in round r (or -1), and the rank of a slot among the live ones (= its row-cache entry while below ROW_K)
JB_HD int plan_slot(const SlotPlan& p, int, int r) {
    return r < __builtin_popcount(p.mine) ? nth_set_bit(p.mine, r) : -1;
}
JB_HD int plan_rank(const SlotPlan& p, int slot) { return __builtin_popcount(p.live & ((1u << slot) - 1u)); }

// The spread part of a substep's plan (see "spread sweeps" above): which slots are spread, this lane's share, and what is left for the
// ordinary loop (body slots, a leg slot beyond the row cache).
template <typename V> struct SpreadPlan {
    bool on = false;
    unsigned slots = 0;          // wave-uniform: the live leg slots with a cached row
    unsigned mg = 0;             // ... of this lane's group
    typename lane_traits<V>::uint own;       // this lane's live leg slots
    SpreadItem<V> item0;         // round 0 of the assignment (later rounds are rare and recomputed)
    unsigned rest_mine = 0;      // the ordinary loop's slots of this lane's group, and its rounds
    int rest_rounds = 0;
};
template <typename V> JB_HD SpreadPlan<V> make_spread_plan(const LaneScratch<V>& sc, const SlotPlan& plan, bool enabled) {
    SpreadPlan<V> sp;
    if (!enabled || !plan.grouped || plan.rounds < 2) return sp;
    unsigned t = plan.live & 0x3FFu;
    while (__builtin_popcount(t) > ROW_K) t &= ~(1u << (31 - __builtin_clz(t)));      // (ranks follow the slot order: the leg slots come first)
    if (!t) return sp;
    sp.on = true;
    sp.slots = t;
    sp.mg = group_mask(sc.grp, plan.ngroups) & t;
    sp.own = vtou(sc.ld(SC_OWN));
    sp.item0 = spread_assign<V>(sp.own, sp.mg, 0);
    sp.rest_mine = plan.mine & ~t;
    for (int g = 0; g < plan.ngroups; g++) {
        const int c = __builtin_popcount(plan.live & ~t & group_mask(g, plan.ngroups));
        sp.rest_rounds = c > sp.rest_rounds ? c : sp.rest_rounds;
    }
    return sp;
}

// The rows of the spread slots, built the way they are swept: every lane builds the row of the contact it was dealt (round by round),
// reading the hinge data of that contact's leg from the leg's own LDS column and writing the row there.  Rows of (leg, slot) pairs
// without a contact are not built at all (nobody sums them: contact_apply_leg, rank_one_pass).  Same arithmetic as row_values for a
// leg slot (level 1 or 2 by the lane's slot).
template <typename V>
JB_HD void contact_rows_build_spread(const LaneModel<V>& m, const LaneScratch<V>& sc, const SlotPlan& plan, const SpreadPlan<V>& sp) {
    using U = typename lane_traits<V>::uint;
    using MK = typename lane_traits<V>::mask;
    const U me = quad_lane_id((const V*)nullptr);
    const Vec3<V> w = sc.ld3(SC_ST);
    const V mu = m.c[LM_MU];
#pragma unroll 1
    for (int r = 0;; r++) {
        const SpreadItem<V> it = r == 0 ? sp.item0 : spread_assign<V>(sp.own, sp.mg, r);
        const U below = sub_u(shl_u(zero_u<V>() + 1u, it.slot), zero_u<V>() + 1u);
        const U e0 = popc_u(and_u(below, zero_u<V>() + plan.live)) * (unsigned)ROW_F + (unsigned)SC_ROWS;
        const Vec3<V> x = v3<V>(ld_leg(sc.p, sc.stride, e0, it.src, me), ld_leg(sc.p, sc.stride, e0 + 1u, it.src, me), ld_leg(sc.p, sc.stride, e0 + 2u, it.src, me));
        const V dist = ld_leg(sc.p, sc.stride, e0 + 3u, it.src, me);
        auto ldl = [&](int i) { return ld_leg(sc.p, sc.stride, zero_u<V>() + (unsigned)i, it.src, me); };
        const V thd1 = ldl(SC_ST + 3), thd2 = ldl(SC_ST + 4);
        const Vec3<V> e1 = v3<V>(ldl(SC_DD + 9), ldl(SC_DD + 10), ldl(SC_DD + 11)), a1 = v3<V>(ldl(SC_DD + 12), ldl(SC_DD + 13), ldl(SC_DD + 14));
        const Vec3<V> e2 = v3<V>(ldl(SC_DD + 15), ldl(SC_DD + 16), ldl(SC_DD + 17)), a2 = v3<V>(ldl(SC_DD + 18), ldl(SC_DD + 19), ldl(SC_DD + 20));
        const MK lvl2 = lt_u(it.slot, 5u);
        const V f_sh = V(1), f_kn = sel(lvl2, V(1), V(0));
        const V tran = sel(lvl2, quad_pick(m.c.tran2, it.src), quad_pick(m.c.tran1, it.src));
        const MK valid = lt(dist, V(0));
        const V imp = impedance(m, dist);
        const V invD = (V(1) - imp) * (tran * ((V(1) + m.c[LM_FR2]) * (V(2) * mu * mu)));
        const V D = sel(valid, imp * vrcp(invD), V(0));
        const V jdot = f_kn * thd2;
        const Vec3<V> p1 = cross(e1, x - a1), p2 = cross(e2, x - a2);
        st_leg(sc.p, sc.stride, e0 + 3u, it.src, me, D, it.valid);
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const Vec3<V> d = sc.ld3(SC_DD + 3 * k);
            const Vec3<V> ang = cross(x, d);
            const V jsh = f_sh * dot(d, p1);
            const V j7 = f_kn * dot(d, p2);
            const V vel = dot(ang, w) + sc.ld(SC_DD + 21 + k) + jsh * thd1 + j7 * jdot;
            V ah = -m.c[LM_BB] * vel;
            if (k == 0) ah = ah - m.c[LM_KK] * imp * dist;
            st_leg(sc.p, sc.stride, e0 + (unsigned)(4 + k), it.src, me, jsh, it.valid);
            st_leg(sc.p, sc.stride, e0 + (unsigned)(7 + k), it.src, me, j7, it.valid);
            st_leg(sc.p, sc.stride, e0 + (unsigned)(10 + k), it.src, me, ah, it.valid);
        }
        if (!any_lane(it.more)) break;
    }
}

// y-independent rows of the live slots, once per substep (slots beyond the cache keep their candidate only; their rows are
// recomputed in registers in every pass)
template <typename V, bool PAIR = false>
JB_HD void contact_rows_build_all(const LaneModel<V>& m, const LaneScratch<V>& sc, bool xtra, const SlotPlan& plan, const SpreadPlan<V>& sp) {
    if (!plan.grouped && sc.grp != 0) return;
    unsigned mine = plan.mine;
    int rounds = plan.rounds;
    if (sp.on) { contact_rows_build_spread<V>(m, sc, plan, sp); mine = sp.rest_mine; rounds = sp.rest_rounds; }
#pragma unroll 1
    for (int r = 0; r < rounds; r++) {
        const int slot = r < __builtin_popcount(mine) ? nth_set_bit(mine, r) : -1;
        if (slot >= 0) {
            const int entry = plan_rank(plan, slot);
            if (entry < ROW_K) contact_rows_build<V, PAIR>(m, sc, xtra, slot, entry);
        }
    }
}

// every live candidate slot against the iterate y kept in the scratch (SC_Y); with helper groups the partial sums of the
// groups are combined by cross-lane exchanges so that every group ends with the complete accumulator
// (returns the number of spread rounds it made: diagnostics only)
template <typename V, bool PAIR = false, bool LS = false>
JB_HD int contact_sweep(const LaneModel<V>& m, const LaneScratch<V>& sc, bool xtra, const SlotPlan& plan, const SpreadPlan<V>& sp, int mode, const Vec3<V> (&dk)[3], NewtonAcc<V>& acc, const bool zero_g1 = false) {
    // zero_g1 (SimOpts::offload): group 1 must leave with an all-zero accumulator - it factorises M + h diag(b)
    // with the instruction stream that factorises the main lanes' Newton system (substep_impl)
    const bool g1z = zero_g1 && sc.grp == 1;
    int spread_rounds = 0;
    if (!plan.grouped && sc.grp != 0 && !g1z) return 0;
    if (mode == 2) { acc.bw0 = zero_u<V>(); acc.bw1 = zero_u<V>(); acc.xh = zero_u<V>(); }      // the check only records the active set
    else if (LS && mode == 3) { acc.ls_g = V(0); acc.ls_h = V(0); acc.ls_a = V(0); }                  // the line search only sums its scalars
    else acc_clear(acc);
    LineDir<V> ldir;
    const int box = ls_mailbox(sc);
    if (LS && mode == 3) {      // direction and trial step: left in the mailbox by the main lanes (newton_phase)
#pragma unroll
        for (int i = 0; i < 6; i++) ldir.dr[i] = sc.ld(box + i);
        ldir.dl[0] = sc.ld(box + 6); ldir.dl[1] = sc.ld(box + 7); ldir.dm = sc.ld(box + 8); ldir.alpha = sc.ld(box + 9);
    }
    const V mu = m.c[LM_MU];
    Pk2<V> yr[3];
    V yl[2], ym;
#pragma unroll
    for (int ip = 0; ip < 3; ip++) yr[ip] = Pk2<V>(sc.ld(SC_Y + 2 * ip), sc.ld(SC_Y + 2 * ip + 1));
    yl[0] = sc.ld(SC_Y + 6); yl[1] = sc.ld(SC_Y + 7); ym = sc.ld(SC_Y + 8);
    const int g = plan.grouped ? sc.grp : 0;
    unsigned rest_mine = plan.mine;
    int rest_rounds = plan.rounds;
    if (sp.on) {
        // spread rounds over the leg slots (normally one): every lane of the quad works on a contact of its env, whichever leg it sits on
        using U = typename lane_traits<V>::uint;
        const U me = quad_lane_id((const V*)nullptr);
#pragma unroll 1
        for (int r = 0;; r++) {
            const SpreadItem<V> it = r == 0 ? sp.item0 : spread_assign<V>(sp.own, sp.mg, r);
            const U below = sub_u(shl_u(zero_u<V>() + 1u, it.slot), zero_u<V>() + 1u);
            const U e0 = popc_u(and_u(below, zero_u<V>() + plan.live)) * (unsigned)ROW_F + (unsigned)SC_ROWS;      // the slot's cached row: entry = rank among the live slots
            RowVals<V> rv;
            rv.x = v3<V>(ld_leg(sc.p, sc.stride, e0, it.src, me), ld_leg(sc.p, sc.stride, e0 + 1u, it.src, me), ld_leg(sc.p, sc.stride, e0 + 2u, it.src, me));
            rv.D = ld_leg(sc.p, sc.stride, e0 + 3u, it.src, me);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                rv.jsh[k] = ld_leg(sc.p, sc.stride, e0 + (unsigned)(4 + k), it.src, me); rv.j7[k] = ld_leg(sc.p, sc.stride, e0 + (unsigned)(7 + k), it.src, me);
                rv.ah[k] = ld_leg(sc.p, sc.stride, e0 + (unsigned)(10 + k), it.src, me);
            }
            V yls[2];
            yls[0] = ld_leg(sc.p, sc.stride, zero_u<V>() + (unsigned)(SC_Y + 6), it.src, me); yls[1] = ld_leg(sc.p, sc.stride, zero_u<V>() + (unsigned)(SC_Y + 7), it.src, me);
            if (LS && mode == 3) {
                V dls[2];
                dls[0] = ld_leg(sc.p, sc.stride, zero_u<V>() + (unsigned)(box + 6), it.src, me); dls[1] = ld_leg(sc.p, sc.stride, zero_u<V>() + (unsigned)(box + 7), it.src, me);
                contact_apply_leg<V, LS>(rv, dk, mu, it, 3, yr, yls, acc, &ldir, dls);
            } else contact_apply_leg<V, LS>(rv, dk, mu, it, mode, yr, yls, acc);
            spread_rounds = r + 1;
            if (!any_lane(it.more)) break;
        }
        rest_mine = sp.rest_mine; rest_rounds = sp.rest_rounds;
    }
#pragma unroll 1
    for (int r = 0; r < rest_rounds; r++) {
        const int mine = r < __builtin_popcount(rest_mine) ? nth_set_bit(rest_mine, r) : -1;
        const bool lane_on = mine >= 0 && !(g1z && !plan.grouped);       // (no helper groups this substep: group 1 only rides along for its zeros)
        const int slot = lane_on ? mine : __builtin_ctz(sp.on ? (plan.live & ~sp.slots) : plan.live);       // idle lanes read some BUILT entry (spread mode: one of this loop's) and contribute nothing
        const int rank = plan_rank(plan, slot);
        RowVals<V> rv;
        if (rank < ROW_K) row_load<V>(sc, rank, rv);
        else {      // beyond the cache: the candidate -> row values, every pass
            const int c0 = 4 * (rank - ROW_K) * sc.ovc_stride;
            row_values<V, PAIR>(m, sc, xtra, slot, v3<V>(sc.ovc[c0], sc.ovc[c0 + sc.ovc_stride], sc.ovc[c0 + 2 * sc.ovc_stride]), sc.ovc[c0 + 3 * sc.ovc_stride], rv);
        }
        contact_apply<V, PAIR, LS>(rv, dk, mu, slot, lane_on, mode, yr, yl, ym, acc, (LS && mode == 3) ? &ldir : nullptr);
    }
    if (LS && mode == 3) {      // the env's totals on every lane: over the legs, then over the lane groups
        acc.ls_g = quad_sum(acc.ls_g); acc.ls_h = quad_sum(acc.ls_h); acc.ls_a = quad_sum(acc.ls_a);
        if (plan.grouped) { acc.ls_g = group_sum(sc, acc.ls_g); acc.ls_h = group_sum(sc, acc.ls_h); acc.ls_a = group_sum(sc, acc.ls_a); }
        return spread_rounds + 8 * rest_rounds;
    }
    if (plan.grouped) {
        acc.bw0 = group_sum_u<V>(sc, acc.bw0); acc.bw1 = group_sum_u<V>(sc, acc.bw1); acc.xh = group_sum_u<V>(sc, acc.xh);
        if (mode == 0 && sc.ngrp == 4 && sc.gstride == 16 && sc.red_lds) {
            // Only the main lanes need the totals: reduce four values at a time so that row (= group) g ends with the total
            // of value 4k+g, hand the totals over through the scratch (the overflow row entries are dead here) and let the
            // main lanes read all of them.  Same association as group_sum, a quarter of its instructions.
            constexpr int NQ4 = PAIR ? 14 : 13;       // quadruples of values (PAIR: X rides as value 52, padded with zeros)
            V v[56];
            acc_pack(acc, v);
            if (PAIR) { v[52] = acc.X; v[53] = V(0); v[54] = V(0); v[55] = V(0); }
            static_assert(4 * NQ4 <= 56, "reduction buffer");
#pragma unroll
            for (int k = 0; k < NQ4; k++) sc.st(SC_RED + 4 * k + sc.grp, row_transpose_sum(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]));
            wave_sync();          // every group's totals are in the scratch
            if (sc.grp == 0 || g1z) {
                const int src = g1z ? SC_ZERO : SC_RED;        // per lane: the same loads, another base address
#pragma unroll
                for (int i = 0; i < (PAIR ? 53 : 52); i++) v[i] = sc.ld(src + i);
                acc_unpack(v, acc);
                if (PAIR) acc.X = v[52];
            }
        } else if (mode == 0) {
            V v[52];
            acc_pack(acc, v);
#pragma unroll
            for (int i = 0; i < 52; i++) v[i] = group_sum(sc, v[i]);
            acc_unpack(v, acc);
            if (PAIR) acc.X = group_sum(sc, acc.X);
            if (g1z) acc_clear(acc);
        }
    }
    return spread_rounds + 8 * rest_rounds;
}

// Cylinder vs floor, restating MuJoCo's plane-cylinder routine (mjc_PlaneCylinder in MuJoCo's engine_collision_primitive.c - third
// party, not under /root/reference; the oracle's collide() restates the same routine and cites it likewise): up to 4 points.  c: centre, ax: unit axis,
// xa: geom x axis (degenerate case), all in root coordinates relative to the root origin; nb: floor normal in
// root coordinates; pz: world height of the root origin.  Outputs positions (relative to the root origin) and
// distances; a point is a contact when its mask is set.
template <typename V> struct CylContacts {
    Vec3<V> x[4];
    V dist[4];
    typename lane_traits<V>::mask on[4];
};
template <typename V>
JB_HD void cylinder_floor(const Vec3<V>& c, const Vec3<V>& ax_in, const Vec3<V>& xa, const V& rad, const V& half, const Vec3<V>& nb, const V& pz, const V& pz_lo,
                          const typename lane_traits<V>::mask& enabled, CylContacts<V>& out) {
    V prj = dot(ax_in, nb);
    auto flip = gt(prj, V(0));
    Vec3<V> ax = v3<V>(sel(flip, -ax_in.x, ax_in.x), sel(flip, -ax_in.y, ax_in.y), sel(flip, -ax_in.z, ax_in.z));
    prj = sel(flip, -prj, prj);
    V dist0 = (pz + dot(c, nb)) + pz_lo;
    Vec3<V> vec = ax * prj - nb;
    V len2 = dot(vec, vec);
    auto degenerate = lt(len2, V(1e-20));
    V scl = rad * vrsqrt(vmax(len2, V(1e-30)));
    vec = v3<V>(sel(degenerate, xa.x * rad, vec.x * scl), sel(degenerate, xa.y * rad, vec.y * scl), sel(degenerate, xa.z * rad, vec.z * scl));
    V prjvec = dot(vec, nb);
    Vec3<V> axh = ax * half;
    V prjaxis = prj * half;
    V d1 = dist0 + prjaxis + prjvec;
    auto on1 = mand(enabled, lt(d1, V(0)));
    if (!any_lane(on1)) {
        // the first point is the cylinder's lowest: when no lane of the wave has it below the floor there is no contact at all, and the
        // other points' positions (a cross product, a reciprocal square root, four position sums) need not be worked out - the usual case
#pragma unroll
        for (int k = 0; k < 4; k++) { out.dist[k] = V(1); out.on[k] = on1; out.x[k] = c; }
        return;
    }
    out.dist[0] = d1; out.on[0] = on1;
    out.x[0] = c + vec + axh - nb * (d1 * V(0.5));
    V d2 = dist0 - prjaxis + prjvec;
    out.dist[1] = d2; out.on[1] = mand(on1, lt(d2, V(0)));
    out.x[1] = c + vec - axh - nb * (d2 * V(0.5));
    V d3 = dist0 + prjaxis - V(0.5) * prjvec;
    auto on3 = mand(on1, lt(d3, V(0)));
    Vec3<V> v1 = cross(vec, axh);
    v1 = v1 * (rad * V(0.8660254037844386) * vrsqrt(vmax(dot(v1, v1), V(1e-30))));
    Vec3<V> base = c + axh - vec * V(0.5) - nb * (d3 * V(0.5));
    out.dist[2] = d3; out.on[2] = on3; out.x[2] = base + v1;
    out.dist[3] = d3; out.on[3] = on3; out.x[3] = base - v1;
}


// store a candidate: position and effective distance (+1 when it is not a contact)
// Returns the wave-uniform bit "some lane of the wave has a contact in this slot" (shifted to the slot's position):
// the sweeps of the substep visit only slots whose bit is set.
// `live_before`: the live bits of the slots BELOW this one (candidates are generated in increasing slot order), so that the slot's
// entry - its rank among the live slots - is known as it is stored: a slot nobody touches takes no entry at all.
template <typename V, typename MKT>
JB_HD unsigned cand_store(const LaneScratch<V>& sc, unsigned live_before, int slot, const Vec3<V>& x, const V& dist, const MKT& on, typename lane_traits<V>::uint* own = nullptr) {
    if (!any_lane(on)) return 0u;
    if (own) *own = *own + selu(on, zero_u<V>() + (1u << slot), zero_u<V>());      // leg slots: which of them are contacts of THIS lane's leg
    const int rank = __builtin_popcount(live_before & ((1u << slot) - 1u));
    const V d = sel(on, dist, V(1));
    if (sc.aux_lane) return 1u << slot;          // (an aux lane's masks are never set; it mirrors a leg lane's addresses and must not write there)
    if (rank < ROW_K) {
        const int e0 = SC_ROWS + ROW_F * rank;
        sc.st3(e0, x); sc.st(e0 + 3, d);
    } else {
        const int c0 = 4 * (rank - ROW_K) * sc.ovc_stride;
        sc.ovc[c0] = x.x; sc.ovc[c0 + sc.ovc_stride] = x.y; sc.ovc[c0 + 2 * sc.ovc_stride] = x.z; sc.ovc[c0 + 3 * sc.ovc_stride] = d;
    }
    return 1u << slot;
}
template <typename V, typename MKT>
JB_HD unsigned cand_store_cyl(const LaneScratch<V>& sc, unsigned live_before, int slot0, const CylContacts<V>& c, const MKT& gate, typename lane_traits<V>::uint* own = nullptr) {
    unsigned live = live_before;
#pragma unroll
    for (int k = 0; k < 4; k++) live |= cand_store(sc, live, slot0 + k, c.x[k], c.dist[k], mand(c.on[k], gate), own);
    return live & ~live_before;
}

// ----------------------------------------------------------------------------- the geom-geom pair: mass ellipsoid against the own upper-leg cylinder
// The narrow phase both the oracle (oracle/jb_oracle.c pair_geometric, where it is also held against a restatement of MuJoCo's MPR)
// and this kernel use: x* = the point of the cylinder's axis segment with the smallest SIGNED distance to the ellipsoid, q* its nearest
// ellipsoid point; normal = the ellipsoid's outward normal at q* (from the mass to the leg), distance = sd - r_cyl, position = the
// middle of the overlap.  In the ellipsoid's own axes the nearest point of y is q_i = s_i^2 y_i / (s_i^2 + lam) with lam the root of
//     F(lam) = sum_i s_i^2 y_i^2 / (s_i^2 + lam)^2 - 1        (lam > 0 outside, < 0 inside; F convex and decreasing: Newton),
// and then y - q = lam g with g_i = y_i / (s_i^2 + lam) (half the gradient at q): signed distance = lam |g|, normal = g / |g|.  Along the
// axis the signed distance is convex, so its derivative f(t) = u . normal is monotone; its root lies within the largest semi-axis of
// the ellipsoid centre's own projection on the axis - a bracket from which an Illinois iteration with fixed counts converges.
// Everything is branch-free with fixed iteration counts: all lanes of a wave walk through it together.
template <typename V> JB_HD void ell_lambda(const Vec3<V>& s2, const V& lam_min, const Vec3<V>& y, V& lam, int iters) {
    // Newton on 1 / N(lam) - 1 with N^2 = F + 1 (the "secular equation" form: 1 / N is nearly linear in lam, all the way to the pole at
    // -min s^2, so the iteration neither crawls away from the pole nor overshoots far past the root as Newton on F itself does for
    // points inside the ellipsoid):  lam <- lam + (N - 1) N^2 / sum_i p_i / (s_i^2 + lam)^3.
    const V px = s2.x * y.x * y.x, py = s2.y * y.y * y.y, pz = s2.z * y.z * y.z;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        const V ix = vrcp(s2.x + lam), iy = vrcp(s2.y + lam), iz = vrcp(s2.z + lam);
        const V tx = px * ix * ix, ty = py * iy * iy, tz = pz * iz * iz;
        const V N2 = vmax(tx + ty + tz, V(1e-30)), S3 = vmax(tx * ix + ty * iy + tz * iz, V(1e-30));
        const V N = N2 * vrsqrt(N2);
        lam = vmax(lam + (N - V(1)) * N2 * vrcp(S3), lam_min);
    }
}
template <typename V>
JB_HD void pair_narrow(const Vec3<V>& ce, const Mat3<V>& Re, const Vec3<V>& sz, const Vec3<V>& cc, const Vec3<V>& ua, const V& rad, const V& half,
                       V& dist, Vec3<V>& n, Vec3<V>& pos, V* t_out = nullptr, V* lam_out = nullptr) {
    const Vec3<V> cl = mulT(Re, cc - ce), ul = mulT(Re, ua);
    const Vec3<V> s2 = v3<V>(sz.x * sz.x, sz.y * sz.y, sz.z * sz.z);
    const V lam_min = V(-0.95) * vmin(s2.x, vmin(s2.y, s2.z)), smax = vmax(sz.x, vmax(sz.y, sz.z));
    V lam = V(0);
    Vec3<V> x, g;
    auto eval = [&](const V& t, int iters) -> V {
        x = cl + ul * t;
        ell_lambda<V>(s2, lam_min, x, lam, iters);
        g = v3<V>(x.x * vrcp(s2.x + lam), x.y * vrcp(s2.y + lam), x.z * vrcp(s2.z + lam));
        return dot(g, ul) * vrsqrt(dot(g, g));
    };
    const V t0 = -dot(cl, ul);
    const V a0 = vmax(t0 - smax, -half), b0 = vmin(vmax(t0 + smax, -half), half);
    V ta = a0, tb = vmax(b0, a0);
    V fa = eval(ta, 8);
    V fb = eval(tb, 6);
    const auto at_a = mnot(lt(fa, V(0))), at_b = mnot(gt(fb, V(0)));      // the minimiser sits on an end of the bracket (an end of the leg)
    V tc = tb;
#pragma unroll 1
    for (int it = 0; it < 5; it++) {
        const V den = fb - fa;
        const auto ok = gt(vabs(den), V(1e-20));
        tc = sel(ok, tb - fb * (tb - ta) * vrcp(sel(ok, den, V(1))), tb);
        tc = vmin(vmax(tc, vmin(ta, tb)), vmax(ta, tb));
        const V fc = eval(tc, 4);
        // Illinois: keep a bracket; when the same end survives twice, halve its function value
        const auto opposite = lt(fc * fb, V(0));
        ta = sel(opposite, tb, ta); fa = sel(opposite, fb, fa * V(0.5));
        tb = tc; fb = fc;
    }
    // Newton on f inside the bracket the Illinois steps leave (quadratic: the normal is wanted to round-off, and it is first order in
    // the error of t).  With x = q + lam g, g = S^-2 q on the ellipsoid:  q' = A (u - lam' g), A = diag(s^2 / (s^2 + lam)),
    // lam' = (g.A u) / (g.A g),  g' = S^-2 q',  n' = (g' - n (n.g')) / |g|,  f' = u.n'.
    const V lo_t = vmin(ta, tb), hi_t = vmax(ta, tb);
    V fc = fb;
#pragma unroll 1
    for (int it = 0; it < 3; it++) {
        const Vec3<V> A = v3<V>(s2.x * vrcp(s2.x + lam), s2.y * vrcp(s2.y + lam), s2.z * vrcp(s2.z + lam));
        const Vec3<V> Au = v3<V>(A.x * ul.x, A.y * ul.y, A.z * ul.z), Ag = v3<V>(A.x * g.x, A.y * g.y, A.z * g.z);
        const V dlam = dot(g, Au) * vrcp(dot(g, Ag));
        const Vec3<V> dq = Au - Ag * dlam;
        const Vec3<V> dg = v3<V>(dq.x * vrcp(s2.x), dq.y * vrcp(s2.y), dq.z * vrcp(s2.z));
        const V gg_ = dot(g, g), ig_ = vrsqrt(gg_);
        const V ndg = dot(g, dg) * ig_;                                   // n . g'
        const V df = (dot(ul, dg) - (dot(g, ul) * ig_) * ndg) * ig_;      // u . n'
        const auto okd = gt(df, V(1e-12));
        tc = vmin(vmax(tc - fc * vrcp(sel(okd, df, V(1))) * sel(okd, V(1), V(0)), lo_t), hi_t);
        fc = eval(tc, 4);
    }
    tc = sel(at_a, a0, sel(at_b, vmax(b0, a0), tc));
    (void)eval(tc, 5);
    const V gg = dot(g, g), ig = vrsqrt(gg);
    const Vec3<V> nl = g * ig;
    dist = lam * gg * ig - rad;                                   // signed distance of the axis point = lam |g|
    const Vec3<V> pl = x - g * (V(0.5) * lam) - nl * (V(0.5) * rad);    // (q + x - r n) / 2 with q = x - lam g
    n = mul(Re, nl);
    pos = ce + mul(Re, pl);
    if (t_out) { *t_out = tc; *lam_out = lam; }
}
// The same contact from the PREVIOUS substep's solution (axis parameter t, multiplier lam): between two substeps the pair moves by a
// fraction of a millimetre, so Newton on f(t) started there converges quadratically - 3 steps of 3 multiplier iterations instead of the
// cold scheme's 51.  Returns, per lane, whether it did converge (|f| within 64 ulp of zero, or pinned at an end of the leg with f
// pointing outwards, and the multiplier's own equation satisfied); a lane that did not takes the cold scheme's result instead - its own
// data decide, never its wave-mates'.  The first substep of a control step always runs the cold scheme (the warm state is not part of
// the simulator's state: K single steps and one K-step launch must agree bit for bit).  The oracle keeps the cold scheme: both converge
// to the same contact to round-off.
template <typename V>
JB_HD typename lane_traits<V>::mask pair_narrow_warm(const Vec3<V>& ce, const Mat3<V>& Re, const Vec3<V>& sz, const Vec3<V>& cc, const Vec3<V>& ua, const V& rad, const V& half,
                                                      V& t, V& lam, V& dist, Vec3<V>& n, Vec3<V>& pos) {
    using R = typename lane_traits<V>::real;
    const Vec3<V> cl = mulT(Re, cc - ce), ul = mulT(Re, ua);
    const Vec3<V> s2 = v3<V>(sz.x * sz.x, sz.y * sz.y, sz.z * sz.z);
    const V lam_min = V(-0.95) * vmin(s2.x, vmin(s2.y, s2.z)), smax = vmax(sz.x, vmax(sz.y, sz.z));
    Vec3<V> x, g;
    auto eval = [&](const V& tt, int iters) -> V {
        x = cl + ul * tt;
        ell_lambda<V>(s2, lam_min, x, lam, iters);
        g = v3<V>(x.x * vrcp(s2.x + lam), x.y * vrcp(s2.y + lam), x.z * vrcp(s2.z + lam));
        return dot(g, ul) * vrsqrt(dot(g, g));
    };
    const V t0 = -dot(cl, ul);
    const V a0 = vmax(t0 - smax, -half), b0 = vmax(vmin(vmax(t0 + smax, -half), half), a0);
    lam = vmax(lam, lam_min);
    V tc = vmin(vmax(t, a0), b0);
    V fc = eval(tc, 3);
#pragma unroll 1
    for (int it = 0; it < 3; it++) {
        const Vec3<V> A = v3<V>(s2.x * vrcp(s2.x + lam), s2.y * vrcp(s2.y + lam), s2.z * vrcp(s2.z + lam));
        const Vec3<V> Au = v3<V>(A.x * ul.x, A.y * ul.y, A.z * ul.z), Ag = v3<V>(A.x * g.x, A.y * g.y, A.z * g.z);
        const V dlam = dot(g, Au) * vrcp(dot(g, Ag));
        const Vec3<V> dq = Au - Ag * dlam;
        const Vec3<V> dg = v3<V>(dq.x * vrcp(s2.x), dq.y * vrcp(s2.y), dq.z * vrcp(s2.z));
        const V gg_ = dot(g, g), ig_ = vrsqrt(gg_);
        const V ndg = dot(g, dg) * ig_;
        const V df = (dot(ul, dg) - (dot(g, ul) * ig_) * ndg) * ig_;
        const auto okd = gt(df, V(1e-12));
        tc = vmin(vmax(tc - fc * vrcp(sel(okd, df, V(1))) * sel(okd, V(1), V(0)), a0), b0);
        fc = eval(tc, 3);
    }
    // converged?  f at the solution, and the multiplier's equation N(lam) = 1 at that point
    const V tol = V(R(64) * (sizeof(R) == 4 ? R(1.1920929e-07) : R(2.220446049250313e-16)));
    const V ix = vrcp(s2.x + lam), iy = vrcp(s2.y + lam), iz = vrcp(s2.z + lam);
    const V N2 = s2.x * x.x * x.x * ix * ix + s2.y * x.y * x.y * iy * iy + s2.z * x.z * x.z * iz * iz;
    const auto lam_ok = lt(vabs(N2 - V(1)), V(4) * tol);
    const auto at_lo = mand(mnot(gt(tc, a0)), mnot(lt(fc, V(0)))), at_hi = mand(mnot(lt(tc, b0)), mnot(gt(fc, V(0))));
    const auto conv = mand(lam_ok, mor(lt(vabs(fc), tol), mor(at_lo, at_hi)));
    const V gg = dot(g, g), ig = vrsqrt(gg);
    const Vec3<V> nl = g * ig;
    dist = lam * gg * ig - rad;
    const Vec3<V> pl = x - g * (V(0.5) * lam) - nl * (V(0.5) * rad);
    n = mul(Re, nl);
    pos = ce + mul(Re, pl);
    t = tc;
    return conv;
}

// ----------------------------------------------------------------------------- the second geom-geom pair: motor-axis thread against the own upper-leg cylinder
// (oracle/jb_oracle.c pair_thread_geometric states the definition: the point of the LEG's axis segment with the smallest signed distance to
// the thread cylinder, the thread's nearest surface point and outward normal there.)  Signed distance of a point to a finite cylinder with
// flat caps (centre tc, unit axis ta, radius R, half length H), branch-free: q = nearest surface point, nrm = outward normal.
template <typename V>
JB_HD V cyl_nearest(const Vec3<V>& tc, const Vec3<V>& ta, const V& R, const V& H, const Vec3<V>& x, Vec3<V>& q, Vec3<V>& nrm) {
    const Vec3<V> y = x - tc;
    const V z = dot(y, ta);
    const Vec3<V> rv = y - ta * z;
    const V rho2 = dot(rv, rv), irho = vrsqrt(vmax(rho2, V(1e-30))), rho = rho2 * irho;
    const Vec3<V> er = rv * irho;                                    // (on the axis itself: a zero vector - the side is never the nearest surface there unless R < H... see below)
    const V dr = rho - R, az = vabs(z), dz = az - H, sz = sel(lt(z, V(0)), V(-1), V(1));
    const auto rim = mand(gt(dr, V(0)), gt(dz, V(0))), side = mand(mnot(rim), gt(dr, dz));
    const V d2 = dr * dr + dz * dz, id = vrsqrt(vmax(d2, V(1e-30))), drim = d2 * id;
    const Vec3<V> q_side = tc + er * R + ta * z, q_rim = tc + er * R + ta * (sz * H), q_cap = tc + rv + ta * (sz * H);
    const Vec3<V> n_rim = er * (dr * id) + ta * (sz * dz * id), n_cap = ta * sz;
    q = sel_v3(rim, q_rim, sel_v3(side, q_side, q_cap));
    nrm = sel_v3(rim, n_rim, sel_v3(side, er, n_cap));
    return sel(rim, drim, sel(side, dr, dz));
}
// smallest distance between the segments c1 + s u1 (|s| <= h1) and c2 + t u2 (|t| <= h2), unit directions (Ericson 5.1.9, with selects)
template <typename V>
JB_HD V segment_distance(const Vec3<V>& c1, const Vec3<V>& u1, const V& h1, const Vec3<V>& c2, const Vec3<V>& u2, const V& h2) {
    const Vec3<V> r = c1 - c2;
    const V b = dot(u1, u2), c = dot(u1, r), f = dot(u2, r), den = V(1) - b * b;
    V s = sel(gt(den, V(1e-12)), (b * f - c) * vrcp(vmax(den, V(1e-12))), V(0));
    s = vmin(vmax(s, -h1), h1);
    V t = b * s + f;
    const auto lo = lt(t, -h2), hi = gt(t, h2);
    const V tcl = vmin(vmax(t, -h2), h2);
    const V s2 = vmin(vmax(b * tcl - c, -h1), h1);
    s = sel(mor(lo, hi), s2, s);
    t = tcl;
    const Vec3<V> dd = r + u1 * s - u2 * t;
    const V d2 = dot(dd, dd);
    return d2 * vrsqrt(vmax(d2, V(1e-30)));
}
#ifndef JB_THREAD_BISECT
#define JB_THREAD_BISECT 26
#endif
// the contact: dist (< 0: overlap), mo = the thread's outward normal at the contact (thread -> leg), pos; all in root coordinates
template <typename V>
JB_HD void thread_narrow(const Vec3<V>& tc, const Vec3<V>& ta, const V& R, const V& H, const Vec3<V>& cc, const Vec3<V>& ua, const V& rad, const V& half,
                         V& dist, Vec3<V>& mo, Vec3<V>& pos) {
    Vec3<V> q, nrm;
    V ta_ = -half, tb_ = half;
    (void)cyl_nearest<V>(tc, ta, R, H, cc + ua * ta_, q, nrm);
    const auto at_a = mnot(lt(dot(nrm, ua), V(0)));          // the signed distance already grows at the lower end: the minimiser is that end
    (void)cyl_nearest<V>(tc, ta, R, H, cc + ua * tb_, q, nrm);
    const auto at_b = mnot(gt(dot(nrm, ua), V(0)));
#pragma unroll 1
    for (int it = 0; it < JB_THREAD_BISECT; it++) {           // f(t) = m(x(t)) . u is monotone (the signed distance to a convex body is convex along a line)
        const V t = V(0.5) * (ta_ + tb_);
        (void)cyl_nearest<V>(tc, ta, R, H, cc + ua * t, q, nrm);
        const auto neg = lt(dot(nrm, ua), V(0));
        ta_ = sel(neg, t, ta_); tb_ = sel(neg, tb_, t);
    }
    const V t = sel(at_a, -half, sel(at_b, half, V(0.5) * (ta_ + tb_)));
    const Vec3<V> x = cc + ua * t;
    const V sd = cyl_nearest<V>(tc, ta, R, H, x, q, nrm);
    dist = sd - rad;
    mo = nrm;
    pos = (q + x - nrm * rad) * V(0.5);
}

#if !defined(__HIPCC__)
inline long g_pair_narrow_stats[2] = {0, 0};      // host harness only: substeps whose narrow phase was entirely warm-started / needed the cold scheme
#endif

// ----------------------------------------------------------------------------- options
struct SimOpts {
    int contacts;        // 0: contacts disabled (MuJoCo disableflags=contact)
    int max_newton;      // cap on Newton iterations per substep
    int implicit_damp;   // 1: MuJoCo Euler implicit joint damping
    int rank_one;        // 1: single-edge changes of the active set are rank-one passes (0: diagnostic, always full passes)
    int lean = 0;        // 1: LEAN kernel variant - the lane state, the joint-space system and the kept factorisation live in the scratch
                         //    between the phases that use them (register budget of two waves per SIMD); same arithmetic, same results
    int offload = 0;     // 1 (one-wave-per-SIMD kernels with helper groups): lane group 1 is a full REPLICA of the main lanes - it loads the same state
                         //    and runs phase A and phase C with them, instruction for instruction, so that it holds the joint-space system too.
                         //    While the main lanes factorise the first Newton system of a substep, the replica factorises M + h diag(b) with the
                         //    very same instructions (zero accumulator, hb on its diagonal); the final pass is then a substitution in the replica
                         //    and a 9-value hand-over instead of a second factorisation on the critical path.  Bit-identical results.
    int aux = 0;         // 1 (offload kernels with four lane groups, shared model): lane groups 2 and 3 - idle in phase A otherwise - hold the env's root state too and run
                         //    phase A / C WITH the main lanes and their replica, instruction for instruction, on two "legs" of their own: the motor body
                         //    (lane 0 of their quad) and the root body's own mass (lane 1); build_aux_block.  What the four leg lanes (and the replica) used to
                         //    repeat - motor kinematics, composite inertia and bias, root bias: ~9 % of a substep's instructions - is then computed ONCE, by the
                         //    very instructions that compute the legs, and handed over with 23 cross-lane swaps (group 0 <- 2, group 1 <- 3).
    int spread = 1;      // 1: a plan with more than one round sweeps the leg slots in spread mode (lanes of idle legs adopt contacts of a leg that has
                         //    several: "spread sweeps" below); 0: diagnostic, the ordinary sweep only
    float* capture = nullptr;           // diagnostic builds (-DJB_CAPTURE): ring of JB_CAPTURE_SLOTS records x 64 floats - the entry state of substeps whose contact solve stayed
    unsigned* capture_count = nullptr;  //   unconverged after the line-searched pass (substep_impl), and how many there were
    unsigned long long* prof;   // diagnostic builds: per-wave cycle accumulators [phaseA, check sweeps, full sweeps, solves, integrate]
    unsigned long long* hist;   // diagnostic builds: per-wave [0..27] live-slot counts over all-geom substeps, [28..35] rounds histogram, [36..43] same for ordinary substeps
};

// small-angle sin/cos for the leg hinges (|th| < 1: truncation < 1e-9), exact libm otherwise
template <typename V> JB_HD void sincos_small(const V& x, V& s, V& c) {
    V x2 = x * x;
    s = x * (V(1) + x2 * (V(-1.0 / 6) + x2 * (V(1.0 / 120) + x2 * (V(-1.0 / 5040) + x2 * (V(1.0 / 362880) + x2 * V(-1.0 / 39916800))))));
    c = V(1) + x2 * (V(-0.5) + x2 * (V(1.0 / 24) + x2 * (V(-1.0 / 720) + x2 * (V(1.0 / 40320) + x2 * (V(-1.0 / 3628800) + x2 * V(1.0 / 479001600))))));
}

#ifdef JB_WAVE_STATS
#if defined(__HIPCC__)
template <typename V> __device__ inline void stats_hist(const SimOpts& o, const LaneScratch<V>&, const SlotPlan& plan, bool xtra, bool any_contact) {
    // o.hist is THIS wave's private row of 64 counters (no atomics: they would serialise the waves when many are on the rare path)
    if (!o.hist || !any_contact || (threadIdx.x & 63) != 0) return;
    if (xtra) for (int sl = 0; sl < 28; sl++) if (plan.live >> sl & 1u) o.hist[sl] += 1ull;
    o.hist[(xtra ? 28 : 36) + (plan.rounds < 7 ? plan.rounds : 7)] += 1ull;
}
#else
template <typename V> inline void stats_hist(const SimOpts&, const LaneScratch<V>&, const SlotPlan&, bool, bool) {}
#endif
#endif

// ---- LEAN variant: park / fetch long-lived values in the lane's scratch (lane-private columns: no hand-over between lanes)
// (the root block [[J, [h]x], [[h]x^T, m 1]] is parked as its 10 distinct values: J (6), h (3), m)
template <typename V> JB_HD void sys_store(const LaneScratch<V>& sc, const StarSys<V>& y) {
    sc.st(SC_SYS + 0, s6get(y.A, 0, 0)); sc.st(SC_SYS + 1, s6get(y.A, 1, 0)); sc.st(SC_SYS + 2, s6get(y.A, 1, 1));
    sc.st(SC_SYS + 3, s6get(y.A, 2, 0)); sc.st(SC_SYS + 4, s6get(y.A, 2, 1)); sc.st(SC_SYS + 5, s6get(y.A, 2, 2));
    sc.st(SC_SYS + 6, s6get(y.A, 4, 2)) /*hx*/; sc.st(SC_SYS + 7, s6get(y.A, 5, 0)) /*hy*/; sc.st(SC_SYS + 8, s6get(y.A, 3, 1)) /*hz*/; sc.st(SC_SYS + 9, s6get(y.A, 3, 3)) /*m*/;
#pragma unroll
    for (int i = 0; i < 6; i++) { sc.st(SC_SYS + 10 + 2 * i, b6get(y.B, i, 0)); sc.st(SC_SYS + 11 + 2 * i, b6get(y.B, i, 1)); sc.st(SC_SYS + 25 + i, v6get(y.Bm, i)); sc.st(SC_SYS + 32 + i, v6get(y.tr, i)); }
    sc.st(SC_SYS + 22, y.C[0]); sc.st(SC_SYS + 23, y.C[1]); sc.st(SC_SYS + 24, y.C[2]); sc.st(SC_SYS + 31, y.Cm);
    sc.st(SC_SYS + 38, y.tl[0]); sc.st(SC_SYS + 39, y.tl[1]); sc.st(SC_SYS + 40, y.tm);
}
// the root block [[J, [h]x], [[h]x^T, m 1]] with [h]x = [[0,-hz,hy],[hz,0,-hx],[-hy,hx,0]] from its 10 distinct values, as column pairs
template <typename V> JB_HD void root_block(const Sym3<V>& J, const Vec3<V>& h, const V& mt, Sym6P<V>& A) {
    using W = Pk2<V>;
    const V Z = V(0);
    A.p[s6p(0, 0)] = W(J.xx, J.xy); A.p[s6p(1, 0)] = W(J.xz, Z);    A.p[s6p(2, 0)] = W(-h.z, h.y);
    A.p[s6p(0, 1)] = W(J.xy, J.yy); A.p[s6p(1, 1)] = W(J.yz, h.z);  A.p[s6p(2, 1)] = W(Z, -h.x);
    A.p[s6p(1, 2)] = W(J.zz, -h.y); A.p[s6p(2, 2)] = W(h.x, Z);
    A.p[s6p(1, 3)] = W(-h.y, mt);   A.p[s6p(2, 3)] = W(Z, Z);
    A.p[s6p(2, 4)] = W(mt, Z);
    A.p[s6p(2, 5)] = W(Z, mt);
}
template <typename V> JB_HD void sys_load(const LaneScratch<V>& sc, StarSys<V>& y) {
    using W = Pk2<V>;
    Sym3<V> J;
    J.xx = sc.ld(SC_SYS + 0); J.xy = sc.ld(SC_SYS + 1); J.yy = sc.ld(SC_SYS + 2); J.xz = sc.ld(SC_SYS + 3); J.yz = sc.ld(SC_SYS + 4); J.zz = sc.ld(SC_SYS + 5);
    root_block<V>(J, v3<V>(sc.ld(SC_SYS + 6), sc.ld(SC_SYS + 7), sc.ld(SC_SYS + 8)), sc.ld(SC_SYS + 9), y.A);
#pragma unroll
    for (int ip = 0; ip < 3; ip++) {
        y.B[ip][0] = W(sc.ld(SC_SYS + 10 + 4 * ip), sc.ld(SC_SYS + 12 + 4 * ip)); y.B[ip][1] = W(sc.ld(SC_SYS + 11 + 4 * ip), sc.ld(SC_SYS + 13 + 4 * ip));
        y.Bm[ip] = W(sc.ld(SC_SYS + 25 + 2 * ip), sc.ld(SC_SYS + 26 + 2 * ip)); y.tr[ip] = W(sc.ld(SC_SYS + 32 + 2 * ip), sc.ld(SC_SYS + 33 + 2 * ip));
    }
    y.C[0] = sc.ld(SC_SYS + 22); y.C[1] = sc.ld(SC_SYS + 23); y.C[2] = sc.ld(SC_SYS + 24); y.Cm = sc.ld(SC_SYS + 31);
    y.tl[0] = sc.ld(SC_SYS + 38); y.tl[1] = sc.ld(SC_SYS + 39); y.tm = sc.ld(SC_SYS + 40);
}
template <typename V> JB_HD void fac_store_cx(const LaneScratch<V>& sc, const StarFactor<V>& F) { sc.st(SC_CX_LEAN, F.cx0); sc.st(SC_CX_LEAN + 1, F.cx1); }
template <typename V> JB_HD void fac_load_cx(const LaneScratch<V>& sc, StarFactor<V>& F) { F.cx0 = sc.ld(SC_CX_LEAN); F.cx1 = sc.ld(SC_CX_LEAN + 1); }
template <typename V> JB_HD void fac_store(const LaneScratch<V>& sc, const StarFactor<V>& F) {
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = 0; j <= i; j++) sc.st(SC_FAC + tri(i, j), s6get(F.S, i, j));
#pragma unroll
    for (int i = 0; i < 6; i++) { sc.st(SC_FAC + 21 + 2 * i, b6get(F.B, i, 0)); sc.st(SC_FAC + 22 + 2 * i, b6get(F.B, i, 1)); sc.st(SC_FAC + 36 + i, v6get(F.bm, i)); }
    sc.st(SC_FAC + 33, F.i11); sc.st(SC_FAC + 34, F.i12); sc.st(SC_FAC + 35, F.i22); sc.st(SC_FAC + 42, F.icm);
}
template <typename V> JB_HD void fac_load(const LaneScratch<V>& sc, StarFactor<V>& F) {
    using W = Pk2<V>;
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
        for (int ip = j / 2; ip < 3; ip++) F.S.p[s6p(ip, j)] = W(sc.ld(SC_FAC + tri(2 * ip, j)), sc.ld(SC_FAC + tri(2 * ip + 1, j)));
#pragma unroll
    for (int ip = 0; ip < 3; ip++) {
        F.B[ip][0] = W(sc.ld(SC_FAC + 21 + 4 * ip), sc.ld(SC_FAC + 23 + 4 * ip)); F.B[ip][1] = W(sc.ld(SC_FAC + 22 + 4 * ip), sc.ld(SC_FAC + 24 + 4 * ip));
        F.bm[ip] = W(sc.ld(SC_FAC + 36 + 2 * ip), sc.ld(SC_FAC + 37 + 2 * ip));
    }
    F.i11 = sc.ld(SC_FAC + 33); F.i12 = sc.ld(SC_FAC + 34); F.i22 = sc.ld(SC_FAC + 35); F.icm = sc.ld(SC_FAC + 42);
}
template <typename V> JB_HD void state_store(const LaneScratch<V>& sc, const LaneState<V>& s) {
    const V v[36] = {s.px, s.py, s.pz, s.qw, s.qx, s.qy, s.qz, s.pz_lo, s.qw_lo, s.qx_lo, s.qy_lo, s.qz_lo, s.vx, s.vy, s.vz, s.wx, s.wy, s.wz,
                     s.phi, s.phid, s.turns, s.th1, s.th2, s.thd1, s.thd2, s.wa[0], s.wa[1], s.wa[2], s.wl[0], s.wl[1], s.wl[2], s.wj[0], s.wj[1], s.wm, s.fail, V(0)};
#pragma unroll
    for (int i = 0; i < 35; i++) sc.st(SC_LSTATE + i, v[i]);
}
template <typename V> JB_HD void state_load(const LaneScratch<V>& sc, LaneState<V>& s) {
    V v[35];
#pragma unroll
    for (int i = 0; i < 35; i++) v[i] = sc.ld(SC_LSTATE + i);
    s.px = v[0]; s.py = v[1]; s.pz = v[2]; s.qw = v[3]; s.qx = v[4]; s.qy = v[5]; s.qz = v[6];
    s.pz_lo = v[7]; s.qw_lo = v[8]; s.qx_lo = v[9]; s.qy_lo = v[10]; s.qz_lo = v[11];
    s.vx = v[12]; s.vy = v[13]; s.vz = v[14]; s.wx = v[15]; s.wy = v[16]; s.wz = v[17];
    s.phi = v[18]; s.phid = v[19]; s.turns = v[20]; s.th1 = v[21]; s.th2 = v[22]; s.thd1 = v[23]; s.thd2 = v[24];
    s.wa[0] = v[25]; s.wa[1] = v[26]; s.wa[2] = v[27]; s.wl[0] = v[28]; s.wl[1] = v[29]; s.wl[2] = v[30]; s.wj[0] = v[31]; s.wj[1] = v[32]; s.wm = v[33]; s.fail = v[34];
}

// ---- compensated arithmetic for the position state.  two_sum: s + e == a + b exactly (Knuth); the adds are pinned with
// vadd_rn so that no optimisation re-associates them.
template <typename V> JB_HD void two_sum(const V a, const V b, V& s, V& e) {      // (by value: callers pass `s` as an input too)
    s = vadd_rn(a, b);
    const V bb = vadd_rn(s, -a);
    e = vadd_rn(vadd_rn(a, -vadd_rn(s, -bb)), vadd_rn(b, -bb));
}
// (hi, lo) += d, result renormalised so that hi is the rounding of the sum
// big_hi: the caller knows |hi| >= |d| (Dekker's three-operation sum is then exact)
template <typename V> JB_HD void comp_add(V& hi, V& lo, const V& d, bool big_hi = false) {
    V s, e;
    if (big_hi) { s = vadd_rn(hi, d); e = vadd_rn(d, -vadd_rn(s, -hi)); }
    else two_sum(hi, d, s, e);
    const V l = lo + e;
    hi = vadd_rn(s, l);
    lo = vadd_rn(l, -vadd_rn(hi, -s));
}
// the same for two values at once (a pair of quaternion components): Knuth's two_sum on packed adds - no products in it, so nothing the
// compiler could fuse once the addend is pinned
template <typename V> JB_HD void comp_add2(Pk2<V>& hi, Pk2<V>& lo, const Pk2<V>& d_in) {
    const Pk2<V> d = pk_pin(d_in);
    const Pk2<V> s = hi + d, bb = s - hi, e = (hi - (s - bb)) + (d - bb);
    const Pk2<V> l = lo + e, h2 = s + l;
    lo = l - (h2 - s);
    hi = h2;
}
// |q|^2 - 1 of a quaternion held as hi + lo, to ~1e-14: exact squares (fma residuals) and a compensated sum
template <typename V> JB_HD V quat_norm_excess(const V (&h)[4], const V (&l)[4]) {
    V p[4], r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) { p[k] = vmul_rn(h[k], h[k]); r[k] = vfma(h[k], h[k], -p[k]); }
    V s, e1, e2, e3;
    two_sum(p[0], p[1], s, e1);
    two_sum(s, p[2], s, e2);
    two_sum(s, p[3], s, e3);
    const V small = ((e1 + e2) + e3) + ((r[0] + r[1]) + (r[2] + r[3])) + V(2) * ((h[0] * l[0] + h[1] * l[1]) + (h[2] * l[2] + h[3] * l[3]));
    return vadd_rn(s, V(-1)) + small;       // s is within a few ulp of 1: the subtraction is exact
}
// q <- q / |q| for |q| = 1 + O(1e-6): first-order correction, applied to the hi/lo pair
template <typename V> JB_HD void quat_normalise_comp(V (&h)[4], V (&l)[4]) {
    const V half_eps = V(0.5) * quat_norm_excess(h, l);
#pragma unroll
    for (int k = 0; k < 4; k++) comp_add(h[k], l[k], -h[k] * half_eps);
}

#if defined(__HIPCC__)
// Split tables (LaneConsts): all 64 lanes of the wave copy the overlay's LM_SPLIT_OVL entries x 16 main lanes from the envs' tables in global
// memory into the scratch.  Load k of a lane fetches slot 4k + (lane / 16) of main lane (lane % 16): a wave-wide load lands in 64
// consecutive floats of LDS, and every load is an immediate offset from ONE per-lane pointer (ovl_src already points at the lane's first slot).
static_assert(SC_FAC + 43 - SC_SYS == LM_SPLIT_OVL, "the overlay is exactly the parked system + factorisation");
static_assert((LM_HOT - LM_INV) % 4 == 0, "the three runs of the overlay start on multiples of four slots");
__device__ __forceinline__ void restage_overlay(const LaneConsts<float>& c) {
    constexpr int K0 = (LM_HOT - LM_INV) / 4;
    float v[K0 + 5];
#pragma unroll
    for (int k = 0; k < K0; k++) v[k] = c.ovl_src[16 * k];
#pragma unroll
    for (int k = 0; k < 4; k++) v[K0 + k] = c.ovl_src[4 * (LM_UC_D - LM_INV) + 16 * k];
    v[K0 + 4] = c.ovl_src[4 * (LM_PE_IS - 1 - LM_INV)];
#pragma unroll
    for (int k = 0; k < K0 + 5; k++) c.ovl_dst[64 * k] = v[k];
    wave_sync();
}
#endif

// ----------------------------------------------------------------------------- phase B of a substep: the contact solve
// Primal Newton on the active set, then MuJoCo's Euler step with implicit joint damping, ONE loop:
//   while the active set changes:  H(active set at y) y' = tau + contact rhs          (M without damping)
//   then:  (M + h diag(b)) qacc = tau + qfrc_constraint(y)
// Main lanes own the iterate and the solves; every decision that steers the loop is broadcast so that the helper groups follow the same
// control flow.  Reads phase A's products (the system `sys`, the slot plans, the built rows in the scratch, the warm start in `s`) and
// leaves them as they were: it may be called twice for one substep.
//
// LS = false - the hot instantiation: full Newton steps, rank-one passes while they suffice; stops when every env's set repeats (exact
//   minimisers) or at o.max_newton checks.  Returns whether some env of the wave was still moving at the cap (`capped`: which, on the
//   main lanes); such a wave calls the other instantiation for the same substep (substep_impl).
// LS = true - MuJoCo's Newton solver (reference jitterbug.xml:18 leaves <option> at its defaults: Newton, exact line search): every
//   update goes to the minimum of the cost along the Newton direction.  The cost along  y + alpha d,  d = (this pass's solution) - y,  is
//   convex and piecewise quadratic:
//       phi'(alpha) = d.(M (y + alpha d) - tau) + sum over contacts and pyramid edges of D min(0, r + alpha s) s,
//   r, s = the edge's residual at y and its slope along d.  Safeguarded Newton on phi' in [0, 1] with a fixed count; every lane group
//   sweeps its share of the contacts (contact_sweep mode 3), the main lanes hold M and steer.  alpha = 1 - the full Newton step - whenever
//   that is still downhill.  No rank-one passes.  Cold code: it runs for a few substeps in ten million.
#if !defined(__HIPCC__)
inline int g_ls_trace = 0;                    // host harness only: print the line-searched iteration
inline long g_ls_stats[4] = {0, 0, 0, 0};      // host harness only: substeps solved a second time / outer passes of those solves / line searches that shortened a step / solves that hit NEWTON_LS_CAP
#endif
#ifndef JB_CAPTURE_SLOTS
#define JB_CAPTURE_SLOTS 256
#endif
#ifndef JB_LS_CAP
#define JB_LS_CAP 60
#endif
#ifndef JB_LS_EVALS
#define JB_LS_EVALS 8
#endif
#ifndef JB_TINY_STEP
#define JB_TINY_STEP 1e-12f
#endif
constexpr int NEWTON_LS_CAP = JB_LS_CAP;          // outer passes of the line-searched solve
constexpr int NEWTON_LS_EVALS = JB_LS_EVALS;      // evaluations of phi' per line search
constexpr float NEWTON_TINY_STEP = JB_TINY_STEP;  // |d|^2 / |y|^2 below which a step is rounding noise and the iterate is accepted
#if !defined(__HIPCC__)
template <typename V, typename MK> inline void ls_trace_pass(int, const V&, const MK&, const MK&, const V (&)[6], const V (&)[6], const V (&)[2], const V (&)[2], const V&, const V&) {}
template <typename T> inline void ls_trace_pass(int it, const Quad<T>& al, const Mask4& take, const Mask4& tiny, const Quad<T> (&yr)[6], const Quad<T> (&nyr)[6], const Quad<T> (&yl)[2], const Quad<T> (&nyl)[2], const Quad<T>& ym, const Quad<T>& nym) {
    double dn = 0, yn = 0;
    for (int i = 0; i < 6; i++) { dn += (double)(nyr[i].v[0] - yr[i].v[0]) * (nyr[i].v[0] - yr[i].v[0]); yn += (double)nyr[i].v[0] * nyr[i].v[0]; }
    for (int l = 0; l < 4; l++) for (int j = 0; j < 2; j++) { dn += (double)(nyl[j].v[l] - yl[j].v[l]) * (nyl[j].v[l] - yl[j].v[l]); yn += (double)nyl[j].v[l] * nyl[j].v[l]; }
    dn += (double)(nym.v[0] - ym.v[0]) * (nym.v[0] - ym.v[0]); yn += (double)nym.v[0] * nym.v[0];
    std::printf("  pass it=%2d alpha %.6f take %d tiny %d  |step|/|y| %.3e  y0 %.8g yl0 %.8g ym %.8g\n", it, (double)al.v[0], (int)take.v[0], (int)tiny.v[0], std::sqrt(dn / (yn + 1e-300)), (double)nyr[0].v[0], (double)nyl[0].v[0], (double)nym.v[0]);
}
template <typename MK, typename U> inline void ls_trace_check(int, const MK&, const U&, const U&, const U&, const U&, const U&, const U&) {}
inline void ls_trace_check(int it, const Mask4& unc, const UQuad& b0, const UQuad& p0, const UQuad& b1, const UQuad& p1, const UQuad& xh, const UQuad& pxh) {
    std::printf("  check it=%2d changed %d | bw0 %08x %08x %08x %08x (flips %08x %08x %08x %08x) bw1 flips %08x %08x %08x %08x xh-changed %d%d%d%d\n", it, (int)unc.v[0], b0.v[0], b0.v[1], b0.v[2], b0.v[3],
                b0.v[0] ^ p0.v[0], b0.v[1] ^ p0.v[1], b0.v[2] ^ p0.v[2], b0.v[3] ^ p0.v[3], b1.v[0] ^ p1.v[0], b1.v[1] ^ p1.v[1], b1.v[2] ^ p1.v[2], b1.v[3] ^ p1.v[3],
                (int)(xh.v[0] != pxh.v[0]), (int)(xh.v[1] != pxh.v[1]), (int)(xh.v[2] != pxh.v[2]), (int)(xh.v[3] != pxh.v[3]));
}
#endif
template <typename V, bool PAIR, bool LS>
JB_HD bool newton_phase(const LaneModel<V>& m, const LaneScratch<V>& sc, LaneState<V>& s, const SimOpts& o, const bool xtra, const SlotPlan& plan, const SpreadPlan<V>& spl,
                        const Mat3<V>& Rw, StarSys<V>& sys, const Vec3<V> (&dk)[3], const bool any_contact, const typename lane_traits<V>::mask& env_con,
                        Pk2<V> (&yr)[3], V (&yl)[2], V& ym, typename lane_traits<V>::mask& capped) {
    using MK = typename lane_traits<V>::mask;
    using U = typename lane_traits<V>::uint;
    using W = Pk2<V>;
    const V h = m.c[LM_H];
    auto store_root = [&](int at, const W (&y)[3]) {      // a root 6-vector (three row pairs) into the scratch
#pragma unroll
        for (int ip = 0; ip < 3; ip++) { sc.st(at + 2 * ip, pk_lo(y[ip])); sc.st(at + 2 * ip + 1, pk_hi(y[ip])); }
    };
    auto load_root = [&](int at, W (&y)[3]) {
#pragma unroll
        for (int ip = 0; ip < 3; ip++) y[ip] = W(sc.ld(at + 2 * ip), sc.ld(at + 2 * ip + 1));
    };
    const bool is_main = (sc.grp == 0);
    const bool g1 = o.offload && sc.grp == 1;       // the replica group (its LDS stores repeat the main lanes': same address, same value)
    const bool rep = is_main || g1;
    const int cap = LS ? NEWTON_LS_CAP : o.max_newton;
    bool wave_capped = false;
    JB_PROF_T0();
    {
        NewtonAcc<V> acc;
        StarFactor<V> fac;      // factorisation of the last Newton system (main lanes)
        bool final_pass = !any_contact;
        MK unconverged = lt(V(0), V(1));
        MK fac_valid = lt(V(1), V(0)), fast_env = lt(V(1), V(0));
        // LS only (a lane word, not three masks: the scalar registers are the scarce ones in this loop).
        //   bit 0  the last update was a shortened step: its iterate minimises no set and is never accepted as converged
        //   bit 1  the last step was below the iterate's rounding: whatever the sets say, this is the solution
        //   bit 2  the last shortened step stayed on its pass's set - that set's minimiser is the WHOLE step: take it next time
        U ls_flags = zero_u<V>();
        U prev_bw0 = zero_u<V>(), prev_bw1 = zero_u<V>(), prev_xh = zero_u<V>();
        if (any_contact) {
            if (is_main) {      // warm start (world linear part rotated into the root frame)
                Vec3<V> lw = mulT(Rw, v3<V>(s.wl[0], s.wl[1], s.wl[2]));
                yr[0] = W(s.wa[0], s.wa[1]); yr[1] = W(s.wa[2], lw.x); yr[2] = W(lw.y, lw.z);
                yl[0] = s.wj[0]; yl[1] = s.wj[1]; ym = s.wm;
                store_root(SC_Y, yr);
                sc.st(SC_Y + 6, yl[0]); sc.st(SC_Y + 7, yl[1]); sc.st(SC_Y + 8, ym);
            }
            wave_sync();          // the helper groups read the iterate from the scratch
        } else {
            acc_clear(acc);
        }
        const V hb1 = o.implicit_damp ? h * m.c[LM_B1] : V(0), hb2 = o.implicit_damp ? h * m.c[LM_B2] : V(0);
        // offload: the replica leaves every full pass with the factorisation of M + h diag(b).  A wave without any contact makes one turn of the
        // outer loop too (no sweep, zero accumulator), so that this factorisation always comes from the SAME instructions: an env's bits must
        // not depend on whether a wave-mate has a contact (two inlined copies of one expression may fuse their multiply-adds differently).
        const bool have_dfac = o.offload != 0;
#ifdef JB_WAVE_STATS
        if (!LS && is_main && any_contact) { s.st_contact = s.st_contact + V(1); s.st_slots = s.st_slots + V((float)__builtin_popcount(plan.live)); }
        if (!LS) stats_hist(o, sc, plan, xtra, any_contact);
#endif
        // Two nested loops.  OUTER: one FULL pass per turn - a sweep over every live slot and a new factorisation, always at the top, so that
        // the factorisation object has a single definition per turn (a conditional redefinition inside one loop costs a register copy per
        // value and iteration once the object is live after the loop, as the replica's is).  INNER: checks of the active set at the new
        // iterate, with rank-one passes while every env that still moves differs from its factorisation by a single pyramid edge.
        int it = 0;
        if (any_contact || have_dfac) {
#pragma unroll 1
            for (;;) {
                // ---- full pass (reads the iterate of the last check from the scratch: rank-one results are stored only after this pass)
                if (any_contact) {
                    const int sweep_rounds = contact_sweep<V, PAIR, LS>(m, sc, xtra, plan, spl, 0, dk, acc, have_dfac);
                    (void)sweep_rounds;
                    prev_bw0 = acc.bw0; prev_bw1 = acc.bw1; prev_xh = acc.xh;
                    JB_PROF_ADD(o, 2);
#ifdef JB_WAVE_STATS
                    if (is_main) s.st_sweeps = s.st_sweeps + V(1);
#if defined(__HIPCC__)
                    if (o.hist && (threadIdx.x & 63) == 0) {      // full sweeps by the rounds they took: [53 + spread rounds (0-3)], [57 + rest rounds (0-3)]
                        o.hist[53 + ((sweep_rounds & 7) < 3 ? (sweep_rounds & 7) : 3)] += 1ull;
                        o.hist[57 + ((sweep_rounds >> 3) < 3 ? (sweep_rounds >> 3) : 3)] += 1ull;
                    }
#endif
#endif
                }
                // (LS: every lane group comes along - the line search below has hand-over points that all groups must pass; the helper groups'
                // solve works on whatever their registers hold and nobody reads it)
                if (o.offload || is_main || LS) {
                    W nyr[3];
                    V nyl[2], nym;
                    if (o.lean) sys_load(sc, sys);
                    // offload: EVERY lane runs this - the replica on M + h diag(b) (zero accumulator from the sweep, hb on its diagonal), in
                    // every full pass; the other helper groups on whatever their registers hold (nobody reads their result).  No lane
                    // predicate on register-only work.
                    const V hx1 = g1 ? hb1 : V(0), hx2 = g1 ? hb2 : V(0);
                    star_solve<V, PAIR>(sys, acc, hx1, hx2, fac, nyr, nyl, nym);
                    if (o.lean && is_main) { fac_store(sc, fac); if (PAIR) fac_store_cx(sc, fac); }
                    JB_PROF_ADD(o, 6);
                    V ls_alpha = V(1);
                    MK ls_tiny = lt(V(1), V(0));
                    if (LS && any_contact) {
                        const int box = ls_mailbox(sc);
                        // (cold code: element by element, out of the packed layouts)
                        V dr[6], dl[2], dm, ys[6];
#pragma unroll
                        for (int i = 0; i < 6; i++) { ys[i] = v6get(yr, i); dr[i] = v6get(nyr, i) - ys[i]; }
                        dl[0] = nyl[0] - yl[0]; dl[1] = nyl[1] - yl[1]; dm = nym - ym;
                        auto matvec = [&](const V (&vr)[6], const V (&vl)[2], const V& vm, V (&wr)[6], V (&wl)[2], V& wm) {
                            wl[0] = sys.C[0] * vl[0] + sys.C[1] * vl[1]; wl[1] = sys.C[1] * vl[0] + sys.C[2] * vl[1]; wm = sys.Cm * vm;
#pragma unroll
                            for (int i = 0; i < 6; i++) {
                                V t = quad_sum(b6get(sys.B, i, 0) * vl[0] + b6get(sys.B, i, 1) * vl[1]) + v6get(sys.Bm, i) * vm;
#pragma unroll
                                for (int j = 0; j < 6; j++) t = t + s6get(sys.A, i, j) * vr[j];
                                wr[i] = t;
                                wl[0] = wl[0] + b6get(sys.B, i, 0) * vr[i]; wl[1] = wl[1] + b6get(sys.B, i, 1) * vr[i]; wm = wm + v6get(sys.Bm, i) * vr[i];
                            }
                        };
                        V myr[6], myl[2], mym, mdr[6], mdl[2], mdm;
                        matvec(ys, yl, ym, myr, myl, mym);
                        matvec(dr, dl, dm, mdr, mdl, mdm);
                        V g0 = quad_sum(dl[0] * (myl[0] - sys.tl[0]) + dl[1] * (myl[1] - sys.tl[1])) + dm * (mym - sys.tm);
                        V g1_ = quad_sum(dl[0] * mdl[0] + dl[1] * mdl[1]) + dm * mdm;
#pragma unroll
                        for (int i = 0; i < 6; i++) { g0 = g0 + dr[i] * (myr[i] - v6get(sys.tr, i)); g1_ = g1_ + dr[i] * mdr[i]; }
                        if (is_main) {
#pragma unroll
                            for (int i = 0; i < 6; i++) sc.st(box + i, dr[i]);
                            sc.st(box + 6, dl[0]); sc.st(box + 7, dl[1]); sc.st(box + 8, dm); sc.st(box + 9, V(1));
                        }
                        V lo = V(0), hi = V(1), al = V(1);
#pragma unroll 1
                        for (int li = 0; li < NEWTON_LS_EVALS; li++) {
                            wave_sync();
                            contact_sweep<V, PAIR, LS>(m, sc, xtra, plan, spl, 3, dk, acc);
                            const V f1 = g0 + al * g1_ + acc.ls_g, f2 = g1_ + acc.ls_h;
                            const auto below = lt(f1, V(0));
                            lo = sel(below, al, lo); hi = sel(below, hi, al);
                            V an = al - f1 * vrcp(vmax(f2, V(1e-30)));
                            an = sel(mand(gt(an, lo), lt(an, hi)), an, V(0.5) * (lo + hi));
                            al = an;
                            wave_sync();
                            if (is_main) sc.st(box + 9, al);
                        }
                        {   // A step below the iterate's own rounding: nothing left to gain.  An edge whose residual is zero to within the rounding of the
                            // solve flips for ever - the minimisers of the two sets lie a few 1e-6 |y| apart (fp32 through a Schur solve) and each sends
                            // the iteration to the other.  A line-searched Newton iteration on this convex cost is past its large steps within a few
                            // passes, so what counts as rounding GROWS with the pass count: |d|^2 < 1e-12 * 2^it * |y|^2 - 1e-6 |y| at once,
                            // 3e-5 |y| after ten passes, and every iteration ends (the error it leaves is bounded by the step it declined to take).
                            V dn = quad_sum(dl[0] * dl[0] + dl[1] * dl[1]) + dm * dm, yn = quad_sum(yl[0] * yl[0] + yl[1] * yl[1]) + ym * ym;
#pragma unroll
                            for (int i = 0; i < 6; i++) { dn = dn + dr[i] * dr[i]; yn = yn + ys[i] * ys[i]; }
                            ls_tiny = lt(dn, V(NEWTON_TINY_STEP * (float)(1u << (it < 30 ? it : 30))) * yn);
                        }
                        // A minimum short of the full step sits on a kink of the cost - an edge's residual crossing zero - and the next pass must be
                        // built on the set BEYOND it, or it would propose the same step again: go a thousandth further, so that the check
                        // classifies that edge on the far side.  Such an iterate is never accepted as converged (`ls_flags` bit 0).
                        const auto whole = gt(al, V(0.999));
                        ls_alpha = sel(whole, V(1), al * V(1.001) + V(1e-6));
                        wave_sync();      // (the mailbox is the reduction hand-over of the next full sweep)
                    }
                    if (is_main) {
                        // envs whose active set already repeated keep their (exact) solution, rank-one envs theirs
                        const MK take = mand(unconverged, mnot(fast_env));
                        if (LS) {      // the step to the minimum of the cost along the Newton direction
                            // Close to the solution d is tiny, phi' is a difference of large terms and alpha is rounding noise: a shortened step that
                            // stayed on the set the pass was built on (the cost is ONE quadratic from y to there) is followed by the whole step.
                            ls_alpha = sel(mor(neq_u(and_u(ls_flags, zero_u<V>() + 4u), zero_u<V>()), ls_tiny), V(1), ls_alpha);
                            const MK part = lt(ls_alpha, V(1));      // (a whole step keeps the pass's own bits)
#pragma unroll
                            for (int ip = 0; ip < 3; ip++) nyr[ip] = selw(part, yr[ip] + W(ls_alpha) * (nyr[ip] - yr[ip]), nyr[ip]);
                            nyl[0] = sel(part, yl[0] + ls_alpha * (nyl[0] - yl[0]), nyl[0]); nyl[1] = sel(part, yl[1] + ls_alpha * (nyl[1] - yl[1]), nyl[1]);
                            nym = sel(part, ym + ls_alpha * (nym - ym), nym);
                            ls_flags = mbit(mand(take, part)) + mbit(mand(take, ls_tiny)) * 2u;
#if !defined(__HIPCC__)
                            if (g_ls_trace) {
                                V ty[6], tn[6];
                                for (int i = 0; i < 6; i++) { ty[i] = v6get(yr, i); tn[i] = v6get(nyr, i); }
                                ls_trace_pass(it, ls_alpha, take, ls_tiny, ty, tn, yl, nyl, ym, nym);
                            }
                            g_ls_stats[1]++;
                            if (any_lane(mand(take, part))) g_ls_stats[2]++;
#endif
                        }
#pragma unroll
                        for (int ip = 0; ip < 3; ip++) yr[ip] = selw(take, nyr[ip], yr[ip]);
                        yl[0] = sel(take, nyl[0], yl[0]); yl[1] = sel(take, nyl[1], yl[1]); ym = sel(take, nym, ym);
                        fac_valid = mnot(fast_env);         // a rank-one env's factorisation is one edge behind its set
                        store_root(SC_Y, yr);
                        sc.st(SC_Y + 6, yl[0]); sc.st(SC_Y + 7, yl[1]); sc.st(SC_Y + 8, ym);
                    }
                }
                JB_PROF_ADD(o, 3);
                wave_sync();          // new iterate stored by the main lanes -> read by every group's next sweep
                if (!any_contact) break;      // (nothing to iterate on: the turn was for the replica's factorisation)
                // ---- checks (and rank-one passes) until the sets repeat or some env needs a full pass
                bool full_pass = false;
#pragma unroll 1
                for (;;) {
                    it++;
                    // cheap pass: only the active set at the new iterate.  The ENV's set changed if any lane of the quad
                    // saw a different record; when nobody's changed, every y is the exact minimiser
                    contact_sweep<V, PAIR, LS>(m, sc, xtra, plan, spl, 2, dk, acc);
                    JB_PROF_ADD(o, 1);
#ifdef JB_WAVE_STATS
                    if (is_main) s.st_checks = s.st_checks + V(1);
#endif
                    unsigned fin = 0u, full = 1u;
                    if (is_main) {
                        MK changed = mor(mor(neq_u(acc.bw0, prev_bw0), neq_u(acc.bw1, prev_bw1)), neq_u(acc.xh, prev_xh));
                        unconverged = neq_u(quad_sum_u(mbit(changed)), zero_u<V>());
#if !defined(__HIPCC__)
                        if (LS && g_ls_trace) ls_trace_check(it, unconverged, acc.bw0, prev_bw0, acc.bw1, prev_bw1, acc.xh, prev_xh);
#endif
                        if (LS) {      // the last update came from the line search
                            const MK shortened = neq_u(and_u(ls_flags, zero_u<V>() + 1u), zero_u<V>()), settled = neq_u(and_u(ls_flags, zero_u<V>() + 2u), zero_u<V>());
                            ls_flags = mbit(mand(shortened, mnot(unconverged))) * 4u;
                            unconverged = mand(mor(unconverged, shortened), mnot(settled));
                        }
#if defined(JB_WAVE_STATS) && defined(__HIPCC__)
                                            if (!LS && o.hist && !xtra) {      // how many active-set bits flipped per unconverged env (ordinary substeps: exact records)
                                                const unsigned fl = quad_sum_u((unsigned)__builtin_popcount(acc.bw0 ^ prev_bw0));
                                                const bool unc = unconverged;
                                                const unsigned long long m_unc = __builtin_amdgcn_ballot_w64(unc), m_multi = __builtin_amdgcn_ballot_w64(unc && fl > 1u);
                                                const unsigned long long m1 = __builtin_amdgcn_ballot_w64(unc && fl == 1u), m2 = __builtin_amdgcn_ballot_w64(unc && fl == 2u), m3 = __builtin_amdgcn_ballot_w64(unc && fl == 3u), m4 = __builtin_amdgcn_ballot_w64(unc && fl >= 4u);
                                                if ((threadIdx.x & 63) == 0 && m_unc) {
                                                    o.hist[44] += 1ull; if (!m_multi) o.hist[45] += 1ull;
                                                    o.hist[46] += __builtin_popcountll(m1) / 4; o.hist[47] += __builtin_popcountll(m2) / 4; o.hist[48] += __builtin_popcountll(m3) / 4; o.hist[49] += __builtin_popcountll(m4) / 4;
                                                }
                                                // of the multi-flip envs: those whose flips sit in different lanes (at most one per leg) - what a per-lane rank-k pass could take
                                                const unsigned lane_fl = (unsigned)__builtin_popcount(acc.bw0 ^ prev_bw0) + (unsigned)__builtin_popcount(acc.bw1 ^ prev_bw1);
                                                const bool spread = quad_sum_u(lane_fl > 1u ? 1u : 0u) == 0u;
                                                const unsigned long long m_sp = __builtin_amdgcn_ballot_w64(unc && fl > 1u && spread), m_sp2 = __builtin_amdgcn_ballot_w64(unc && fl == 2u && spread);
                                                const unsigned long long m_bad = __builtin_amdgcn_ballot_w64(unc && !spread);
                                                if ((threadIdx.x & 63) == 0 && m_unc) {
                                                    o.hist[50] += __builtin_popcountll(m_sp) / 4; o.hist[51] += __builtin_popcountll(m_sp2) / 4;
                                                    if (m_multi && !m_bad) o.hist[52] += 1ull;          // a changed-set check that needs a full pass today and would not with per-lane rank-k
                                                }
                                            }
#endif
                        if (!any_lane(unconverged) || it >= cap) {
                            capped = unconverged;                      // (all false unless the cap ended the iteration)
                            fin = any_lane(unconverged) ? 2u : 1u;
                        } else {
                            // An env whose set differs from the factored one by a single pyramid edge of a leg slot (exact
                            // records) takes a rank-one pass; the decision is the ENV's own (its records, its factorisation),
                            // so its arithmetic never depends on its wave-mates.  A full pass runs only if some env needs one.
                            const U dfl = xor_u(acc.bw0, prev_bw0), dfl1 = xor_u(acc.bw1, prev_bw1);
                            const MK one_flip = eq_u(quad_sum_u(popc_u(dfl) + popc_u(dfl1)), zero_u<V>() + 1u);
                            const MK xh_same = eq_u(quad_sum_u(mbit(neq_u(acc.xh, prev_xh))), zero_u<V>());
                            U f_entry;
                            MK f_is, f_plus, f_tan2, f_on;
                            flip_decode(dfl, dfl1, acc.bw0, acc.bw1, plan.live, f_entry, f_is, f_plus, f_tan2, f_on);
                            // the flipped contact's row must be in the row cache (beyond ROW_K live slots rows are rebuilt per pass)
                            const MK undecodable = mand(neq_u(or_u(dfl, dfl1), zero_u<V>()), mnot(mand(f_is, lt_u(f_entry, (unsigned)ROW_K))));
                            const MK rows_ok = eq_u(quad_sum_u(mbit(undecodable)), zero_u<V>());
                            fast_env = mand(mand(unconverged, fac_valid), mand(mand(one_flip, xh_same), rows_ok));
                            if (LS || !o.rank_one) fast_env = lt(V(1), V(0));      // (the line-searched iteration works on full passes)
                            full = any_lane(mand(unconverged, mnot(fast_env))) ? 1u : 0u;
                            if (!LS && any_lane(fast_env)) {
                                W fyr[3];
                                V fyl[2], fym;
                                if (o.lean) { fac_load(sc, fac); if (PAIR) fac_load_cx(sc, fac); }
                                rank_one_pass<V, PAIR>(sc, fac, dk, m.c[LM_MU], f_entry, f_is, f_plus, f_tan2, f_on, yr, yl, ym, fyr, fyl, fym);
#pragma unroll
                                for (int ip = 0; ip < 3; ip++) yr[ip] = selw(fast_env, fyr[ip], yr[ip]);
                                yl[0] = sel(fast_env, fyl[0], yl[0]); yl[1] = sel(fast_env, fyl[1], yl[1]); ym = sel(fast_env, fym, ym);
#ifdef JB_WAVE_STATS
                                s.st_fast = s.st_fast + V(1);
#endif
                            }
                            prev_bw0 = acc.bw0; prev_bw1 = acc.bw1; prev_xh = acc.xh;      // the sets the new iterates are solved for
                        }
                    }
                    {
                        const unsigned f = wave_bcast_u(fin);
                        final_pass = f != 0u;
                        wave_capped = f == 2u;
                    }
                    full_pass = wave_bcast_u(full) != 0u;
                    JB_PROF_ADD(o, 7);
                    if (final_pass || full_pass) break;
                    if (is_main) {      // rank-one passes only: their iterates go to the scratch for the next check
                        fac_valid = mand(fac_valid, mnot(fast_env));
                        store_root(SC_Y, yr);
                        sc.st(SC_Y + 6, yl[0]); sc.st(SC_Y + 7, yl[1]); sc.st(SC_Y + 8, ym);
                    }
                    JB_PROF_ADD(o, 3);
                    wave_sync();
                }
                if (final_pass) break;
            }
        }
#if !defined(__HIPCC__)
        if (LS && is_main && wave_capped) g_ls_stats[3]++;
#endif
        // ---- the final pass: MuJoCo's Euler step with implicit joint damping.
        // At the minimiser  H y = tau + sum B^T W ahat,  so the constraint force is  qfrc = M y - tau  and the step solves
        //     (M + h diag(b)) qacc = tau + qfrc = M y,  i.e.  qacc = y - d,   (M + h diag(b)) d = h diag(b) y        (nonzero only in the leg rows).
        // star_subst adds tau itself, so the contact envs cancel it (replicated parts enter the quad sums as 1/4).
        // An env without contacts has qfrc = 0 exactly, whatever the rest of its wave is doing: plain solve.
        if (have_dfac || rep) {
            W nyr[3];
            V nyl[2], nym;
            if (o.lean) sys_load(sc, sys);
            if (have_dfac) {
                // Every lane takes the iterate from the scratch (what the main lanes hold in registers) and runs the substitution on whatever
                // factorisation it holds: the REPLICA's is the one of M + h diag(b), and only its result is handed on.
                load_root(SC_Y, yr);
                yl[0] = sc.ld(SC_Y + 6); yl[1] = sc.ld(SC_Y + 7); ym = sc.ld(SC_Y + 8);
            }
            if (any_contact) {
#pragma unroll
                for (int ip = 0; ip < 3; ip++) acc.rr[ip] = selw(env_con, W(-0.25) * sys.tr[ip], W(0));
                acc.rl[0] = sel(env_con, hb1 * yl[0] - sys.tl[0], V(0));
                acc.rl[1] = sel(env_con, hb2 * yl[1] - sys.tl[1], V(0));
                acc.rm = sel(env_con, V(-0.25) * sys.tm, V(0));
            }
            if (!have_dfac) {      // (no replica - the LEAN variant, a single lane group: the main lanes factorise on the spot)
                StarFactor<V> fd;
                star_factor<V, false>(sys, acc, hb1, hb2, fd);           // acc holds only right-hand sides here
                star_subst<V>(fd, acc.rr, sys.tr, sys.tl[0] + acc.rl[0], sys.tl[1] + acc.rl[1], sys.tm + quad_sum(acc.rm), nyr, nyl, nym);
            } else {
                star_subst<V>(fac, acc.rr, sys.tr, sys.tl[0] + acc.rl[0], sys.tl[1] + acc.rl[1], sys.tm + quad_sum(acc.rm), nyr, nyl, nym);
            }
            JB_PROF_ADD(o, 6);
            if (have_dfac) {
                if (g1) {
                    store_root(SC_RED, nyr);
                    sc.st(SC_RED + 6, nyl[0]); sc.st(SC_RED + 7, nyl[1]); sc.st(SC_RED + 8, nym);
                }
                wave_sync();
                load_root(SC_RED, nyr);
                nyl[0] = sc.ld(SC_RED + 6); nyl[1] = sc.ld(SC_RED + 7); nym = sc.ld(SC_RED + 8);
            }
#pragma unroll
            for (int ip = 0; ip < 3; ip++) yr[ip] = selw(env_con, yr[ip] - nyr[ip], nyr[ip]);
            yl[0] = sel(env_con, yl[0] - nyl[0], nyl[0]); yl[1] = sel(env_con, yl[1] - nyl[1], nyl[1]); ym = sel(env_con, ym - nym, nym);
        }
    }
    return wave_capped;
}

// ----------------------------------------------------------------------------- the substep
template <typename V, bool PAIR = false>
JB_HD bool substep_impl(const LaneModel<V>& m, const LaneScratch<V>& sc, LaneState<V>& s, const V& ctrl, const SimOpts& o, const bool xtra, const bool xbody, const Mat3<V>& Rw) {
    using MK = typename lane_traits<V>::mask;
    using U = typename lane_traits<V>::uint;
    const V h = m.c[LM_H];
    const Vec3<V> w = v3<V>(s.wx, s.wy, s.wz);
    bool any_contact = false;
    unsigned live_slots = 0;            // wave-uniform: slots in which some lane has a contact
    MK env_con = lt(V(1), V(0));       // this env (quad) has at least one contact
    JB_PROF_T0();

    const bool is_main = (sc.grp == 0);
    const bool g1 = o.offload && sc.grp == 1;       // the replica group (its LDS stores repeat the main lanes': same address, same value)
    const bool rep = is_main || g1;
    // AUX lanes (SimOpts::aux: lane groups 2 and 3): the same phase A / C on the motor body (lane 0 of the quad) and the root body's own mass
    // (lane 1) as "legs"; what they find goes to the main lanes (group 0 <- 2) and the replica (1 <- 3) by cross-lane swaps below.
    const bool axl = o.aux != 0 && sc.grp >= 2;
    const int axoff = 2 * sc.gstride;               // lane distance to the partner group
    const MK ax_mask = axl ? lt(V(0), V(1)) : lt(V(1), V(0));
    const MK ax_motor = mand(ax_mask, eq_u(quad_lane_id((const V*)nullptr), 0u));
    const MK lane_ok = mnot(ax_mask);               // this lane stands for a real leg: it may have contacts
    if (axl) {      // the aux lanes' "leg" state: the motor's angle and rate in the shoulder slot of lane 0, nothing else
        s.th1 = sel(ax_motor, s.phi, V(0)); s.thd1 = sel(ax_motor, s.phid, V(0)); s.th2 = V(0); s.thd2 = V(0);
    }
    StarSys<V> sys;     // written and read by the main lanes (and their replica) only
    if (rep || axl) {   // ================= phase A: kinematics, composite inertia, bias forces, contact candidates -> scratch
        if (!axl) { sc.st3(SC_ST, w); sc.st(SC_ST + 3, s.thd1); sc.st(SC_ST + 4, s.thd2); sc.st(SC_ST + 5, s.phid); }      // for the helper groups
        const Mat3<V>& R = Rw;                                       // root rotation of the normalised quaternion (substep())
        Vec3<V> nb = v3<V>(R.m[6], R.m[7], R.m[8]);                 // R^T ez
        Vec3<V> u = mulT(R, v3<V>(s.vx, s.vy, s.vz));
        Vec3<V> AO = -mulT(R, ldv3(m, LM_GRAV));                    // fictitious root acceleration = -g (root coords)

        // ---- own leg kinematics.  The upper and the lower body of a leg go through the same formulas on different data: from here to the end of
        // phase A they travel as PAIRS (W: lo = shoulder / upper leg, hi = knee / lower leg), one packed instruction for the two of them.
        using W = Pk2<V>;
        Vec3<V> a1 = ldv3(m, LM_A1), e1 = ldv3(m, LM_E1);
        V s1, c1, s2, c2;
        {
            W sn, cs;
            sincos_small(W(s.th1, s.th2), sn, cs);
            s1 = pk_lo(sn); s2 = pk_hi(sn); c1 = pk_lo(cs); c2 = pk_hi(cs);
        }
        if (any_lane(mand(lane_ok, mor(gt(vabs(s.th1), V(0.9)), gt(vabs(s.th2), V(0.9)))))) { vsincos_pi(s.th1, s1, c1); vsincos_pi(s.th2, s2, c2); }      // (the quarter-turn reduction: good to 1e-7 for any angle a hinge can reach, and no libm slow path in the kernel)
        V sp, cp;
        vsincos_pi(s.phi, sp, cp);
        if (o.aux) { s1 = sel(ax_motor, sp, s1); c1 = sel(ax_motor, cp, c1); }      // the motor's angle is wrapped to [-pi, pi): its own sine / cosine routine
        const Mat3<W> Rj = rodrigues(pk(e1, ldv3(m, LM_E2)), W(s1, s2), W(c1, c2));      // the two hinge rotations
        const Mat3<V> R1 = lo(Rj);
        const Mat3<V> R12 = mul(R1, hi(Rj));
        const Mat3<W> RR = pk(R1, R12);                                      // body rotations: (upper, lower)
        const Vec3<W> t_ae = mul(both(R1), pk(ldv3(m, LM_DA2), ldv3(m, LM_E2)));       // (knee anchor - a1, knee axis)
        Vec3<V> a2 = a1 + lo(t_ae);
        Vec3<V> e2 = hi(t_ae);
        const Vec3<W> aw = pk(a1, a2), ew = pk(e1, e2);                      // hinge anchors and axes of the two bodies
        const Vec3<W> ccw = aw + mul(RR, pk(ldv3(m, LM_DC1), ldv3(m, LM_DC2)));        // centres of mass
        const Vec3<V> cc2 = hi(ccw);
        const Sym3<W> Iw = rotate(RR, pk(ldsym(m, LM_I1), ldsym(m, LM_I2)));           // inertias about the centres of mass, root axes
        const V m1 = m.c[LM_M1], m2 = m.c[LM_M2];

        // ---- motor body kinematics (replicated in every lane - unless the aux lanes have the motor body: SimOpts::aux)
        Vec3<V> am = ldv3(m, LM_AM), em = ldv3(m, LM_EM);
        Mat3<V> Rm;
        Vec3<V> cm;
        Sym3<V> Im;
        if (!o.aux) {
            Rm = rodrigues(em, sp, cp);
            cm = am + mul(Rm, ldv3(m, LM_DCM));
            Im = rotate(Rm, ldsym(m, LM_IM));
        }
        const V mm = m.c[LM_MM], m0 = m.c[LM_M0];
        Vec3<V> c0 = ldv3(m, LM_C0);
        Sym3<V> I0 = ldsym(m, LM_I0);

        // ---- contact candidates and contact-frame direction data (before the heavy dynamics, while the
        //      kinematic quantities are still live)
        if (o.contacts) {
            U own = zero_u<V>();            // the leg slots in which this lane's leg has a contact
            const Vec3<W> t_fl = both(a2) + mul(both(R12), pk(ldv3(m, LM_DFOOT), ldv3(m, LM_LC_D)));      // (foot centre, lower cylinder's centre)
            const Vec3<W> t_ax = mul(both(R12), pk(ldv3(m, LM_LC_AX), ldv3(m, LM_LC_XA)));               // (the cylinder's axis, its x axis)
            Vec3<V> foot = lo(t_fl);
            V fdist = (s.pz + dot(foot, nb) - m.c[LM_FOOT_R]) + s.pz_lo;
            MK fon = mand(lane_ok, lt(fdist, V(0)));
            live_slots |= cand_store(sc, live_slots, 0, foot - nb * (m.c[LM_FOOT_R] + fdist * V(0.5)), fdist, fon, &own);
            CylContacts<V> lc;
            MK all_on = lane_ok;
            cylinder_floor(hi(t_fl), lo(t_ax), hi(t_ax), m.c[LM_LC_R], m.c[LM_LC_H], nb, s.pz, s.pz_lo, all_on, lc);
            live_slots |= cand_store_cyl(sc, live_slots, 1, lc, all_on, &own);
            MK any_con = mor(fon, lc.on[0]);
            if (xtra) {
                // every remaining geom of the model against the floor
                CylContacts<V> cy;
                const Vec3<W> t_ut = both(a1) + mul(both(R1), pk(ldc3(m, LM_UC_D), ldc3(m, LM_DTIP)));          // (upper cylinder's centre, knee tip)
                const Vec3<W> t_ux = mul(both(R1), pk(ldc3(m, LM_UC_AX), ldc3(m, LM_UC_XA)));
                cylinder_floor(lo(t_ut), lo(t_ux), hi(t_ux), ldc(m, LM_UC_R), ldc(m, LM_UC_H), nb, s.pz, s.pz_lo, all_on, cy);
                live_slots |= cand_store_cyl(sc, live_slots, 5, cy, all_on, &own);
                any_con = mor(any_con, cy.on[0]);
                Vec3<V> tip = hi(t_ut);
                V tipd = (s.pz + dot(tip, nb) - ldc(m, LM_TIP_R)) + s.pz_lo;
                live_slots |= cand_store(sc, live_slots, 9, tip - nb * (ldc(m, LM_TIP_R) + tipd * V(0.5)), tipd, mand(lane_ok, lt(tipd, V(0))), &own);
                any_con = mor(any_con, mand(lane_ok, lt(tipd, V(0))));
              if (xbody) {
                if (o.aux) Rm = rodrigues(em, sp, cp);          // (the motor body's rotation for its geoms: rare path, computed where it is needed)
                // lane cylinder (root body: lane 2 screw1; motor body: lane 3 threadMass)
                MK x_onm = gt(ldc(m, LM_X_ONM), V(0.5));
                Vec3<V> xc_c = ldc3(m, LM_XC_C), xc_ax = ldc3(m, LM_XC_AX), xc_xa = ldc3(m, LM_XC_XA);
                cylinder_floor(sel_v3(x_onm, am + mul(Rm, xc_c - am), xc_c), sel_v3(x_onm, mul(Rm, xc_ax), xc_ax), sel_v3(x_onm, mul(Rm, xc_xa), xc_xa),
                               ldc(m, LM_XC_R), ldc(m, LM_XC_H), nb, s.pz, s.pz_lo, mand(lane_ok, gt(ldc(m, LM_XC_EN), V(0.5))), cy);
                live_slots |= cand_store_cyl(sc, live_slots, 10, cy, mnot(x_onm));      // (slots in increasing order: 10-13, 14, 15-22, then the motor body's 23-27)
                any_con = mor(any_con, cy.on[0]);
                Vec3<V> ellx_m = c0;
                V elld_m = V(1);
                MK ellon_m = lt(V(1), V(0));
                {   // lane ellipsoid: support point in direction -n.  A sphere of the largest semi-axis around its centre first - when no lane's
                    // reaches the floor (the usual case: a robot on its side rests on legs and screws) the rotation products are not needed
                    const Vec3<V> ec0 = ldc3(m, LM_XE_C), sz = ldc3(m, LM_XE_S);
                    const Vec3<V> ec = sel_v3(x_onm, am + mul(Rm, ec0 - am), ec0);
                    const MK een = mand(lane_ok, gt(ldc(m, LM_XE_EN), V(0.5)));
                    const V ech = (s.pz + dot(ec, nb)) + s.pz_lo;
                    if (any_lane(mand(een, lt(ech, vmax(sz.x, vmax(sz.y, sz.z)))))) {
                        Mat3<V> Re0, Re;
#pragma unroll
                        for (int i = 0; i < 9; i++) Re0.m[i] = ldc(m, LM_XE_R + i);
                        Mat3<V> Rem = mul(Rm, Re0);
#pragma unroll
                        for (int i = 0; i < 9; i++) Re.m[i] = sel(x_onm, Rem.m[i], Re0.m[i]);
                        Vec3<V> dl = mulT(Re, -nb);
                        V iden = vrsqrt(sz.x * sz.x * dl.x * dl.x + sz.y * sz.y * dl.y * dl.y + sz.z * sz.z * dl.z * dl.z);
                        Vec3<V> sup = ec + mul(Re, v3<V>(sz.x * sz.x * dl.x * iden, sz.y * sz.y * dl.y * iden, sz.z * sz.z * dl.z * iden));
                        V elld = (s.pz + dot(sup, nb)) + s.pz_lo;
                        MK ellon = mand(een, lt(elld, V(0)));
                        Vec3<V> ellx = sup - nb * (elld * V(0.5));
                        live_slots |= cand_store(sc, live_slots, 14, ellx, elld, mand(ellon, mnot(x_onm)));
                        ellx_m = ellx; elld_m = elld; ellon_m = mand(ellon, x_onm);
                        any_con = mor(any_con, ellon);
                    }
                }
                {   // lane box (root body): first 4 penetrating vertices in vertex order.  A vertex's height is the centre's plus / minus the three
                    // half-extents projected on the floor normal; the positions are worked out for penetrating vertices only, and nothing
                    // at all when the lowest vertex of every lane's box is above the floor
                    Mat3<V> Rb;
#pragma unroll
                    for (int i = 0; i < 9; i++) Rb.m[i] = ldc(m, LM_XB_R + i);
                    const Vec3<V> bc = ldc3(m, LM_XB_C), bs = ldc3(m, LM_XB_S);
                    const MK ben = mand(lane_ok, gt(ldc(m, LM_XB_EN), V(0.5)));
                    const Vec3<V> hb = mulT(Rb, nb);
                    const V hx = bs.x * hb.x, hy = bs.y * hb.y, hz = bs.z * hb.z;
                    const V bch = dot(bc, nb);
                    if (any_lane(mand(ben, lt((s.pz + (bch - (vabs(hx) + vabs(hy) + vabs(hz)))) + s.pz_lo, V(0))))) {
                        V cnt = V(0);
#pragma unroll
                        for (int vtx = 0; vtx < 8; vtx++) {
                            const V d = (s.pz + (bch + (((vtx & 1) ? hx : -hx) + ((vtx & 2) ? hy : -hy) + ((vtx & 4) ? hz : -hz)))) + s.pz_lo;
                            const MK on = mand(mand(ben, lt(d, V(0))), lt(cnt, V(3.5)));
                            cnt = cnt + sel(on, V(1), V(0));
                            if (any_lane(on)) {
                                const Vec3<V> l = v3<V>((vtx & 1) ? bs.x : -bs.x, (vtx & 2) ? bs.y : -bs.y, (vtx & 4) ? bs.z : -bs.z);
                                const Vec3<V> pnt = bc + mul(Rb, l);
                                live_slots |= cand_store(sc, live_slots, 15 + vtx, pnt - nb * (d * V(0.5)), d, on);
                            }
                            any_con = mor(any_con, on);
                        }
                    }
                }
                live_slots |= cand_store_cyl(sc, live_slots, 23, cy, x_onm);
                live_slots |= cand_store(sc, live_slots, 27, ellx_m, elld_m, ellon_m);
              }
            }
            if (PAIR) {
                const Vec3<W> t_uc = mul(both(R1), pk(ldc3(m, LM_UC_D), ldc3(m, LM_UC_AX)));
                const Vec3<V> uc = a1 + lo(t_uc), ua = hi(t_uc);
                const V uh = ldc(m, LM_UC_H);
                const Vec3<V> isv = ldc3(m, LM_PE_IS);
                // The thread pair first (slot 28: its values are dead before the mass pair's narrow phase, the register peak of this block): the motor-axis
                // thread against the own upper leg, flagged lanes only - the model builder marks the legs
                // the thread can come near at all (0.2 % of the reference's draws); broad phase: the two AXES within r_thread + r_leg + 0.2 mm.
                MK ton = lt(V(1), V(0));
                V tdist = V(1);
                Vec3<V> tpos = uc;
                // (the builder's flag covers shoulder angles up to 0.4 rad - rollouts stay below 0.15; a leg bent further takes the broad phase whatever the flag says)
                const MK tflag = mor(lt(isv.y, V(0)), gt(vabs(s.th1), V(0.4)));
                if (any_lane(tflag)) {
                    const Vec3<V> tcw = am + mul(Rm, ldc3(m, LM_PT_C) - am), taw = mul(Rm, ldc3(m, LM_PT_AX));
                    const V tr = ldc(m, LM_PT_R), thh = ldc(m, LM_PT_H), ur = ldc(m, LM_UC_R);
                    const MK tnear = mand(tflag, lt(segment_distance<V>(uc, ua, uh, tcw, taw, thh), tr + ur + V(2e-4)));
                    if (any_lane(tnear)) {
                        Vec3<V> tm;
                        thread_narrow<V>(tcw, taw, tr, thh, uc, ua, ur, uh, tdist, tm, tpos);
                        ton = mand(tnear, lt(tdist, V(0)));
                        if (any_lane(ton) && !sc.aux_lane) {
                            // MuJoCo's normal runs leg -> thread (geom1 = the leg's cylinder, the lower geom id) with rows jac(motor) - jac(leg):
                            // the rows of the mass pair with the frame (n, t1, t2) negated - stored as (m, -t1, m x t1) with m = -n the thread's
                            // outward normal (t1 is even in the normal's sign, t2 = n x t1 odd).
                            const Vec3<V> wy = v3<V>(R.m[3], R.m[4], R.m[5]);
                            const Vec3<V> ys = sel_v3(lt(vabs(dot(tm, wy)), V(0.5)), wy, nb);
                            Vec3<V> t1 = ys - tm * dot(tm, ys);
                            t1 = t1 * vrsqrt(vmax(dot(t1, t1), V(1e-30)));
                            const Vec3<V> t2 = cross(tm, t1);
                            const V fr[9] = {tm.x, tm.y, tm.z, -t1.x, -t1.y, -t1.z, t2.x, t2.y, t2.z};
#pragma unroll
                            for (int i = 0; i < 9; i++) sc.ovc[(sc.pd2 + i) * sc.ovc_stride] = fr[i];
                        }
                    }
                }
                live_slots |= cand_store(sc, live_slots, SLOT_THREAD, tpos, tdist, ton);
                any_con = mor(any_con, ton);
                // The geom-geom pair: the eccentric-mass ellipsoid (motor body) against the own upper-leg cylinder.  Broad phase: the
                // ellipsoid's centre within (largest semi-axis + cylinder radius) of the leg's axis segment; the narrow phase runs for
                // the whole wave when some lane is that close.
                const Vec3<V> pe = am + mul(Rm, ldc3(m, LM_PE_C) - am);
                Mat3<V> Re0;
#pragma unroll
                for (int i = 0; i < 9; i++) Re0.m[i] = ldc(m, LM_PE_R + i);
                const Mat3<V> Re = mul(Rm, Re0);
                // broad phase, exact for what it tests: does the axis segment enter the ellipsoid with semi-axes s + (r_cyl + 0.2 mm)?  In that
                // ellipsoid's unit-sphere coordinates the segment is still a segment: its closest point to the origin decides.
                const Vec3<V> cl = mulT(Re, uc - pe), ul = mulT(Re, ua);
                const V isvy = vabs(isv.y);          // (the sign carries the thread flag, below)
                const Vec3<V> cs = v3<V>(cl.x * isv.x, cl.y * isvy, cl.z * isv.z), us = v3<V>(ul.x * isv.x, ul.y * isvy, ul.z * isv.z);
                const V tpr = vmin(vmax(-dot(cs, us) * vrcp(dot(us, us)), -uh), uh);
                const Vec3<V> pcl = cs + us * tpr;
                const MK near_pair = mand(gt(isv.x, V(0)), lt(dot(pcl, pcl), V(1)));
                MK pon = lt(V(1), V(0));
                V pdist = V(1);
                Vec3<V> pn = nb, ppos = uc;
                if (any_lane(near_pair)) {
                    // warm start from the previous substep's solution where this lane has one (sc.pd + 9 .. 11), the cold fixed-count scheme
                    // for the lanes that have none or did not converge from it
                    V wt = sc.ld(sc.pd + 9), wlam = sc.ld(sc.pd + 10);
                    MK warm_ok = mand(near_pair, gt(sc.ld(sc.pd + 11), V(0.5)));
                    if (any_lane(warm_ok)) {
                        const MK conv = pair_narrow_warm<V>(pe, Re, ldc3(m, LM_PE_S), uc, ua, ldc(m, LM_UC_R), uh, wt, wlam, pdist, pn, ppos);
                        warm_ok = mand(warm_ok, conv);
                    }
#if !defined(__HIPCC__)
                    if (sc.grp == 0) g_pair_narrow_stats[any_lane(mand(near_pair, mnot(warm_ok))) ? 1 : 0]++;
#endif
                    if (any_lane(mand(near_pair, mnot(warm_ok)))) {
                        V cdist, ct, clam;
                        Vec3<V> cn, cpos;
                        pair_narrow<V>(pe, Re, ldc3(m, LM_PE_S), uc, ua, ldc(m, LM_UC_R), uh, cdist, cn, cpos, &ct, &clam);
                        pdist = sel(warm_ok, pdist, cdist); pn = sel_v3(warm_ok, pn, cn); ppos = sel_v3(warm_ok, ppos, cpos);
                        wt = sel(warm_ok, wt, ct); wlam = sel(warm_ok, wlam, clam);
                    }
                    sc.st(sc.pd + 9, wt); sc.st(sc.pd + 10, wlam);
                    pon = mand(near_pair, lt(pdist, V(0)));
                }
                if (any_lane(pon)) {
                    // contact frame: MuJoCo's mju_makeFrame on the WORLD normal (y = world y unless the normal is within 60 degrees of
                    // it, then world z), expressed in root coordinates like everything else.  Normal mass -> leg (geom1 = the ellipsoid),
                    // rows jac(leg) - jac(motor).
                    const Vec3<V> wy = v3<V>(R.m[3], R.m[4], R.m[5]);
                    const V ny = dot(pn, wy);
                    const Vec3<V> ys = sel_v3(lt(vabs(ny), V(0.5)), wy, nb);
                    Vec3<V> t1 = ys - pn * dot(pn, ys);
                    t1 = t1 * vrsqrt(vmax(dot(t1, t1), V(1e-30)));
                    sc.st3(sc.pd, pn); sc.st3(sc.pd + 3, t1); sc.st3(sc.pd + 6, cross(pn, t1));
                }
                sc.st(sc.pd + 11, sel(near_pair, V(1), V(0)));          // the next substep may start from this one's solution (only while the pair stays near)
                live_slots |= cand_store(sc, live_slots, SLOT_MASS, ppos, pdist, pon);
                any_con = mor(any_con, pon);
            }
            any_contact = any_lane(any_con);
            env_con = neq_u(quad_sum_u(mbit(any_con)), zero_u<V>());
            if (o.aux) {      // the aux lanes integrate the env's root state too (phase C): they need to know whether THEIR env has a contact
                const U ec = xor_get_any_u(mbit(env_con), axoff);
                if (axl) env_con = neq_u(ec, zero_u<V>());
            }
            if (any_contact && !axl) sc.st(SC_OWN, utov(own, (const V*)nullptr));
            if (any_contact && !axl) {
                // direction data for the contact frame (n, t1, t2) = R^T (ez, ey, -ex)
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    Vec3<V> d = (k == 0) ? nb : (k == 1) ? v3<V>(R.m[3], R.m[4], R.m[5]) : v3<V>(-R.m[0], -R.m[1], -R.m[2]);
                    sc.st3(SC_DD + 3 * k, d);
                    sc.st(SC_DD + 21 + k, dot(d, u));
                }
                sc.st3(SC_DD + 9, e1); sc.st3(SC_DD + 12, a1); sc.st3(SC_DD + 15, e2); sc.st3(SC_DD + 18, a2);
            }
        }
        JB_SCHED_FENCE();

        // ---- composite rigid bodies -> star system of M (into the scratch)
        const W mw = W(m1, m2);
        const Vec3<W> hw = ccw * mw;                                  // first moments (upper, lower)
        const Sym3<W> Jo = about_origin(Iw, mw, ccw);                 // inertias about the root origin (upper, lower)
        const Sym3<V> J2 = hi(Jo);
        const Sym3<V> J12 = lo(Jo) + J2;
        const Vec3<V> h2 = hi(hw), h12 = lo(hw) + h2;
        const V m12 = m1 + m2;
        // the joint motion vectors and what each hinge moves: the shoulder both bodies (m12, h12, J12), the knee the lower one
        const Vec3<W> sw = cross(aw, ew);                             // linear part of the joint motion vectors (shoulder, knee)
        const Vec3<W> hcw = pk(h12, h2);
        const Vec3<W> fw = sw * W(m12, m2) + cross(ew, hcw), nw = mul(pk(J12, J2), ew) + cross(hcw, sw);
        const Vec3<V> fS = lo(fw), fK = hi(fw), nS = lo(nw), nK = hi(nw);
        // (the solver's layout: rows in pairs, one column per hinge - a 2 x 2 transposition of the (shoulder, knee) pairs)
        sys.B[0][0] = W(nS.x, nS.y); sys.B[1][0] = W(nS.z, fS.x); sys.B[2][0] = W(fS.y, fS.z);
        sys.B[0][1] = W(nK.x, nK.y); sys.B[1][1] = W(nK.z, fK.x); sys.B[2][1] = W(fK.y, fK.z);
        {
            const W cd = dot(ew, nw) + dot(sw, fw);
            sys.C[0] = pk_lo(cd); sys.C[2] = pk_hi(cd);
            sys.C[1] = dot(e1, nK) + dot(lo(sw), fK);
        }
        // the legs' sums over the quad; in the aux quads the same instructions sum the motor body and the root body's own mass
        const Vec3<V> h12q = qsum(h12);
        const Sym3<V> J12q = qsum(J12);
        Vec3<V> ht;
        Sym3<V> Jt;
        if (o.aux) {
            // hand-over 1 of 2 (group 0 <- 2, group 1 <- 3): the aux quad's sums, and its lane 0's shoulder column = the motor's column
            ht = v3<V>(xor_get(h12q.x, axoff), xor_get(h12q.y, axoff), xor_get(h12q.z, axoff)) + h12q;
            Sym3<V> Jx;
            Jx.xx = xor_get(J12q.xx, axoff); Jx.yy = xor_get(J12q.yy, axoff); Jx.zz = xor_get(J12q.zz, axoff);
            Jx.xy = xor_get(J12q.xy, axoff); Jx.xz = xor_get(J12q.xz, axoff); Jx.yz = xor_get(J12q.yz, axoff);
            Jt = Jx + J12q;
#pragma unroll
            for (int ip = 0; ip < 3; ip++) sys.Bm[ip] = W(xor_get(quad_bcast<0>(pk_lo(sys.B[ip][0])), axoff), xor_get(quad_bcast<0>(pk_hi(sys.B[ip][0])), axoff));
            sys.Cm = xor_get(quad_bcast<0>(sys.C[0]), axoff);
        } else {
            Vec3<V> sM = cross(am, em);
            Vec3<V> hm = cm * mm;
            Sym3<V> Jm = about_origin(Im, mm, cm);
            Vec3<V> fM = sM * mm + cross(em, hm), nM = mul(Jm, em) + cross(hm, sM);
            sys.Bm[0] = W(nM.x, nM.y); sys.Bm[1] = W(nM.z, fM.x); sys.Bm[2] = W(fM.y, fM.z);
            sys.Cm = dot(em, nM) + dot(sM, fM);
            ht = c0 * m0 + hm + h12q;
            Jt = about_origin(I0, m0, c0) + Jm + J12q;
        }
        root_block<V>(Jt, ht, m.c[LM_MTOT], sys.A);
        JB_SCHED_FENCE();

        // ---- bias forces (Newton-Euler with qacc = 0 in MuJoCo coordinates) and applied forces
        {
            const V w_2 = dot(w, w);
            auto accel_at = [&](const Vec3<V>& x) { return AO + wxwx(w, w_2, x); };
            // own leg.  Angular velocities, angular accelerations and the hinge anchors' linear accelerations are a short chain (the lower body
            // rides on the upper one); forces and moments of the two bodies are then the same formulas on pairs.
            const Vec3<V> Aa1 = accel_at(a1);
            const Vec3<V> w1 = w + e1 * s.thd1, w2 = w1 + e2 * s.thd2;
            const Vec3<W> www = pk(w1, w2);
            const Vec3<W> dal = cross(pk(w, w1), ew) * W(s.thd1, s.thd2);      // (al1, al2 - al1)
            const Vec3<V> al1 = lo(dal), al2 = al1 + hi(dal);
            const Vec3<W> alw = pk(al1, al2);
            const Vec3<W> rw = ccw - aw;                                       // centres of mass from the own hinge anchors
            const W w2w = dot(www, www);
            const Vec3<V> r12 = a2 - a1;
            const Vec3<V> Aa2 = Aa1 + cross(al1, r12) + wxwx(w1, pk_lo(w2w), r12);
            const Vec3<W> Fw = (pk(Aa1, Aa2) + cross(alw, rw) + wxwx(www, w2w, rw)) * mw;
            const Vec3<W> Nw = mul(Iw, alw) + cross(www, mul(Iw, www));
            const Vec3<W> tw = Nw + cross(rw, Fw);                             // moments about the own hinge anchors
            const Vec3<V> F2 = hi(Fw);
            V cK = dot(e2, hi(tw));
            V cS = dot(e1, lo(tw) + hi(Nw) + cross(cc2 - a1, F2));
            Vec3<V> legF = lo(Fw) + F2;
            const Vec3<W> Now = Nw + cross(ccw, Fw);                           // moments about the root origin
            Vec3<V> legN = lo(Now) + hi(Now);
            const Vec3<V> legFq = qsum(legF), legNq = qsum(legN);
            Vec3<V> bl, ba;
            V cM;
            if (o.aux) {
                // hand-over 2 of 2: the aux quad's force / moment sums (motor body + root body) and its lane 0's shoulder bias = the motor's
                bl = v3<V>(xor_get(legFq.x, axoff), xor_get(legFq.y, axoff), xor_get(legFq.z, axoff)) + legFq;
                ba = v3<V>(xor_get(legNq.x, axoff), xor_get(legNq.y, axoff), xor_get(legNq.z, axoff)) + legNq;
                cM = xor_get(quad_bcast<0>(cS), axoff);
            } else {
                Vec3<V> F0 = accel_at(c0) * m0;
                Vec3<V> N0 = cross(w, mul(I0, w));
                // motor body
                Vec3<V> wm = w + em * s.phid;
                Vec3<V> alm = cross(w, em) * s.phid;
                Vec3<V> rm = cm - am;
                Vec3<V> Fm = (accel_at(am) + cross(alm, rm) + wxwx(wm, dot(wm, wm), rm)) * mm;
                Vec3<V> Nm = mul(Im, alm) + cross(wm, mul(Im, wm));
                cM = dot(em, Nm + cross(rm, Fm));
                bl = F0 + Fm + legFq;
                ba = N0 + cross(c0, F0) + Nm + cross(cm, Fm) + legNq;
            }
            sys.tr[0] = W(-ba.x, -ba.y); sys.tr[1] = W(-ba.z, -bl.x); sys.tr[2] = W(-bl.y, -bl.z);
            sys.tl[0] = -cS - m.c[LM_K1] * s.th1 - m.c[LM_B1] * s.thd1;
            sys.tl[1] = -cK - m.c[LM_K2] * s.th2 - m.c[LM_B2] * s.thd2;
            V uc = vmin(vmax(ctrl, m.c[LM_CTRL_LO]), m.c[LM_CTRL_HI]);
            V gear = m.c[LM_GEAR];
            V len = gear * (s.phi + s.turns * V(6.283185307179586));
            V force = m.c[LM_GAIN] * uc + m.c[LM_BIAS] + m.c[LM_BIAS + 1] * len + m.c[LM_BIAS + 2] * gear * s.phid;
            sys.tm = -cM + gear * force;
        }
    }
    JB_SCHED_FENCE();
    if (o.lean && is_main) sys_store(sc, sys);

    JB_PROF_ADD(o, 0);
    // the helper groups learn what phase A found (values of the first lane, a main lane)
    any_contact = wave_bcast_u(any_contact ? 1u : 0u) != 0u;
    live_slots = wave_bcast_u(live_slots);
    wave_sync();          // phase A's scratch (SC_OWN, SC_ST, SC_DD, the candidate rows: written under `if (rep)`) -> read by every group below
    const SlotPlan plan = make_slot_plan(sc, live_slots);
    const SpreadPlan<V> spl = make_spread_plan<V>(sc, plan, o.spread != 0 && any_contact);

    // ================= phase B: contact solve (primal Newton on the active set) and final acceleration: newton_phase above.
    // The y-independent rows of the live contacts are built ONCE per substep; the solve may run twice.
    Vec3<V> dk[3];
    if (any_contact) {
#pragma unroll
        for (int k = 0; k < 3; k++) dk[k] = sc.ld3(SC_DD + 3 * k);      // contact-frame directions: once per substep, every lane
        contact_rows_build_all<V, PAIR>(m, sc, xtra, plan, spl);
        JB_PROF_ADD(o, 5);
    }
    Pk2<V> yrp[3];
    V yl[2], ym;
    MK capped = lt(V(1), V(0));        // main lanes: this env's iteration ran into the cap
    bool redone = newton_phase<V, PAIR, false>(m, sc, s, o, xtra, plan, spl, Rw, sys, dk, any_contact, env_con, yrp, yl, ym, capped);
#ifdef JB_NO_RESOLVE      // A/B measurements only: the hot instantiation alone (a capped substep keeps its last iterate and is not counted)
    redone = false;
#endif
    // The plain active-set iteration takes full Newton steps and can cycle between sets (robots lying on their side, a dozen contacts on
    // the body geoms: a few substeps in ten million).  A wave in which some env hit the cap solves THIS substep's contact problem again the
    // way MuJoCo's Newton solver does - every step to the exact minimum of the cost along the Newton direction (newton_phase<LS = true>) - from
    // the same warm start; the substep's entry state is untouched until phase C, and phase A's system, plan and rows are still there.
    // Only the envs that hit the cap take the second solve's accelerations: their decision and their arithmetic are their own, their
    // wave-mates keep the bits of the first solve.  The code sits in a cold block outside the Newton loop: the other substeps - all but a
    // few in ten million - pay one scalar branch.
    if (__builtin_expect(redone, false)) {
        JB_ASM_MARK("jb-cold-solve-begin");          // (a comment in the ISA: tools/asm_spills.py tells the cold block's register spills from the hot loop's)
        const U cf = group_sum_u<V>(sc, mbit(capped));      // every lane group learns which envs it concerns (helper lanes contribute zeros)
        Pk2<V> zr[3];
        V zl[2], zm;
        MK capped2 = lt(V(1), V(0));
        (void)newton_phase<V, PAIR, true>(m, sc, s, o, xtra, plan, spl, Rw, sys, dk, any_contact, env_con, zr, zl, zm, capped2);
        const U cf2 = group_sum_u<V>(sc, mbit(capped2));
        const MK re = neq_u(cf, zero_u<V>());
#pragma unroll
        for (int ip = 0; ip < 3; ip++) yrp[ip] = selw(re, zr[ip], yrp[ip]);
        yl[0] = sel(re, zl[0], yl[0]); yl[1] = sel(re, zl[1], yl[1]); ym = sel(re, zm, ym);
        // the failure counter records what even the line-searched solve did not settle (LEAN: the state is parked in the scratch)
        const V inc = sel(mand(re, neq_u(cf2, zero_u<V>())), V(1), V(0));
        if (o.lean) { if (is_main) sc.st(SC_LSTATE + 34, sc.ld(SC_LSTATE + 34) + inc); }
        else s.fail = s.fail + inc;
#if !defined(__HIPCC__)
        if (is_main) g_ls_stats[0]++;
#endif
        JB_ASM_MARK("jb-cold-solve-end");
#if defined(JB_CAPTURE) && defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
        if (o.capture && is_main) {      // the substep's entry state of an env that stayed unconverged (LEAN: not supported by this diagnostic)
            const bool bad = mand(re, neq_u(cf2, zero_u<V>()));
            unsigned slot = 0u;
            if (bad && (threadIdx.x & 3) == 0) slot = atomicAdd(o.capture_count, 1u);
            slot = quad_bcast_u<0>(slot);
            if (bad && slot < (unsigned)JB_CAPTURE_SLOTS) {
                float* r = o.capture + (size_t)slot * 64;
                const int leg = threadIdx.x & 3;
                if (leg == 0) {
                    const float root[28] = {ctrl, s.px, s.py, s.pz, s.qw, s.qx, s.qy, s.qz, s.pz_lo, s.qw_lo, s.qx_lo, s.qy_lo, s.qz_lo, s.vx, s.vy, s.vz, s.wx, s.wy, s.wz, s.phi, s.phid, s.turns,
                                            s.wa[0], s.wa[1], s.wa[2], s.wl[0], s.wl[1], s.wl[2]};
                    for (int i = 0; i < 28; i++) r[i] = root[i];
                    r[28] = s.wm; r[29] = (float)xtra; r[30] = (float)__builtin_popcount(plan.live); r[31] = (float)plan.live;
                }
                r[32 + 6 * leg + 0] = s.th1; r[32 + 6 * leg + 1] = s.th2; r[32 + 6 * leg + 2] = s.thd1; r[32 + 6 * leg + 3] = s.thd2; r[32 + 6 * leg + 4] = s.wj[0]; r[32 + 6 * leg + 5] = s.wj[1];
            }
        }
#endif
    }
    JB_SCHED_FENCE();
    if (!rep && !axl) return redone;

    // ================= phase C: integrate (the replica too: it starts the next substep from the same state)
    if (o.lean) state_load(sc, s);
    const V yr[6] = {pk_lo(yrp[0]), pk_hi(yrp[0]), pk_lo(yrp[1]), pk_hi(yrp[1]), pk_lo(yrp[2]), pk_hi(yrp[2])};
    Vec3<V> lin = mul(Rw, v3<V>(yr[3], yr[4], yr[5]));
    s.wa[0] = yr[0]; s.wa[1] = yr[1]; s.wa[2] = yr[2]; s.wl[0] = lin.x; s.wl[1] = lin.y; s.wl[2] = lin.z; s.wj[0] = yl[0]; s.wj[1] = yl[1]; s.wm = ym;
    // mj_advance: velocities, then positions with the new velocities
    s.wx = s.wx + h * yr[0]; s.wy = s.wy + h * yr[1]; s.wz = s.wz + h * yr[2];
    s.vx = s.vx + h * lin.x; s.vy = s.vy + h * lin.y; s.vz = s.vz + h * lin.z;
    s.thd1 = s.thd1 + h * yl[0]; s.thd2 = s.thd2 + h * yl[1]; s.phid = s.phid + h * ym;
    s.px = s.px + h * s.vx; s.py = s.py + h * s.vy;
    comp_add(s.pz, s.pz_lo, h * s.vz, true);        // the height (~35 mm) dwarfs a substep's travel (< 0.1 mm)
    {   // q <- q * exp(h w / 2), as  q + [q (cos a - 1) + (q x-terms) sin(a)/|w|]: the bracket is ~1e-3 |q|, so computing it in
        // fp32 and adding it to the hi/lo pair with a compensated sum keeps the quaternion to ~1e-11 over a control step
        V wn2 = s.wx * s.wx + s.wy * s.wy + s.wz * s.wz;
        V half = V(0.5) * h;
        // sin(a)/|w| and cos(a) - 1 with a = h|w|/2 (tiny): series in a^2
        V a2_ = half * half * wn2;
        V sc_ = half * (V(1) + a2_ * (V(-1.0 / 6) + a2_ * (V(1.0 / 120) + a2_ * V(-1.0 / 5040))));
        V cm1 = a2_ * (V(-0.5) + a2_ * (V(1.0 / 24) + a2_ * V(-1.0 / 720)));
        V dx = s.wx * sc_, dy = s.wy * sc_, dz = s.wz * sc_;
        // the four components as two pairs (w, x) (y, z):
        //   d0 = qw cm1 - qx dx - qy dy - qz dz      d1 = qw dx + qx cm1 + qy dz - qz dy
        //   d2 = qw dy - qx dz + qy cm1 + qz dx      d3 = qw dz + qx dy - qy dx + qz cm1
        using W = Pk2<V>;
        W q01 = W(s.qw, s.qx), q23 = W(s.qy, s.qz), l01 = W(s.qw_lo, s.qx_lo), l23 = W(s.qy_lo, s.qz_lo);
        const W d01 = W(s.qw) * W(cm1, dx) + W(s.qx) * W(-dx, cm1) + W(s.qy) * W(-dy, dz) + W(s.qz) * W(-dz, -dy);
        const W d23 = W(s.qw) * W(dy, dz) + W(s.qx) * W(-dz, dy) + W(s.qy) * W(cm1, -dx) + W(s.qz) * W(dx, cm1);
        comp_add2(q01, l01, d01); comp_add2(q23, l23, d23);
        // (no renormalisation here: the update preserves |q| up to the rounding of the bracket, ~1e-10 per substep; mj_kinematics'
        //  per-step normalisation is applied once per control step by normalise_state() - the difference is below 1e-8 in |q|)
        s.qw = pk_lo(q01); s.qx = pk_hi(q01); s.qy = pk_lo(q23); s.qz = pk_hi(q23);
        s.qw_lo = pk_lo(l01); s.qx_lo = pk_hi(l01); s.qy_lo = pk_lo(l23); s.qz_lo = pk_hi(l23);
    }
    s.th1 = s.th1 + h * s.thd1; s.th2 = s.th2 + h * s.thd2;
    {
        V pp = s.phi + h * s.phid;
        V k = vfloor((pp + V(3.141592653589793)) * V(0.15915494309189535));
        s.phi = pp - k * V(6.283185307179586);
        s.turns = s.turns + k;
    }
    if (o.lean) state_store(sc, s);
    JB_PROF_ADD(o, 4);
    return redone;      // (wave-uniform: the caller may count the substeps that were solved twice)
}

// mj_kinematics normalises the free-joint quaternion; here once when a state enters the simulator (kernel start, host harness)
template <typename V> JB_HD void normalise_state(LaneState<V>& s) {
    V qn = vrsqrt(s.qw * s.qw + s.qx * s.qx + s.qy * s.qy + s.qz * s.qz);       // bring |q| to 1 +- 1e-6 first (an imported state may be far off)
    V qh[4] = {s.qw * qn, s.qx * qn, s.qy * qn, s.qz * qn}, ql[4] = {s.qw_lo * qn, s.qx_lo * qn, s.qy_lo * qn, s.qz_lo * qn};
    quat_normalise_comp(qh, ql);
    quat_normalise_comp(qh, ql);
    s.qw = qh[0]; s.qx = qh[1]; s.qy = qh[2]; s.qz = qh[3]; s.qw_lo = ql[0]; s.qx_lo = ql[1]; s.qy_lo = ql[2]; s.qz_lo = ql[3];
}

// One physics substep.  A wave-uniform broadphase decides whether only the foot sphere + lower-leg cylinder can
// touch the floor (common) or every geom of the model has to be tested (rare).  Returns (wave-uniform) whether the contact solve ran a
// second time with the exact line search (newton_phase).
template <typename V, bool PAIR = false>
JB_HD bool substep(const LaneModel<V>& m, const LaneScratch<V>& sc, LaneState<V>& s, const V& ctrl, const SimOpts& o) {
    unsigned xt = 0;          // bit 0: some upper leg may touch the floor, bit 1: some root / motor-body geom may
    Mat3<V> Rw;               // root rotation (main lanes): mj_kinematics normalises the quaternion first
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
    if (m.c.split) restage_overlay(m.c);                // split tables: phase A's per-lane constants come back into the (now dead) SC_SYS / SC_FAC scratch
#endif
    if (o.lean && sc.grp == 0) state_load(sc, s);       // LEAN: the state lives in the scratch between substeps
    const bool rep = sc.grp == 0 || (o.offload && sc.grp == 1);      // main lanes and their replica (SimOpts::offload)
    if (rep || (o.aux && sc.grp >= 2)) Rw = quat2mat(s.qw, s.qx, s.qy, s.qz);      // (every lane that holds the root state) the quaternion is kept normalised: normalise_state() once per kernel, phase C after every substep
    if (o.contacts && rep) {
        const Vec3<V> nb = v3<V>(Rw.m[6], Rw.m[7], Rw.m[8]);         // floor normal in root coordinates
        // upper leg: sphere around the upper cylinder (+ slack for the shoulder angle)
        auto near_leg = mor(lt(s.pz + dot(ldv3(m, LM_BS_LEG_C), nb), m.c[LM_BS_LEG_R]), gt(vabs(s.th1), V(0.3)));
        // the lane's root / motor-body geoms: a bounding sphere first (at rest it clears the floor by 8 mm or more), and only
        // when some lane's sphere reaches the floor the two oriented boxes (support function of a box along -n)
        auto near_body = lt(s.pz + dot(ldv3(m, LM_BS_BODY_C), nb), m.c[LM_BS_BODY_R]);
        if (any_lane(near_body)) {
            near_body = lt(V(1), V(0));
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int b = LM_BX + 15 * k;
                Vec3<V> ax0 = ldv3(m, b + 3), ax1 = ldv3(m, b + 6), ax2 = ldv3(m, b + 9);
                V sup = vabs(dot(ax0, nb)) * m.c[b + 12] + vabs(dot(ax1, nb)) * m.c[b + 13] + vabs(dot(ax2, nb)) * m.c[b + 14];
                near_body = mor(near_body, lt(s.pz + dot(ldv3(m, b), nb), sup));
            }
        }
        xt = (any_lane(near_leg) ? 1u : 0u) | (any_lane(near_body) ? 2u : 0u);
    }
    xt = wave_bcast_u(xt);      // helper groups follow the main lanes' decision
    bool xtra = xt != 0u;
    const bool xbody = (xt & 2u) != 0u;
#ifdef JB_NO_XTRA
    xtra = false;
#endif
#ifdef JB_WAVE_STATS
    if (xtra && sc.grp == 0) s.st_xtra = s.st_xtra + V(1);
#endif
    return substep_impl<V, PAIR>(m, sc, s, ctrl, o, xtra, xbody, Rw);
}

}  // namespace jb
