// jb_lane.hpp — lane-value abstraction for the Jitterbug simulator.
//
// The simulator maps ONE ENVIRONMENT TO FOUR LANES (a "quad"): lane l of the quad
// owns leg l (reference: jitterbug.xml:52-103, four structurally identical
// two-hinge legs); root-body and motor quantities are replicated in the four
// lanes; sums over legs are quad all-reductions.  The math in jb_sim.hpp is
// written once against the small vocabulary below:
//
//   device  (hipcc, gfx950):  V = float, mask = bool, quad_sum = 2 DPP quad_perm adds,
//                             any_lane = wave ballot.
//   host    (g++, tests only): V = Quad<T> (4 components = the 4 lanes), which lets
//                             tests/ run the very same source in fp32/fp64 against
//                             the oracle without a GPU.  The host instantiation is
//                             test infrastructure, not a product path.
#pragma once
#include <cmath>
#include <cstdint>
#include <type_traits>
#if !defined(__HIPCC__)
#include <atomic>
#include <thread>
#endif

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define JB_HD __host__ __device__ __forceinline__
#define JB_D __device__ __forceinline__
#else
#define JB_HD inline
#define JB_D inline
#endif

namespace jb {

// ----------------------------------------------------------------------------- scalar lanes (device)
#if defined(__HIPCC__)
JB_D float dpp_quad_xor1(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, false));  // quad_perm [1,0,3,2]
}
JB_D float dpp_quad_xor2(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
}
JB_D float quad_sum(float x) {
    x += dpp_quad_xor1(x);
    x += dpp_quad_xor2(x);
    return x;
}
JB_D unsigned quad_sum_u(unsigned x) {
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);
    x += (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);
    return x;
}
JB_D bool any_lane(bool m) { return __builtin_amdgcn_ballot_w64(m) != 0ull; }
// helper groups: lanes L, L^off hold the same (env, leg) in different slot groups
// xor_sum(x, off) = x + x[lane ^ off] without an LDS round trip: gfx950's v_permlane{16,32}_swap for off 16 / 32, DPP
// row rotations inside a 16-lane row.  `sym2` promises that x already equals x[lane ^ 2*off] (the previous butterfly
// stage), which makes a rotation by 4 as good as the xor.  `off` is a compile-time constant after inlining.
JB_D unsigned xor_sum_bits(unsigned u, int off, bool sym2, bool is_float) {
    auto add = [&](unsigned a, unsigned b) { return is_float ? __builtin_bit_cast(unsigned, __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b)) : a + b; };
    if (off == 32) { auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); return add(r[0], r[1]); }
    if (off == 16) { auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); return add(r[0], r[1]); }
    if (off == 8) return add(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xF, 0xF, false));             // row_ror:8
    if (off == 4 && sym2) return add(u, (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x124, 0xF, 0xF, false));     // row_ror:4
    return add(u, (unsigned)__shfl_xor((int)u, off, 64));
}
// Transposed sum over 4 helper groups that sit in the four 16-lane rows of the wave: row g returns the total of v[g],
// associated (g0 + g2) + (g1 + g3) like the butterfly of xor_sum (high group bit first).  Three swaps and three adds for
// four values, no copies: v_permlane32_swap exchanges the upper half of its first operand with the lower half of the second,
// v_permlane16_swap the odd rows of the first with the even rows of the second.
JB_D float row_transpose_sum(float v0, float v1, float v2, float v3) {
    auto f = [](unsigned u) { return __builtin_bit_cast(float, u); };
    auto b = [](float x) { return __builtin_bit_cast(unsigned, x); };
    auto r02 = __builtin_amdgcn_permlane32_swap(b(v0), b(v2), false, false);
    auto r13 = __builtin_amdgcn_permlane32_swap(b(v1), b(v3), false, false);
    float s02 = f(r02[0]) + f(r02[1]), s13 = f(r13[0]) + f(r13[1]);
    auto r = __builtin_amdgcn_permlane16_swap(b(s02), b(s13), false, false);
    return f(r[0]) + f(r[1]);
}
// The value lane L ^ off holds (off = 32, 16 or 8: the distance to the lane group two groups on).  Exact for the lanes of the LOWER
// group of each pair (what jb_sim.hpp's aux bodies need: groups 0 / 1 read groups 2 / 3); the upper group's result is unspecified.
JB_D unsigned xor_get_bits(unsigned u, int off) {
    if (off == 32) { auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); return r[1]; }      // (vsrc's lower half receives vdst's upper half)
    if (off == 16) { auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); return r[1]; }      // (vsrc's even rows receive vdst's odd rows)
    if (off == 8) return (unsigned)__builtin_amdgcn_update_dpp(0, (int)u, 0x128, 0xF, 0xF, false);      // row_ror:8
    return (unsigned)__shfl_xor((int)u, off, 64);
}
// ... exact for every lane (two more integer operations)
JB_D unsigned xor_get_any_u(unsigned u, int off) {
    if (off == 32) { auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); return r[0] ^ r[1] ^ u; }
    if (off == 16) { auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); return r[0] ^ r[1] ^ u; }
    return xor_get_bits(u, off);
}
JB_D float xor_get(float x, int off) { return __builtin_bit_cast(float, xor_get_bits(__builtin_bit_cast(unsigned, x), off)); }
JB_D unsigned xor_get_u(unsigned x, int off) { return xor_get_bits(x, off); }
template <int J> JB_D float quad_bcast(float x) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), J * 0x55, 0xF, 0xF, false)); }
JB_D float xor_sum(float x, int off, bool sym2) { return __builtin_bit_cast(float, xor_sum_bits(__builtin_bit_cast(unsigned, x), off, sym2, true)); }
JB_D unsigned xor_sum_u(unsigned x, int off, bool sym2) { return xor_sum_bits(x, off, sym2, false); }
JB_D unsigned wave_bcast_u(unsigned x) { return (unsigned)__builtin_amdgcn_readfirstlane((int)x); }   // value of the first active lane
// Hand-over point between lane groups through the scratch (one group wrote, another reads).  The groups are lanes of ONE wave and
// LDS operations of a wave complete in program order, so the hardware needs nothing here - but the COMPILER must not move a read of
// the hand-over across the writes it depends on (it sees one thread and may prove the addresses different once the group is known):
// a wave barrier is that fence and emits no instruction.  The host emulation (one thread per group) needs a real barrier.
// The wave barrier alone is a scheduling fence (IntrNoMem + side effects): it stops the machine scheduler, but IR-level passes (GVN / PRE load
// forwarding) may still move or forward LDS / global accesses across it.  The wavefront-scope release / acquire fences around it are
// what makes the hand-over a memory ordering point for the optimiser too; at wavefront scope they emit no instruction on gfx9.
JB_D void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// ---- quad-level exchanges for the spread contact sweeps (jb_sim.hpp): a lane may work on a contact of ANOTHER leg of its env
// quad_rot<K>: the value of lane (l + K) & 3 of the quad; quad_bcast_u: the value of lane j of the quad (DPP quad_perm, one instruction)
template <int K> JB_D float quad_rot(float x) {
    constexpr int ctrl = K == 1 ? 0x39 : K == 2 ? 0x4E : 0x93;      // quad_perm [1,2,3,0] / [2,3,0,1] / [3,0,1,2]
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), ctrl, 0xF, 0xF, false));
}
template <int K> JB_D unsigned quad_rot_u(unsigned x) {
    constexpr int ctrl = K == 1 ? 0x39 : K == 2 ? 0x4E : 0x93;
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, ctrl, 0xF, 0xF, false);
}
template <int J> JB_D unsigned quad_bcast_u(unsigned x) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, J * 0x55, 0xF, 0xF, false); }
JB_D unsigned quad_lane_id(const float*) { return threadIdx.x & 3u; }
// element idx of the scratch column of lane `src` of the own quad (`me`: the own position in the quad)
JB_D float ld_leg(const float* p, int stride, unsigned idx, unsigned src, unsigned me) { return p[(int)(idx * (unsigned)stride) + (int)src - (int)me]; }
JB_D void st_leg(float* p, int stride, unsigned idx, unsigned src, unsigned me, float v, bool on) { if (on) p[(int)(idx * (unsigned)stride) + (int)src - (int)me] = v; }
// the value lane `src` of the quad holds
JB_D float quad_pick(float x, unsigned src) {
    const int xi = __builtin_bit_cast(int, x);
    const float x0 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x00, 0xF, 0xF, false)), x1 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0x55, 0xF, 0xF, false));
    const float x2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0xAA, 0xF, 0xF, false)), x3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, xi, 0xFF, 0xF, 0xF, false));
    return src == 0u ? x0 : src == 1u ? x1 : src == 2u ? x2 : x3;
}
#endif

JB_HD float sel(bool m, float a, float b) { return m ? a : b; }
JB_HD double sel(bool m, double a, double b) { return m ? a : b; }
JB_HD unsigned selu(bool m, unsigned a, unsigned b) { return m ? a : b; }
JB_HD bool lt(float a, float b) { return a < b; }
JB_HD bool lt(double a, double b) { return a < b; }
JB_HD bool gt(float a, float b) { return a > b; }
JB_HD bool gt(double a, double b) { return a > b; }
JB_HD bool mand(bool a, bool b) { return a && b; }
JB_HD bool mor(bool a, bool b) { return a || b; }
JB_HD bool mnot(bool a) { return !a; }
JB_HD unsigned mbit(bool a) { return a ? 1u : 0u; }
JB_HD unsigned popc_u(unsigned a) { return (unsigned)__builtin_popcount(a); }
JB_HD unsigned xor_u(unsigned a, unsigned b) { return a ^ b; }
JB_HD bool eq_u(unsigned a, unsigned b) { return a == b; }
JB_HD bool lt_u(unsigned a, unsigned b) { return a < b; }
JB_HD unsigned or_u(unsigned a, unsigned b) { return a | b; }
JB_HD unsigned and_u(unsigned a, unsigned b) { return a & b; }
JB_HD unsigned sub_u(unsigned a, unsigned b) { return a - b; }
JB_HD unsigned shl_u(unsigned a, unsigned s) { return a << s; }
JB_HD unsigned ctz_u(unsigned a) { return a ? (unsigned)__builtin_ctz(a) : 0u; }
JB_HD unsigned clear_low_u(unsigned a) { return a & (a - 1u); }
JB_HD float utov(unsigned a, const float*) { return (float)a; }
JB_HD double utov(unsigned a, const double*) { return (double)a; }
// The single flipped bit of an active-set record (5 bits per slot: pyramid edges n+t1, n-t1, n+t2, n-t2, valid): which cached
// row it belongs to (entry = rank of the slot among the live slots) and which edge it is.  A lane without a flipped bit gets
// is_flip = false and entry 0.
JB_HD void flip_decode(unsigned diff0, unsigned diff1, unsigned rec0, unsigned rec1, unsigned live, unsigned& entry, bool& is_flip, bool& plus, bool& tan2, bool& on) {
    // word 0: slots 0-4, word 1: slots 5-9
    const bool w1 = diff0 == 0u;
    const unsigned diff = w1 ? diff1 : diff0, rec = w1 ? rec1 : rec0;
    is_flip = diff != 0u;
    const unsigned pos = is_flip ? (unsigned)__builtin_ctz(diff) : 0u;
    const unsigned sl = (pos * 205u) >> 10, b = pos - 5u * sl, slot = sl + (w1 ? 5u : 0u);
    entry = is_flip ? (unsigned)__builtin_popcount(live & ((1u << slot) - 1u)) : 0u;      // lanes without a flip read a row that exists
    is_flip = is_flip && b < 4u;
    plus = (b & 1u) == 0u; tan2 = b >= 2u; on = ((rec >> pos) & 1u) != 0u;
}
JB_HD float ld_gather(const float* p, int stride, unsigned idx) { return p[idx * stride]; }
JB_HD double ld_gather(const double* p, int stride, unsigned idx) { return p[idx * stride]; }
JB_HD bool neq_u(unsigned a, unsigned b) { return a != b; }
JB_HD float vsqrt(float x) { return sqrtf(x); }
JB_HD double vsqrt(double x) { return sqrt(x); }
// reciprocal / reciprocal square root: one hardware instruction on the device (v_rcp_f32 / v_rsq_f32, ~1 ulp) instead of the
// ~10-instruction IEEE division and square root sequences; exact arithmetic in the fp64 host build
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
JB_HD float vrcp(float x) { return __builtin_amdgcn_rcpf(x); }
JB_HD float vrsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
#else
JB_HD float vrcp(float x) { return 1.0f / x; }
JB_HD float vrsqrt(float x) { return 1.0f / sqrtf(x); }
#endif
JB_HD double vrcp(double x) { return 1.0 / x; }
JB_HD double vrsqrt(double x) { return 1.0 / sqrt(x); }
JB_HD float vabs(float x) { return fabsf(x); }
JB_HD double vabs(double x) { return fabs(x); }
JB_HD float vsin(float x) { return sinf(x); }
JB_HD double vsin(double x) { return sin(x); }
JB_HD float vcos(float x) { return cosf(x); }
JB_HD double vcos(double x) { return cos(x); }
JB_HD float vmin(float a, float b) { return fminf(a, b); }
JB_HD double vmin(double a, double b) { return fmin(a, b); }
JB_HD float vmax(float a, float b) { return fmaxf(a, b); }
JB_HD double vmax(double a, double b) { return fmax(a, b); }
// sin and cos of an angle already wrapped to [-pi, pi] (the motor angle): quarter-turn reduction with a two-term pi/2 and
// Taylor polynomials on [-pi/4, pi/4] (truncation < 3e-9), about a third of the instructions of sinf + cosf.  The fp64
// host build keeps libm.
JB_HD void vsincos_pi(float x, float& s, float& c) {
    const float k = rintf(x * 0.636619772367581343f);
    float r = fmaf(-k, 1.57079625129699707031f, x);
    r = fmaf(-k, 7.54978941586159635336e-08f, r);
    const float r2 = r * r;
    const float sr = r + r * r2 * (-1.0f / 6 + r2 * (1.0f / 120 + r2 * (-1.0f / 5040 + r2 * (1.0f / 362880))));
    const float cr = 1.0f + r2 * (-0.5f + r2 * (1.0f / 24 + r2 * (-1.0f / 720 + r2 * (1.0f / 40320 + r2 * (-1.0f / 3628800)))));
    const int q = (int)k & 3;
    const float a = (q & 1) ? cr : sr, b = (q & 1) ? sr : cr;
    s = (q & 2) ? -a : a;
    c = ((q + 1) & 2) ? -b : b;
}
JB_HD void vsincos_pi(double x, double& s, double& c) { s = sin(x); c = cos(x); }
// sine / cosine of a BOUNDED angle (|x| far below 1e4: a hinge angle, a yaw target, the half-angle of a reset pose) by the same
// quarter-turn reduction: no slow path, so none of libm's private-memory argument reduction ends up in a kernel
JB_HD float vsin_b(float x) { float s, c; vsincos_pi(x, s, c); return s; }
JB_HD float vcos_b(float x) { float s, c; vsincos_pi(x, s, c); return c; }
JB_HD double vsin_b(double x) { return sin(x); }
JB_HD double vcos_b(double x) { return cos(x); }
// products / fused products that the compiler may not re-fuse: the error-free transformations of the compensated position
// state (jb_sim.hpp, two_sum / sq_err) rely on a*b being rounded exactly once
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
JB_HD float vmul_rn(float a, float b) { return __fmul_rn(a, b); }
JB_HD float vadd_rn(float a, float b) { return __fadd_rn(a, b); }
#else
JB_HD float vmul_rn(float a, float b) { volatile float r = a * b; return r; }
JB_HD float vadd_rn(float a, float b) { volatile float r = a + b; return r; }
#endif
JB_HD double vmul_rn(double a, double b) { volatile double r = a * b; return r; }
JB_HD double vadd_rn(double a, double b) { volatile double r = a + b; return r; }
JB_HD float vfma(float a, float b, float c) { return fmaf(a, b, c); }
JB_HD double vfma(double a, double b, double c) { return fma(a, b, c); }
JB_HD float vfloor(float a) { return floorf(a); }
JB_HD double vfloor(double a) { return floor(a); }

template <typename V> struct lane_traits;            // ::mask, ::uint, ::real
template <> struct lane_traits<float> { using mask = bool; using uint = unsigned; using real = float; };
template <> struct lane_traits<double> { using mask = bool; using uint = unsigned; using real = double; };

// ----------------------------------------------------------------------------- Quad<T>: host emulation of 4 lanes
#if !defined(__HIPCC__)
struct Mask4 { bool v[4]; };
struct UQuad { uint32_t v[4]; };
template <typename T> struct Quad {
    T v[4];
    Quad() = default;
    Quad(T s) { v[0] = v[1] = v[2] = v[3] = s; }
    Quad(T a, T b, T c, T d) { v[0] = a; v[1] = b; v[2] = c; v[3] = d; }
    Quad& operator+=(const Quad& o) { for (int i = 0; i < 4; i++) v[i] += o.v[i]; return *this; }
    Quad& operator-=(const Quad& o) { for (int i = 0; i < 4; i++) v[i] -= o.v[i]; return *this; }
    Quad& operator*=(const Quad& o) { for (int i = 0; i < 4; i++) v[i] *= o.v[i]; return *this; }
};
#define JB_QBIN(op) \
    template <typename T> inline Quad<T> operator op(const Quad<T>& a, const Quad<T>& b) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] op b.v[i]; return r; } \
    template <typename T> inline Quad<T> operator op(const Quad<T>& a, T b) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] op b; return r; } \
    template <typename T> inline Quad<T> operator op(T a, const Quad<T>& b) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = a op b.v[i]; return r; }
JB_QBIN(+) JB_QBIN(-) JB_QBIN(*) JB_QBIN(/)
#undef JB_QBIN
template <typename T> inline Quad<T> operator-(const Quad<T>& a) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = -a.v[i]; return r; }
template <typename T> inline Quad<T> quad_sum(const Quad<T>& x) { return Quad<T>((x.v[0] + x.v[1]) + (x.v[2] + x.v[3])); }
inline UQuad quad_sum_u(const UQuad& x) { uint32_t s = (x.v[0] + x.v[1]) + (x.v[2] + x.v[3]); return UQuad{{s, s, s, s}}; }
inline bool any_lane(const Mask4& m) { return m.v[0] || m.v[1] || m.v[2] || m.v[3]; }
// ---- helper groups on the host: one THREAD per group, each running the same source on its own Quad<T> lanes (tests/host_harness.cpp).
// The cross-group operations of the device (permlane swaps, readfirstlane) become mailbox exchanges between the threads; every one of
// them is also a barrier, like the lockstep of a wave.  With no wave installed (g_host_wave == nullptr) there is one group and
// they are identities.
struct HostWave {
    int ngrp = 1, gstride = 16;
    double mbox[4][4][4];           // [group][value][lane]
    uint32_t umbox[4][4];
    std::atomic<int> arrived{0}, phase{0};
    void barrier() {
        const int ph = phase.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) == ngrp - 1) { arrived.store(0, std::memory_order_relaxed); phase.store(ph + 1, std::memory_order_release); }
        else while (phase.load(std::memory_order_acquire) == ph) std::this_thread::yield();
    }
};
inline thread_local HostWave* g_host_wave = nullptr;
inline thread_local int g_host_grp = 0;
inline void wave_sync() { if (g_host_wave) g_host_wave->barrier(); }
template <typename T> inline Quad<T> host_exchange(const Quad<T>& x, int partner) {
    HostWave* w = g_host_wave;
    for (int i = 0; i < 4; i++) w->mbox[g_host_grp][0][i] = (double)x.v[i];
    w->barrier();
    Quad<T> r;
    for (int i = 0; i < 4; i++) r.v[i] = (T)w->mbox[partner][0][i];
    w->barrier();
    return r;
}
// row g returns the total of v[g] over the four groups, associated (g0 + g2) + (g1 + g3) like the device's permlane swaps
template <typename T> inline Quad<T> row_transpose_sum(const Quad<T>& v0, const Quad<T>& v1, const Quad<T>& v2, const Quad<T>& v3) {
    HostWave* w = g_host_wave;
    if (!w) return v0;
    const Quad<T>* v[4] = {&v0, &v1, &v2, &v3};
    for (int k = 0; k < 4; k++) for (int i = 0; i < 4; i++) w->mbox[g_host_grp][k][i] = (double)v[k]->v[i];
    w->barrier();
    Quad<T> r;
    const int g = g_host_grp;
    for (int i = 0; i < 4; i++) r.v[i] = ((T)w->mbox[0][g][i] + (T)w->mbox[2][g][i]) + ((T)w->mbox[1][g][i] + (T)w->mbox[3][g][i]);
    w->barrier();
    return r;
}
template <typename T> inline Quad<T> xor_sum(const Quad<T>& x, int off, bool) {
    if (!g_host_wave) return x;
    return x + host_exchange(x, g_host_grp ^ (off / g_host_wave->gstride));
}
inline UQuad xor_sum_u(const UQuad& x, int off, bool) {
    HostWave* w = g_host_wave;
    if (!w) return x;
    for (int i = 0; i < 4; i++) w->umbox[g_host_grp][i] = x.v[i];
    w->barrier();
    UQuad r;
    const int partner = g_host_grp ^ (off / w->gstride);
    for (int i = 0; i < 4; i++) r.v[i] = x.v[i] + w->umbox[partner][i];
    w->barrier();
    return r;
}
template <typename T> inline Quad<T> xor_get(const Quad<T>& x, int off) {
    if (!g_host_wave) return x;
    return host_exchange(x, g_host_grp ^ (off / g_host_wave->gstride));
}
inline UQuad xor_get_u(const UQuad& x, int off) {
    HostWave* w = g_host_wave;
    if (!w) return x;
    for (int i = 0; i < 4; i++) w->umbox[g_host_grp][i] = x.v[i];
    w->barrier();
    UQuad r;
    const int partner = g_host_grp ^ (off / w->gstride);
    for (int i = 0; i < 4; i++) r.v[i] = w->umbox[partner][i];
    w->barrier();
    return r;
}
inline UQuad xor_get_any_u(const UQuad& x, int off) { return xor_get_u(x, off); }
template <int J, typename T> inline Quad<T> quad_bcast(const Quad<T>& x) { return Quad<T>(x.v[J]); }
inline unsigned wave_bcast_u(unsigned x) {           // the first lane of the wave is a lane of group 0
    HostWave* w = g_host_wave;
    if (!w) return x;
    w->umbox[g_host_grp][0] = x;
    w->barrier();
    const unsigned r = w->umbox[0][0];
    w->barrier();
    return r;
}
inline bool any_lane(bool m) { return m; }
inline float quad_sum(float x) { return x; }
inline double quad_sum(double x) { return x; }
template <typename T> inline Quad<T> sel(const Mask4& m, const Quad<T>& a, const Quad<T>& b) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
inline UQuad selu(const Mask4& m, const UQuad& a, const UQuad& b) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = m.v[i] ? a.v[i] : b.v[i]; return r; }
template <typename T> inline Mask4 lt(const Quad<T>& a, const Quad<T>& b) { Mask4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] < b.v[i]; return r; }
template <typename T> inline Mask4 gt(const Quad<T>& a, const Quad<T>& b) { Mask4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] > b.v[i]; return r; }
inline Mask4 mand(const Mask4& a, const Mask4& b) { Mask4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] && b.v[i]; return r; }
inline Mask4 mor(const Mask4& a, const Mask4& b) { Mask4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] || b.v[i]; return r; }
inline Mask4 mnot(const Mask4& a) { Mask4 r; for (int i = 0; i < 4; i++) r.v[i] = !a.v[i]; return r; }
inline UQuad mbit(const Mask4& a) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] ? 1u : 0u; return r; }
inline UQuad popc_u(const UQuad& a) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = (uint32_t)__builtin_popcount(a.v[i]); return r; }
inline UQuad xor_u(const UQuad& a, const UQuad& b) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] ^ b.v[i]; return r; }
inline Mask4 lt_u(const UQuad& a, uint32_t b) { Mask4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] < b; return r; }
inline UQuad or_u(const UQuad& a, const UQuad& b) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] | b.v[i]; return r; }
inline UQuad and_u(const UQuad& a, const UQuad& b) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] & b.v[i]; return r; }
inline UQuad and_u(const UQuad& a, uint32_t b) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] & b; return r; }
inline UQuad sub_u(const UQuad& a, const UQuad& b) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] - b.v[i]; return r; }
inline UQuad shl_u(const UQuad& a, const UQuad& s) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] << s.v[i]; return r; }
inline UQuad ctz_u(const UQuad& a) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] ? (uint32_t)__builtin_ctz(a.v[i]) : 0u; return r; }
inline UQuad clear_low_u(const UQuad& a) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] & (a.v[i] - 1u); return r; }
inline Mask4 lt_u(const UQuad& a, const UQuad& b) { Mask4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] < b.v[i]; return r; }
inline Mask4 eq_u(const UQuad& a, uint32_t b) { Mask4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] == b; return r; }
template <typename T> inline Quad<T> utov(const UQuad& a, const Quad<T>*) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = (T)a.v[i]; return r; }
template <int K, typename T> inline Quad<T> quad_rot(const Quad<T>& x) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = x.v[(i + K) & 3]; return r; }
template <int K> inline UQuad quad_rot_u(const UQuad& x) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = x.v[(i + K) & 3]; return r; }
template <int J> inline UQuad quad_bcast_u(const UQuad& x) { return UQuad{{x.v[J], x.v[J], x.v[J], x.v[J]}}; }
template <typename T> inline UQuad quad_lane_id(const Quad<T>*) { return UQuad{{0u, 1u, 2u, 3u}}; }
template <typename T> inline void st_leg(Quad<T>* p, int stride, const UQuad& idx, const UQuad& src, const UQuad&, const Quad<T>& v, const Mask4& on) { for (int i = 0; i < 4; i++) if (on.v[i]) p[idx.v[i] * stride].v[src.v[i]] = v.v[i]; }
template <typename T> inline Quad<T> quad_pick(const Quad<T>& x, const UQuad& src) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = x.v[src.v[i]]; return r; }
template <typename T> inline Quad<T> ld_leg(const Quad<T>* p, int stride, const UQuad& idx, const UQuad& src, const UQuad&) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = p[idx.v[i] * stride].v[src.v[i]]; return r; }
inline Mask4 eq_u(const UQuad& a, const UQuad& b) { Mask4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] == b.v[i]; return r; }
inline void flip_decode(const UQuad& diff0, const UQuad& diff1, const UQuad& rec0, const UQuad& rec1, unsigned live, UQuad& entry, Mask4& is_flip, Mask4& plus, Mask4& tan2, Mask4& on) {
    for (int i = 0; i < 4; i++) flip_decode(diff0.v[i], diff1.v[i], rec0.v[i], rec1.v[i], live, entry.v[i], is_flip.v[i], plus.v[i], tan2.v[i], on.v[i]);
}
template <typename T> inline Quad<T> ld_gather(const Quad<T>* p, int stride, const UQuad& idx) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = p[idx.v[i] * stride].v[i]; return r; }
inline Mask4 neq_u(const UQuad& a, const UQuad& b) { Mask4 r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] != b.v[i]; return r; }
inline UQuad operator*(const UQuad& a, uint32_t b) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] * b; return r; }
inline UQuad operator+(const UQuad& a, const UQuad& b) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] + b.v[i]; return r; }
inline UQuad operator+(const UQuad& a, uint32_t b) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] + b; return r; }
#define JB_QUN(name, fn) template <typename T> inline Quad<T> name(const Quad<T>& a) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = fn(a.v[i]); return r; }
template <typename T> inline Quad<T> vmul_rn(const Quad<T>& a, const Quad<T>& b) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = vmul_rn(a.v[i], b.v[i]); return r; }
template <typename T> inline Quad<T> vadd_rn(const Quad<T>& a, const Quad<T>& b) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = vadd_rn(a.v[i], b.v[i]); return r; }
template <typename T> inline Quad<T> vfma(const Quad<T>& a, const Quad<T>& b, const Quad<T>& c) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = vfma(a.v[i], b.v[i], c.v[i]); return r; }
JB_QUN(vsqrt, vsqrt) JB_QUN(vrcp, vrcp) JB_QUN(vrsqrt, vrsqrt) JB_QUN(vabs, vabs) JB_QUN(vsin, vsin) JB_QUN(vcos, vcos) JB_QUN(vfloor, vfloor)
#undef JB_QUN
template <typename T> inline void vsincos_pi(const Quad<T>& x, Quad<T>& s, Quad<T>& c) { for (int i = 0; i < 4; i++) vsincos_pi(x.v[i], s.v[i], c.v[i]); }
template <typename T> inline Quad<T> vmin(const Quad<T>& a, const Quad<T>& b) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = vmin(a.v[i], b.v[i]); return r; }
template <typename T> inline Quad<T> vmax(const Quad<T>& a, const Quad<T>& b) { Quad<T> r; for (int i = 0; i < 4; i++) r.v[i] = vmax(a.v[i], b.v[i]); return r; }
template <typename T> struct lane_traits<Quad<T>> { using mask = Mask4; using uint = UQuad; using real = T; };
#endif

// ----------------------------------------------------------------------------- two values per lane: packed fp32
// W = Pk2<V> holds TWO independent values of lane type V that go through the same arithmetic - the upper and the lower body of a leg in
// phase A of jb_sim.hpp, the two tangential directions of a contact in phase B.  On the device (V = float) it is a 2-wide vector in an
// aligned register pair and every + - * (and the multiply-adds the compiler fuses from them) is ONE v_pk_add_f32 / v_pk_mul_f32 /
// v_pk_fma_f32: a lone wave issues those at the rate of their scalar forms (profiles/r03_pk_issue_microbench.txt), so a pair costs one
// issue slot instead of two.  A value used for both halves (W(x, x)) needs no register of its own: the packed forms read either half of any
// pair for either result (op_sel).  The halves never meet inside a packed operation - they are two IEEE operations side by side.
// On the host (V = Quad<T>) it is a struct of two V: the fp64 harness runs the very same source.
template <typename V> struct Pk2 {
    V lo, hi;
    Pk2() = default;
    template <typename S, typename = typename std::enable_if<std::is_arithmetic<S>::value>::type>
    JB_HD Pk2(S s) : lo(V((typename lane_traits<V>::real)s)), hi(V((typename lane_traits<V>::real)s)) {}
    JB_HD Pk2(const V& a, const V& b) : lo(a), hi(b) {}
    JB_HD explicit Pk2(const V& both) : lo(both), hi(both) {}
};
template <typename V> JB_HD Pk2<V> operator+(const Pk2<V>& a, const Pk2<V>& b) { return Pk2<V>(a.lo + b.lo, a.hi + b.hi); }
template <typename V> JB_HD Pk2<V> operator-(const Pk2<V>& a, const Pk2<V>& b) { return Pk2<V>(a.lo - b.lo, a.hi - b.hi); }
template <typename V> JB_HD Pk2<V> operator*(const Pk2<V>& a, const Pk2<V>& b) { return Pk2<V>(a.lo * b.lo, a.hi * b.hi); }
template <typename V> JB_HD Pk2<V> operator-(const Pk2<V>& a) { return Pk2<V>(-a.lo, -a.hi); }
template <typename V> JB_HD V pk_lo(const Pk2<V>& a) { return a.lo; }
template <typename V> JB_HD V pk_hi(const Pk2<V>& a) { return a.hi; }
#if defined(__HIPCC__)
typedef float jb_f32x2 __attribute__((ext_vector_type(2)));
template <> struct Pk2<float> {
    jb_f32x2 v;
    Pk2() = default;
    JB_HD Pk2(float s) : v{s, s} {}
    JB_HD Pk2(float a, float b) : v{a, b} {}          // (one build-vector: element-wise stores into the uninitialised member leave loads SROA cannot split)
    JB_HD explicit Pk2(jb_f32x2 q) : v(q) {}
};
JB_HD Pk2<float> operator+(const Pk2<float>& a, const Pk2<float>& b) { return Pk2<float>(a.v + b.v); }
JB_HD Pk2<float> operator-(const Pk2<float>& a, const Pk2<float>& b) { return Pk2<float>(a.v - b.v); }
JB_HD Pk2<float> operator*(const Pk2<float>& a, const Pk2<float>& b) { return Pk2<float>(a.v * b.v); }
JB_HD Pk2<float> operator-(const Pk2<float>& a) { return Pk2<float>(-a.v); }
JB_HD float pk_lo(const Pk2<float>& a) { return a.v.x; }
JB_HD float pk_hi(const Pk2<float>& a) { return a.v.y; }
#endif
// A value the optimiser may not look through: an error-free sum (jb_sim.hpp comp_add2) needs its addend rounded ONCE, as a value of its
// own - never folded into a fused multiply-add with the sum that follows.
template <typename V> JB_HD Pk2<V> pk_pin(const Pk2<V>& a) { return a; }          // (the host builds do not contract across statements)
#if defined(__HIPCC__) && defined(__HIP_DEVICE_COMPILE__)
JB_HD Pk2<float> pk_pin(const Pk2<float>& a) { jb_f32x2 r = a.v; asm("" : "+v"(r)); return Pk2<float>(r); }
#endif
template <typename V> struct lane_traits<Pk2<V>> { using mask = typename lane_traits<V>::mask; using uint = typename lane_traits<V>::uint; using real = typename lane_traits<V>::real; };

// unsigned helpers
JB_HD unsigned umulv(unsigned a, unsigned b) { return a * b; }
JB_HD unsigned vtou(float a) { return (unsigned)a; }
JB_HD unsigned vtou(double a) { return (unsigned)a; }
#if !defined(__HIPCC__)
inline UQuad umulv(const UQuad& a, const UQuad& b) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = a.v[i] * b.v[i]; return r; }
template <typename T> inline UQuad vtou(const Quad<T>& a) { UQuad r; for (int i = 0; i < 4; i++) r.v[i] = (uint32_t)a.v[i]; return r; }
#endif

}  // namespace jb
