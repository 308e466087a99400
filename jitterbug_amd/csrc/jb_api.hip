// jb_api.hip — kernels and host side of the C ABI declared in include/jitterbug_hip.h.
//
// Layout in HBM (all fp32 unless noted), N environments, 4 lanes per environment:
//   root [ROOT_F][N]      root pose/velocity, motor angle/rate/turns, warm-start accelerations, target
//   leg  [LEG_F][4N]      per-leg hinge angles/rates and warm-start accelerations (lane-private, fully coalesced)
//   lane_model [T][LM_TABLE]      packed constant tables (jb_sim.hpp LM_INV; T = 1 shared, or N per-env)
//   step_count[N] (i32), episode[N] (u32)
// One launch of jb_step_kernel advances every env by one control step (cfg.substeps physics substeps),
// then computes reward, done, optional in-kernel episode reset, and the observation row.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/jitterbug_hip.h"
#include "jb_default_params.h"
#include "jb_device_guard.hpp"
#include "jb_model_build.hpp"
#include "jb_model_compile.hpp"
#include "jb_nominal_spec.h"
#include "jb_sim.hpp"
#include "jb_step.hpp"
#include "jb_task.hpp"
#include "jb_witness.hpp"

using namespace jb;

#ifndef JB_DEFAULT_MAX_NEWTON
#define JB_DEFAULT_MAX_NEWTON 12      // checks of the active set after which a substep's contact problem goes to the line-searched solve (jb_sim.hpp newton_phase)
#endif

namespace {

constexpr int OVC_FLOATS_PER_LANE = 4 * (NSLOT - ROW_K) + 9;      // LEAN kernels' per-wave block in global memory: overflow candidates + the second pair contact's frame
enum RootF : int { RF_P = 0, RF_Q = 3, RF_V = 7, RF_W = 10, RF_PHI = 13, RF_PHID = 14, RF_TURNS = 15, RF_WA = 16, RF_WL = 19, RF_WM = 22, RF_FAIL = 23, RF_TGT = 24, RF_LO = 27 /*5: low-order words of z and the quaternion*/, ROOT_F = 32 };
enum LegF : int { LF_TH1 = 0, LF_TH2 = 1, LF_THD1 = 2, LF_THD2 = 3, LF_WJ0 = 4, LF_WJ1 = 5, LEG_F = 6 };

struct KArgs {
    int n, task, substeps, step_limit, auto_reset, contacts, max_newton, random_pose, per_env_model;
    int epw;       // environments per wave (= per 64-thread workgroup) of the step kernel
    int rank_one;      // 0: diagnostic, no rank-one Newton passes (jb_config.flags & JB_FLAG_NO_RANK_ONE)
    int spread;        // 0: diagnostic, no spread sweeps (JB_FLAG_NO_SPREAD)
    int lean;          // 1: the LEAN kernel variant (two waves per SIMD; jb_config.flags & JB_FLAG_LEAN)
    int pair;          // 1: the PAIR kernel variant (geom-geom contact mass ellipsoid / upper-leg cylinders; see JB_FLAG_PAIR)
    int packed_rows;   // step kernel output: 0 = obs[N,D] + reward[N] + done[N]; 1 = one float row [obs(D) | reward | done] per env
    unsigned long long seed, env_offset;
    float* root; float* leg; const float* lane_model;
    int* step_count; unsigned* episode;
    float* ovc_buf;    // LEAN variant: per wave, the candidates of the live slots beyond the row cache (global memory: the variant's 20 KB of LDS have no room)
    unsigned long long* wave_stats;   // diagnostic builds (-DJB_WAVE_STATS): [n_waves][4] = cycles, rare-path substeps, Newton sweeps, contact substeps
    float* capture; unsigned* capture_count;      // diagnostic builds (-DJB_CAPTURE): SimOpts::capture
    unsigned long long* resolve_count;   // [1]: wave-substeps whose contact solve ran a second time with the exact line search (jb_sim.hpp newton_phase), since jb_create
};

// What one launch of a step kernel reads and writes besides the state: K control steps in ONE launch (jb_step_many_device; K = 1 is the
// ordinary step - the same kernel, hence the same machine code and the same bits).  A wave owns its environments for all K steps:
// state, step counter, episode and target stay in registers (LEAN: LDS) between the control steps, and no wave waits for another.
struct StepIO {
    int n_steps;                   // K >= 1
    int use_policy;                // 1: the action of every step is the handle's heuristic policy, evaluated in the kernel on the observation the lanes just produced
    const float* actions;          // use_policy = 0: action tape [K, N] (step k reads row k)
    const float* obs_in;           // use_policy = 1, nullable: [N, D] current observation rows for step 0 (NULL: observed from the state in the kernel)
    float* obs_out;                // nullable: obs [.., N, D], or packed rows [.., N, D+2] when KArgs::packed_rows
    float* reward_out;             // nullable [.., N]   (not packed rows)
    unsigned char* done_out;       // nullable [.., N]   (not packed rows)
    int every_step;                // bit 0 / 1 / 2: obs (rows) / reward / done blocks are written for EVERY step ([K, ...] buffers); clear: only the last step's
    PolicyParams<float> pp;
    unsigned long long* wave_clock;      // nullable [n_waves]: how long each wave lived in this launch (s_memrealtime ticks, 100 MHz): the imbalance a fused rollout removes
    const int* wave_order;               // nullable [n_waves]: workgroup b steps wave wave_order[b] - the waves of the PREVIOUS launch, longest first (jb_wave_order_kernel)
};

// Stage the packed constant table(s) of this workgroup (one wave = a.epw envs) into LDS: one LM_TABLE copy for a
// shared model, epw copies for per-env models.  Called by all 64 threads before the idle quads retire.
// aux_grp: -1 = no aux bodies in this kernel; otherwise the caller's lane group (groups 2 and 3 are the aux lanes: SimOpts::aux)
template <bool SPLIT = false>
__device__ __forceinline__ void stage_model(const KArgs& a, float* lds, int lblock, int quad, int leg, LaneModel<float>& m, bool lean = false, bool pair = true, int aux_grp = -1) {
    m.c.lean = lean;
    m.c.split = false;
    const int tsz = pair ? LM_TABLE : LM_TABLE_BASE, gsz = LM_TABLE;      // staged prefix / table pitch in global memory
    if (SPLIT) {
        // split mode (jb_sim.hpp LaneConsts): per env the resident block only - the lane-invariant prefix, the pair's ellipsoid (lane 0's copy)
        // and the five per-lane entries read after phase A; phase A's per-lane entries are re-staged every substep (restage_overlay)
        const int env0 = lblock * a.epw, rsz = LM_SPLIT_RES;
        for (int i = threadIdx.x; i < rsz * a.epw; i += blockDim.x) {
            int e = env0 + i / rsz;
            if (e >= a.n) e = a.n - 1;
            const int k = i % rsz;
            int src = k;
            if (k >= LM_INV + 15) { const int r = (k - LM_INV - 15) >> 2, l = (k - LM_INV - 15) & 3; src = LM_INV + 4 * (lm_split_res_entry(r) - LM_INV) + l; }
            else if (k >= LM_INV) src = LM_INV + 4 * (LM_PE_C + (k - LM_INV) - LM_INV);
            lds[i] = a.lane_model[(size_t)e * gsz + src];
        }
        m.c.inv = lds + quad * rsz;
        m.c.tab = m.c.inv + LM_INV + 15 + leg;
        m.c.split = true;
        const int env = lblock * a.epw + quad;
        m.c.cold = a.lane_model + (size_t)(env < a.n ? env : a.n - 1) * gsz + LM_INV + leg;
        // overlay: scratch entries [SC_SYS, SC_SYS + LM_SPLIT_OVL) of the main lanes (the scratch precedes the resident block in LDS)
        float* const scratch0 = lds - (size_t)SC_COUNT_LEAN_PAIR * LM_SPLIT_STRIDE;
        const int l16 = threadIdx.x % LM_SPLIT_STRIDE, j_sub = threadIdx.x / LM_SPLIT_STRIDE;
        m.c.ovl = scratch0 + SC_SYS * LM_SPLIT_STRIDE + l16;
        int e_src = lblock * a.epw + (l16 >> 2);
        if (e_src >= a.n) e_src = a.n - 1;
        m.c.ovl_src = a.lane_model + (size_t)e_src * gsz + LM_INV + (l16 & 3) + 4 * j_sub;
        m.c.ovl_dst = scratch0 + SC_SYS * LM_SPLIT_STRIDE + threadIdx.x;
        __syncthreads();
        restage_overlay(m.c);             // (preload below reads the per-level weights from it)
        __syncthreads();
    } else if (a.per_env_model) {
        const int env0 = lblock * a.epw;
        for (int i = threadIdx.x; i < tsz * a.epw; i += blockDim.x) {
            int e = env0 + i / tsz;
            if (e >= a.n) e = a.n - 1;
            lds[i] = a.lane_model[(size_t)e * gsz + (i % tsz)];
        }
        if (aux_grp >= 0)          // one aux block per env behind the tables (jb_sim.hpp build_aux_block)
            for (int k = threadIdx.x; k < LM_AUX * a.epw; k += blockDim.x) {
                int e = env0 + k / LM_AUX;
                if (e >= a.n) e = a.n - 1;
                const int kk = k % LM_AUX;
                lds[tsz * a.epw + k] = aux_entry<float>(a.lane_model + (size_t)e * gsz, LM_INV + (kk >> 2), kk & 3);
            }
        __syncthreads();
        m.c.inv = lds + quad * tsz;
        if (aux_grp >= 2) { m.c.tab_rare = m.c.inv + LM_INV + leg; m.c.tab = lds + tsz * a.epw + quad * LM_AUX + leg; }
    } else {
        for (int i = threadIdx.x; i < tsz; i += blockDim.x) lds[i] = a.lane_model[i];
        if (aux_grp >= 0)          // the aux block behind the table (jb_sim.hpp build_aux_block): the motor body and the root body as two "legs"
            for (int k = threadIdx.x; k < LM_AUX; k += blockDim.x) lds[tsz + k] = aux_entry<float>(a.lane_model, LM_INV + (k >> 2), k & 3);
        __syncthreads();
        m.c.inv = lds;
    }
    if (!SPLIT && !(a.per_env_model && aux_grp >= 2)) m.c.tab = m.c.inv + LM_INV + leg;
    if (!SPLIT && !a.per_env_model && aux_grp >= 2) { m.c.tab_rare = m.c.tab; m.c.tab = lds + tsz + leg; }      // an aux lane: hot per-lane entries from the aux block, the rest from the leg it mirrors
    m.c.preload();
}
__device__ __forceinline__ void load_state(const KArgs& a, int env, int lane, LaneState<float>& s) {
    const float* r = a.root + env;
    const int N = a.n;
    s.px = r[(RF_P + 0) * N]; s.py = r[(RF_P + 1) * N]; s.pz = r[(RF_P + 2) * N];
    s.qw = r[(RF_Q + 0) * N]; s.qx = r[(RF_Q + 1) * N]; s.qy = r[(RF_Q + 2) * N]; s.qz = r[(RF_Q + 3) * N];
    s.vx = r[(RF_V + 0) * N]; s.vy = r[(RF_V + 1) * N]; s.vz = r[(RF_V + 2) * N];
    s.wx = r[(RF_W + 0) * N]; s.wy = r[(RF_W + 1) * N]; s.wz = r[(RF_W + 2) * N];
    s.phi = r[RF_PHI * N]; s.phid = r[RF_PHID * N]; s.turns = r[RF_TURNS * N];
#pragma unroll
    for (int i = 0; i < 3; i++) { s.wa[i] = r[(RF_WA + i) * N]; s.wl[i] = r[(RF_WL + i) * N]; }
    s.wm = r[RF_WM * N]; s.fail = r[RF_FAIL * N];
    s.pz_lo = r[(RF_LO + 0) * N]; s.qw_lo = r[(RF_LO + 1) * N]; s.qx_lo = r[(RF_LO + 2) * N]; s.qy_lo = r[(RF_LO + 3) * N]; s.qz_lo = r[(RF_LO + 4) * N];
    const float* l = a.leg + lane;
    const int L = 4 * N;
    s.th1 = l[LF_TH1 * L]; s.th2 = l[LF_TH2 * L]; s.thd1 = l[LF_THD1 * L]; s.thd2 = l[LF_THD2 * L]; s.wj[0] = l[LF_WJ0 * L]; s.wj[1] = l[LF_WJ1 * L];
}
__device__ __forceinline__ void store_state(const KArgs& a, int env, int lane, int leg, const LaneState<float>& s) {
    const int N = a.n, L = 4 * N;
    float* l = a.leg + lane;
    l[LF_TH1 * L] = s.th1; l[LF_TH2 * L] = s.th2; l[LF_THD1 * L] = s.thd1; l[LF_THD2 * L] = s.thd2; l[LF_WJ0 * L] = s.wj[0]; l[LF_WJ1 * L] = s.wj[1];
    // the replicated root block: each of the 4 lanes writes a quarter of the fields
    float* r = a.root + env;
    const float vals[24] = {s.px, s.py, s.pz, s.qw, s.qx, s.qy, s.qz, s.vx, s.vy, s.vz, s.wx, s.wy, s.wz, s.phi, s.phid, s.turns,
                            s.wa[0], s.wa[1], s.wa[2], s.wl[0], s.wl[1], s.wl[2], s.wm, s.fail};
#pragma unroll
    for (int f = 0; f < 24; f++) if ((f & 3) == leg) r[f * N] = vals[f];
    const float los[5] = {s.pz_lo, s.qw_lo, s.qx_lo, s.qy_lo, s.qz_lo};
#pragma unroll
    for (int f = 0; f < 5; f++) if ((f & 3) == leg) r[(RF_LO + f) * N] = los[f];
}
__device__ __forceinline__ void core_from_state(const LaneState<float>& s, const float (&c0)[3], const float (&tgt)[3], EnvCore<float>& e) {
    e.cx = c0[0]; e.cy = c0[1]; e.cz = c0[2];
    e.px = s.px; e.py = s.py; e.pz = s.pz; e.qw = s.qw; e.qx = s.qx; e.qy = s.qy; e.qz = s.qz;
    e.vx = s.vx; e.vy = s.vy; e.vz = s.vz; e.wx = s.wx; e.wy = s.wy; e.wz = s.wz; e.phi = s.phi; e.phid = s.phid;
    e.tx = tgt[0]; e.ty = tgt[1]; e.tpsi = tgt[2];
}
__device__ __forceinline__ void core_from_state(const KArgs& a, int env, const LaneState<float>& s, const float (&c0)[3], EnvCore<float>& e) {
    const float tgt[3] = {a.root[(RF_TGT + 0) * a.n + env], a.root[(RF_TGT + 1) * a.n + env], a.root[(RF_TGT + 2) * a.n + env]};
    core_from_state(s, c0, tgt, e);
}
__device__ __forceinline__ void state_from_reset(const EnvCore<float>& e, LaneState<float>& s) {
    s.px = e.px; s.py = e.py; s.pz = e.pz; s.qw = e.qw; s.qx = e.qx; s.qy = e.qy; s.qz = e.qz;
    s.pz_lo = s.qw_lo = s.qx_lo = s.qy_lo = s.qz_lo = 0.f;
    s.vx = s.vy = s.vz = s.wx = s.wy = s.wz = 0.f; s.phi = 0.f; s.phid = 0.f; s.turns = 0.f;
    s.th1 = s.th2 = s.thd1 = s.thd2 = 0.f;
#pragma unroll
    for (int i = 0; i < 3; i++) { s.wa[i] = 0.f; s.wl[i] = 0.f; }
    s.wj[0] = s.wj[1] = 0.f; s.wm = 0.f;
}
__device__ __forceinline__ void write_obs(const KArgs& a, int env, int leg, const EnvCore<float>& e, float target_z, float* obs_out, int row_stride = 0) {
    if (!obs_out) return;
    float obs[19];
    observe<float>(a.task, e, target_z, obs, 1);
    const int D = obs_dim(a.task);
    float* row = obs_out + (size_t)env * (row_stride ? row_stride : D);
#pragma unroll
    for (int j = 0; j < 19; j++) if ((j & 3) == leg && j < D) row[j] = obs[j];
}
__device__ __forceinline__ void store_target(const KArgs& a, int env, const EnvCore<float>& e) {
    a.root[(RF_TGT + 0) * a.n + env] = e.tx; a.root[(RF_TGT + 1) * a.n + env] = e.ty; a.root[(RF_TGT + 2) * a.n + env] = e.tpsi;
}

// ---------------------------------------------------------------------------------------------- step
// The step kernels' arguments (KArgs, StepIO) live in the kernarg segment.  What the control-step epilogue needs of them (output
// pointers, task options, RNG keys: ~60 scalar registers) must not stay in scalar registers across the 50 substeps - the substep loop
// needs them all, and what does not fit is spilled to VGPR lanes and read back with v_readlane inside the hot loop.  The epilogue
// therefore reads its arguments where it uses them, through a pointer to the kernarg segment that the optimiser cannot see through
// (so the loads cannot be hoisted above the substep loop): a handful of s_load per control step.
typedef const __attribute__((address_space(4))) KArgs* KArgsC;
typedef const __attribute__((address_space(4))) StepIO* StepIOC;
// (AMDGPU kernarg ABI: the explicit arguments lie in declaration order, each at its natural alignment - KArgs at 0, StepIO behind it)
constexpr unsigned STEPIO_KERNARG_OFFSET = (unsigned)((sizeof(KArgs) + alignof(StepIO) - 1) / alignof(StepIO) * alignof(StepIO));
static_assert(alignof(KArgs) == 8 && alignof(StepIO) == 8 && std::is_trivially_copyable<KArgs>::value && std::is_trivially_copyable<StepIO>::value, "kernel arguments are plain data with pointer alignment");
__device__ __forceinline__ const __attribute__((address_space(4))) char* kernarg_base() {
    const __attribute__((address_space(4))) char* p = (const __attribute__((address_space(4))) char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));
    return p;
}
// EPW (envs per wave) is a template parameter so that the scratch stride is a compile-time constant and every LDS access
// of the substep uses an immediate offset instead of integer address arithmetic.
template <int EPW, bool LEAN, bool PAIR = false>
__device__ __forceinline__ void step_body(KArgs a, StepIO io) {
    // one wave per workgroup; quad q (4 lanes) of the wave owns env blockIdx*epw + q.  Quads beyond epw (a small batch is
    // spread over all SIMDs with partially filled waves) and beyond the batch retire at once: DPP quad sums and the
    // wave ballots only ever involve complete, active quads.
    extern __shared__ float lds[];           // [SC_COUNT][4*EPW] per-lane scratch, then the lane constant table(s)
    // Lanes 0 .. 4*EPW-1 are the MAIN lanes (quad q = env q of this wave, lane = leg).  A wave with fewer than 16 envs uses
    // its spare lanes as NGRP-1 helper groups: helper lane L + g*4*EPW mirrors main lane L (same env, leg, scratch and
    // constant-table addresses) and takes a share of the live contact slots in every Newton pass (jb_sim.hpp, SlotPlan).
    constexpr int MAIN = 4 * EPW;
    constexpr int NGRP = EPW == 8 ? 2 : 4;
    static_assert(EPW == 1 || EPW == 2 || EPW == 4 || EPW == 8, "envs per wave");
    const int lane_in_grp = threadIdx.x % MAIN, grp = threadIdx.x / MAIN;
    const int quad = lane_in_grp >> 2, leg = threadIdx.x & 3;
    // XCD-aware workgroup -> env-range map: hardware deals workgroups round-robin over the 8 XCDs (b and b+8 share one),
    // so give XCD x one contiguous range of envs; then every 128-byte line of the SoA state arrays is touched by a single
    // XCD's L2 instead of all eight (placement is a speed/traffic matter only, never correctness).
    const int nb = (int)gridDim.x, b = (int)blockIdx.x, xcd = b & 7, per = nb >> 3, rem = nb & 7;
    // More waves than the device holds at once: the hardware hands out workgroups in index order, so the order decides which waves share
    // a SIMD's time.  Longest-first by the previous launch's measured wave times (a robot that has tipped over, or whose mass rubs a leg,
    // stays slow) keeps the slow waves out of the last round; which workgroup steps which envs never shows in the results.
    const int lblock = io.wave_order ? io.wave_order[b] : xcd * per + (xcd < rem ? xcd : rem) + (b >> 3);
    const int env = lblock * EPW + quad;
    const unsigned long long clock0 = __builtin_amdgcn_s_memrealtime();
    LaneModel<float> m;
    constexpr int SCN = LEAN ? (PAIR ? SC_COUNT_LEAN_PAIR : SC_COUNT_LEAN) : SC_COUNT;      // floats of per-lane scratch
    // AUX bodies (jb_sim.hpp SimOpts::aux): the ordinary kernel with four lane groups and a shared model - groups 2 and 3 run phase A on the
    // motor body and the root body's own mass instead of idling while every leg lane repeats that work
    constexpr bool AUX = !LEAN && !PAIR && NGRP == 4;
    constexpr bool aux_on = AUX;
    stage_model<LEAN && PAIR>(a, lds + SCN * 4 * EPW, lblock, quad, leg, m, LEAN, PAIR || !LEAN, aux_on ? grp : -1);      // (LEAN + PAIR is only launched with one model per env: split tables)
    if (grp >= NGRP || env >= a.n) return;       // whole quads (and their mirrors in every group) retire together
    const int lane = env * 4 + leg;
    LaneScratch<float> scr;
    scr.p = lds + lane_in_grp;
    scr.stride = MAIN;
    scr.grp = grp; scr.ngrp = NGRP; scr.gstride = MAIN;
    // (LEAN: per wave a block in global memory - the overflow candidates, then the second pair contact's frame: rare paths both)
    if (LEAN) { scr.ovc = a.ovc_buf + (size_t)lblock * (OVC_FLOATS_PER_LANE * MAIN) + lane_in_grp; scr.pd2 = 4 * (NSLOT - ROW_K); scr.red_lds = false; }
    else { scr.ovc = lds + SC_OVC * MAIN + lane_in_grp; scr.pd2 = SC_PD2 - SC_OVC; scr.red_lds = true; }
    scr.ovc_stride = MAIN;
    scr.pd = LEAN ? SC_PD_LEAN : SC_PD;
    scr.aux_lane = aux_on && grp >= 2;
#ifdef JB_WAVE_STATS
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif
    LaneState<float> s;
    constexpr bool OFFLOAD = !LEAN;              // lane group 1 replicates the main lanes (jb_sim.hpp SimOpts::offload)
    // the lanes that hold an env's state: the main lanes, their replica and the aux lanes.  All run everything below that changes the state
    // (substeps, failure flag, episode reset) with the very same instructions; only the main lanes write to memory.
    const bool rep = grp == 0 || (OFFLOAD && grp == 1) || scr.aux_lane;
    if (rep) load_state(a, env, lane, s);
    else {
        s.px = s.py = s.pz = 0.f; s.qw = 1.f; s.qx = s.qy = s.qz = 0.f; s.vx = s.vy = s.vz = s.wx = s.wy = s.wz = 0.f;
        s.pz_lo = s.qw_lo = s.qx_lo = s.qy_lo = s.qz_lo = 0.f;
        s.phi = s.phid = s.turns = 0.f; s.th1 = s.th2 = s.thd1 = s.thd2 = 0.f;
        for (int i = 0; i < 3; i++) { s.wa[i] = 0.f; s.wl[i] = 0.f; }
        s.wj[0] = s.wj[1] = 0.f; s.wm = 0.f; s.fail = 0.f;
    }
#ifdef JB_WAVE_STATS
    s.st_xtra = 0.f; s.st_sweeps = 0.f; s.st_contact = 0.f; s.st_slots = 0.f; s.st_fast = 0.f; s.st_checks = 0.f;
#endif
    SimOpts o; o.contacts = a.contacts; o.max_newton = a.max_newton; o.implicit_damp = 1; o.rank_one = a.rank_one; o.lean = LEAN ? 1 : 0; o.offload = OFFLOAD ? 1 : 0; o.spread = a.spread; o.prof = nullptr; o.hist = nullptr;
    o.aux = aux_on ? 1 : 0;
#ifdef JB_CAPTURE
    o.capture = a.capture; o.capture_count = a.capture_count;
#endif
#ifdef JB_WAVE_STATS
    const unsigned long long rt_start = __builtin_amdgcn_s_memrealtime();
    unsigned long long prof_local[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // accumulated in registers, written once at the end
    o.prof = prof_local;
    o.hist = a.wave_stats ? a.wave_stats + (size_t)16 * a.n + (size_t)64 * lblock : nullptr;
#endif
    if (OFFLOAD && grp == 0) {
#pragma unroll
        for (int i = 0; i < 56; i++) scr.st(SC_ZERO + i, 0.f);
    }
    // per-env bookkeeping that a K-step launch carries in registers: step counter, episode number, target
    EpisodeRegs<float> er;
    er.step_count = a.step_count[env];
    er.episode = a.episode[env];
    er.tgt[0] = a.root[(RF_TGT + 0) * a.n + env]; er.tgt[1] = a.root[(RF_TGT + 1) * a.n + env]; er.tgt[2] = a.root[(RF_TGT + 2) * a.n + env];
    const int n_substeps = a.substeps;
    float ctrl_next = 0.f;           // use_policy: the action of the NEXT step, computed right after the observation it reads
    if (io.use_policy && rep) {      // the observation the first action is computed from: the caller's row, or the state observed here
        const int D = obs_dim(a.task);
        float obs0[19];
#pragma unroll
        for (int j = 0; j < 19; j++) obs0[j] = 0.f;
        if (io.obs_in) {
#pragma unroll
            for (int j = 0; j < 19; j++) if (j < D) obs0[j] = io.obs_in[(size_t)env * D + j];
        } else {
            EnvCore<float> e0;
            core_from_lane_state<float>(m, s, er.tgt, e0);
            observe<float>(a.task, e0, m.c[LM_TARGET_Z], obs0, 1);
        }
        ctrl_next = heuristic_policy<float>(a.task, obs0, 1, io.pp);
    }
    // (from here on the arguments are read through kernarg_base(): see above)
    int k = 0;
#pragma unroll 1
    for (;;) {
        float ctrl = ctrl_next;
        {
            const auto kb = kernarg_base();
            const KArgsC ka = (KArgsC)kb;
            const StepIOC ic = (StepIOC)(kb + STEPIO_KERNARG_OFFSET);
            if (rep && !ic->use_policy) ctrl = ic->actions[(size_t)k * ka->n + env];
        }
        if (rep) normalise_state(s);          // mj_kinematics normalises the free-joint quaternion; phase C keeps it normalised from here on
        if (PAIR && rep) scr.st(scr.pd + 11, 0.f);      // the narrow phase starts cold in every control step (its warm start is not simulator state)
        if (LEAN && grp == 0) state_store(scr, s);                   // LEAN: the state lives in the scratch between substeps
#pragma unroll 1
        for (int i = 0; i < n_substeps; i++) {
            if (__builtin_expect(substep<float, PAIR>(m, scr, s, ctrl, o), false)) {      // rare (a few substeps in ten million): counted for jb_solver_stats
                if (threadIdx.x == 0) atomicAdd(((KArgsC)kernarg_base())->resolve_count, 1ull);
            }
        }
        const auto kb = kernarg_base();
        const KArgsC ka = (KArgsC)kb;
        const StepIOC ic = (StepIOC)(kb + STEPIO_KERNARG_OFFSET);
        const bool last = k == ic->n_steps - 1;
        k++;
        if (!rep) { if (last) break; continue; }      // helper lanes only take part in the substeps
        if (LEAN) state_load(scr, s);
        const int task = ka->task, N = ka->n, D = obs_dim(task), packed = ka->packed_rows, W = packed ? D + 2 : D;
        TaskOpts topt;
        topt.task = task; topt.step_limit = ka->step_limit; topt.auto_reset = ka->auto_reset; topt.random_pose = ka->random_pose;
        topt.seed = ka->seed; topt.env_global = ka->env_offset + (unsigned long long)env;
        float obs[19], rew;
        bool done;
        control_step_tail<float>(topt, m, s, er, obs, rew, done);          // jb_step.hpp: failure flag, reward, time limit, in-place reset, observation
        if (ic->use_policy) {
            PolicyParams<float> pp;
            pp.kick_angle = ic->pp.kick_angle; pp.speed = ic->pp.speed; pp.angle_threshold = ic->pp.angle_threshold;
            ctrl_next = heuristic_policy<float>(task, obs, 1, pp);
        }
        if (grp == 0) {                       // the replica holds the new state and the next action too; only the main lanes write
            const int every = ic->every_step, kk = k - 1;
            float* const obs_out = ic->obs_out;
            if (obs_out && (last || (every & 1))) {
                float* row = obs_out + ((every & 1) ? (size_t)kk * (size_t)N * W : (size_t)0) + (size_t)env * W;
#pragma unroll
                for (int j = 0; j < 19; j++) if ((j & 3) == leg && j < D) row[j] = obs[j];
                if (packed && leg == 0) { row[D] = rew; row[D + 1] = done ? 1.f : 0.f; }
            }
            if (leg == 0 && !packed) {
                float* const reward_out = ic->reward_out;
                unsigned char* const done_out = ic->done_out;
                if (reward_out && (last || (every & 2))) reward_out[((every & 2) ? (size_t)kk * (size_t)N : (size_t)0) + env] = rew;
                if (done_out && (last || (every & 4))) done_out[((every & 4) ? (size_t)kk * (size_t)N : (size_t)0) + env] = done ? 1 : 0;
            }
        }
        if (last) break;
    }
    if (grp != 0) return;
    const auto kb = kernarg_base();
    const KArgsC ka = (KArgsC)kb;
    const StepIOC ic = (StepIOC)(kb + STEPIO_KERNARG_OFFSET);
#ifdef JB_WAVE_STATS
    if (threadIdx.x == 0 && a.wave_stats) {
        unsigned long long* ws = a.wave_stats + (size_t)lblock * 16;
        ws[10] = __builtin_amdgcn_s_memrealtime() - rt_start;
        {   // where the wave ran: [63] = workgroup index << 40 | XCC_ID << 32 | HW_ID (se, sh, cu, simd, wave slot) - tools/debug/placement.py
            const unsigned hw = __builtin_amdgcn_s_getreg(63492), xcc = __builtin_amdgcn_s_getreg(63508);
            a.wave_stats[(size_t)16 * a.n + (size_t)64 * lblock + 63] = ((unsigned long long)blockIdx.x << 40) | ((unsigned long long)(xcc & 0xfu) << 32) | hw;
        }
        ws[0] = __builtin_amdgcn_s_memtime() - t_start;
        for (int i = 0; i < 5; i++) ws[4 + i] = prof_local[i]; ws[1] = (unsigned long long)s.st_xtra; ws[2] = (unsigned long long)s.st_sweeps; ws[3] = (unsigned long long)s.st_contact; ws[11] = (unsigned long long)s.st_slots; ws[15] = (unsigned long long)s.st_fast; ws[9] = (unsigned long long)s.st_checks; ws[12] = prof_local[5]; ws[13] = prof_local[6]; ws[14] = prof_local[7];
    }
#endif
    {
        KArgs af;                 // the fields store_state reads
        af.n = ka->n; af.root = ka->root; af.leg = ka->leg;
        store_state(af, env, lane, leg, s);
        if (leg == 0) {
            ka->step_count[env] = er.step_count;
            ka->episode[env] = er.episode;
            af.root[(RF_TGT + 0) * af.n + env] = er.tgt[0]; af.root[(RF_TGT + 1) * af.n + env] = er.tgt[1]; af.root[(RF_TGT + 2) * af.n + env] = er.tgt[2];
        }
    }
    unsigned long long* const wc = ic->wave_clock;
    if (threadIdx.x == 0 && wc) wc[lblock] = __builtin_amdgcn_s_memrealtime() - clock0;
}

template <int EPW>
__global__ __launch_bounds__(64) void jb_step_kernel(KArgs a, StepIO io) {
    step_body<EPW, false>(a, io);
}
// The LEAN variant: the same substep with the register budget of TWO waves per SIMD (256 registers per lane).  Model constants are read
// from LDS where they are used, and the lane state, the joint-space system and the kept factorisation are parked in the lane's
// scratch between the phases that use them (jb_sim.hpp SimOpts::lean).  The same algorithm, but NOT the same bits as the one-wave kernel:
// the damping factorisation comes from star_factor<false> instead of the replica group's star_solve stream, and the compiler fuses
// multiply-adds differently in the two instantiations - the results agree to fp32 rounding (on the host, without contraction, they are
// identical).  Every shard of one batch must therefore run the same variant (jitterbug_amd.variants resolves it from the global batch).
// A second resident wave doubles the SIMD's VALU issue rate, which pays when a GPU holds >= 2048 waves.
template <int EPW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void jb_step_kernel_lean(KArgs a, StepIO io) {
    step_body<EPW, true>(a, io);
}

// The PAIR variant: the same step with the one geom-geom contact randomised models need (jb_sim.hpp pair_narrow / pair_rows_build,
// the shoulder - motor cross term in the star solve).  A separate instantiation so that the nominal model's kernel stays exactly
// the code it was; chosen per handle (launch_step).
template <int EPW>
__global__ __launch_bounds__(64) void jb_step_kernel_pair(KArgs a, StepIO io) {
    step_body<EPW, false, true>(a, io);
}

// LEAN + PAIR: one model per env on the two-waves-per-SIMD kernel (BASELINE configs[4]'s 8192-env shard).  Four 3 KB tables per wave do not
// fit next to the scratch, so only the entries of the common path are staged (LaneConsts split mode): 25 KB per wave, six waves per CU.
template <int EPW>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(2, 2))) void jb_step_kernel_lean_pair(KArgs a, StepIO io) {
    step_body<EPW, true, true>(a, io);
}

// Self-check of the kernarg-segment reads above, run once per process before the first handle is handed out: a kernel with the step kernels'
// argument list compares what it reads through kernarg_base() with the arguments it received by value.
__global__ void jb_kernarg_probe_kernel(KArgs a, StepIO io, int* ok) {
    const auto kb = kernarg_base();
    const KArgsC ka = (KArgsC)kb;
    const StepIOC ic = (StepIOC)(kb + STEPIO_KERNARG_OFFSET);
    *ok = (ka->n == a.n && ka->seed == a.seed && ka->root == a.root && ka->step_limit == a.step_limit && ic->n_steps == io.n_steps && ic->obs_out == io.obs_out &&
           ic->every_step == io.every_step && ic->wave_clock == io.wave_clock && ic->pp.angle_threshold == io.pp.angle_threshold) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------- reset / observe
__global__ __launch_bounds__(64) void jb_reset_kernel(KArgs a, const unsigned char* __restrict__ mask, float* __restrict__ obs_out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int env = t >> 2, leg = t & 3;
    if (env >= a.n) return;
    const int lane = env * 4 + leg;
    const float* tab = a.lane_model + (a.per_env_model ? (size_t)env * LM_TABLE : 0);
    const float target_z = tab[lm_offset(LM_TARGET_Z, leg)], root_z0 = tab[lm_offset(LM_ROOT_Z0, leg)];
    const float c0[3] = {tab[lm_offset(LM_C0, leg)], tab[lm_offset(LM_C0 + 1, leg)], tab[lm_offset(LM_C0 + 2, leg)]};
    LaneState<float> s;
    EnvCore<float> e;
    e.cx = c0[0]; e.cy = c0[1]; e.cz = c0[2];
    if (!mask || mask[env]) {
        unsigned ep = a.episode[env];
        episode_reset<float>(a.task, a.random_pose, a.seed, a.env_offset + (unsigned long long)env, ep, root_z0, e);
        state_from_reset(e, s);
        s.fail = a.root[RF_FAIL * a.n + env];
        store_state(a, env, lane, leg, s);
        if (leg == 0) { a.episode[env] = ep + 1; a.step_count[env] = 0; store_target(a, env, e); }
    } else {
        load_state(a, env, lane, s);
        core_from_state(a, env, s, c0, e);
    }
    write_obs(a, env, leg, e, target_z, obs_out);
}

__global__ __launch_bounds__(64) void jb_observe_kernel(KArgs a, float* __restrict__ obs_out, float* __restrict__ reward_out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int env = t >> 2, leg = t & 3;
    if (env >= a.n) return;
    const float* tab = a.lane_model + (a.per_env_model ? (size_t)env * LM_TABLE : 0);
    const float target_z = tab[lm_offset(LM_TARGET_Z, leg)];
    const float c0[3] = {tab[lm_offset(LM_C0, leg)], tab[lm_offset(LM_C0 + 1, leg)], tab[lm_offset(LM_C0 + 2, leg)]};
    LaneState<float> s;
    load_state(a, env, env * 4 + leg, s);
    EnvCore<float> e;
    core_from_state(a, env, s, c0, e);
    write_obs(a, env, leg, e, target_z, obs_out);
    if (leg == 0 && reward_out) reward_out[env] = reward<float>(a.task, e, target_z);
}

// ---------------------------------------------------------------------------------------------- heuristic policies
__global__ void jb_policy_kernel(int n, int task, PolicyParams<float> pp, const float* __restrict__ obs, float* __restrict__ action) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= n) return;
    action[env] = heuristic_policy<float>(task, obs + (size_t)env * obs_dim(task), 1, pp);
}

// the four reward terms [P, H, V, U] of every env (reference jitterbug.py:840-889), whatever the handle's task
__global__ __launch_bounds__(64) void jb_reward_terms_kernel(KArgs a, float* __restrict__ terms_out) {
    int t = blockIdx.x * blockDim.x + threadIdx.x;
    int env = t >> 2, leg = t & 3;
    if (env >= a.n) return;
    const float* tab = a.lane_model + (a.per_env_model ? (size_t)env * LM_TABLE : 0);
    const float target_z = tab[lm_offset(LM_TARGET_Z, leg)];
    const float c0[3] = {tab[lm_offset(LM_C0, leg)], tab[lm_offset(LM_C0 + 1, leg)], tab[lm_offset(LM_C0 + 2, leg)]};
    LaneState<float> s;
    load_state(a, env, env * 4 + leg, s);
    EnvCore<float> e;
    core_from_state(a, env, s, c0, e);
    float out[4];
    reward_terms<float>(e, target_z, out);
    terms_out[(size_t)env * 4 + leg] = out[leg];
}

// ---------------------------------------------------------------------------------------------- witness for the unsimulated geom pairs
// One thread per env (fp64, private memory: a diagnostic pass, jb_witness.hpp): smallest distance over the geom pairs MuJoCo would test and the
// step kernels do not collide, in the env's current state.  clear_out / pair_out: this pass's result; min_acc / count_acc (nullable): running
// minimum and the number of passes that found a pair interpenetrating (JB_FLAG_PAIR_WITNESS: one pass behind every step launch).
__global__ __launch_bounds__(64) void jb_witness_kernel(KArgs a, float* __restrict__ clear_out, int* __restrict__ pair_out, float* __restrict__ min_acc, unsigned* __restrict__ count_acc) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= a.n) return;
    const float* tab = a.lane_model + (a.per_env_model ? (size_t)env * LM_TABLE : 0);
    const int L = 4 * a.n;
    double th1[4], th2[4];
    for (int l = 0; l < 4; l++) { th1[l] = a.leg[LF_TH1 * L + env * 4 + l]; th2[l] = a.leg[LF_TH2 * L + env * 4 + l]; }
    int pair[2];
    const double d = witness_clearance<float>(tab, th1, th2, (double)a.root[RF_PHI * a.n + env], pair);
    if (clear_out) clear_out[env] = (float)d;
    if (pair_out) { pair_out[2 * env] = pair[0]; pair_out[2 * env + 1] = pair[1]; }
    if (min_acc) min_acc[env] = fminf(min_acc[env], (float)d);
    if (count_acc && !(d > 0.0)) count_acc[env] += 1u;
}

// ---------------------------------------------------------------------------------------------- launch order of the waves
// order[0 .. n) = the waves sorted by their last measured lifetime, longest first (a 1024-bin counting sort: exact order inside a bin does
// not matter).  One workgroup; runs on the handle's stream right after a step launch whenever the batch has more waves than the device
// holds at once.  All-zero clocks (nothing measured yet) give the identity.
// fold_from > 0 (two-waves-per-SIMD kernels whose batch fits the device at once, but only with both wave slots of some SIMDs in use): the
// hardware gives workgroups b and b + fold_from the same SIMD (measured, tools/debug/placement.py: all 1024 SIMDs, index distance 1024),
// so the sorted list is FOLDED: with n <= 2 fold_from waves the 2 fold_from - n longest get a SIMD to themselves, and of the others the
// longest shares its SIMD with the shortest, the second longest with the second shortest.
__global__ __launch_bounds__(1024) void jb_wave_order_kernel(const unsigned long long* __restrict__ clock, int* __restrict__ order, int n, int fold_from) {
    __shared__ unsigned long long s_max[1024];
    __shared__ unsigned s_bin[1024], s_pos[1024];
    const int t = threadIdx.x;
    unsigned long long mx = 0;
    for (int i = t; i < n; i += 1024) mx = clock[i] > mx ? clock[i] : mx;
    s_max[t] = mx; s_bin[t] = 0u;
    __syncthreads();
    for (int st = 512; st > 0; st >>= 1) { if (t < st) s_max[t] = s_max[t] > s_max[t + st] ? s_max[t] : s_max[t + st]; __syncthreads(); }
    mx = s_max[0];
    if (mx == 0ull) { for (int i = t; i < n; i += 1024) order[i] = i; return; }
    const double scale = 1023.0 / (double)mx;
    for (int i = t; i < n; i += 1024) atomicAdd(&s_bin[1023 - (int)((double)clock[i] * scale)], 1u);      // bin 0 = the longest waves
    __syncthreads();
    if (t == 0) { unsigned acc = 0; for (int k = 0; k < 1024; k++) { s_pos[k] = acc; acc += s_bin[k]; } }
    __syncthreads();
    for (int i = t; i < n; i += 1024) {
        int p = (int)atomicAdd(&s_pos[1023 - (int)((double)clock[i] * scale)], 1u);
        if (fold_from > 0) {      // S = fold_from SIMDs, n <= 2 S waves: the 2 S - n longest run alone, the rest pair longest with shortest
            const int S = fold_from, alone = 2 * S - n;
            p = p < alone ? (n - S) + p : p < S ? p - alone : S + (n - 1 - p);
        }
        order[p] = i;
    }
}

// ---------------------------------------------------------------------------------------------- diagnostics
// Fills LDS with NaN bit patterns (LDS keeps its contents between kernels): a step kernel that reads scratch it has not
// written in the same launch then produces NaNs deterministically instead of depending on what ran before it.
__global__ void jb_poison_lds_kernel(int n_floats, float* sink) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < n_floats; i += blockDim.x) lds[i] = __uint_as_float(0x7FC00000u + (unsigned)(i & 0xFFFF));
    __syncthreads();
    if (sink && lds[(threadIdx.x * 97) % n_floats] == 0.f) sink[0] = 1.f;     // keep the stores alive
}

// ---------------------------------------------------------------------------------------------- observation encoder (tiny dense network per row)
struct EncArgs {
    int n, n_layers, vae, in_dim, out_dim;
    int dims[JB_ENC_MAX_LAYERS + 1], acts[JB_ENC_MAX_LAYERS], woff[JB_ENC_MAX_LAYERS], boff[JB_ENC_MAX_LAYERS];
    const float* params;          // weights then biases
    unsigned long long seed, env_offset;
    unsigned call;
};
__global__ void jb_encode_kernel(EncArgs e, const float* __restrict__ obs, float* __restrict__ out) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= e.n) return;
    float a[JB_ENC_MAX_WIDTH], b[JB_ENC_MAX_WIDTH];
    for (int i = 0; i < e.in_dim; i++) a[i] = obs[(size_t)env * e.in_dim + i];
    for (int l = 0; l < e.n_layers; l++) {
        const int din = e.dims[l], dout = e.dims[l + 1];
        const float* W = e.params + e.woff[l];
        const float* bias = e.params + e.boff[l];
        for (int j = 0; j < dout; j++) {
            float t = bias[j];
            for (int i = 0; i < din; i++) t = fmaf(a[i], W[i * dout + j], t);
            b[j] = e.acts[l] == JB_ACT_TANH ? tanhf(t) : e.acts[l] == JB_ACT_RELU ? fmaxf(t, 0.f) : t;
        }
        for (int j = 0; j < dout; j++) a[j] = b[j];
    }
    if (e.vae) {   // code = mean + std * eps, eps ~ N(0,1) by Box-Muller on Philox words
        const int L = e.out_dim;
        for (int j0 = 0; j0 < L; j0 += 4) {
            uint32_t r[4];
            philox4x32(e.seed ^ 0x656E636F64657273ull, e.env_offset + (unsigned long long)env, e.call, (uint32_t)(j0 >> 2), r);
            float u0 = ((float)(r[0] >> 8) + 0.5f) * (1.0f / 16777216.0f), u1 = ((float)(r[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            float u2 = ((float)(r[2] >> 8) + 0.5f) * (1.0f / 16777216.0f), u3 = ((float)(r[3] >> 8) + 0.5f) * (1.0f / 16777216.0f);
            float ra = sqrtf(-2.f * logf(u0)), rb = sqrtf(-2.f * logf(u2));
            float eps[4] = {ra * cosf(6.2831853f * u1), ra * sinf(6.2831853f * u1), rb * cosf(6.2831853f * u3), rb * sinf(6.2831853f * u3)};
            for (int k = 0; k < 4 && j0 + k < L; k++) out[(size_t)env * L + j0 + k] = a[j0 + k] + a[L + j0 + k] * eps[k];
        }
    } else {
        for (int j = 0; j < e.out_dim; j++) out[(size_t)env * e.out_dim + j] = a[j];
    }
}

// ---------------------------------------------------------------------------------------------- state import / export (fp64 MuJoCo layout)
__global__ void jb_export_kernel(KArgs a, double* __restrict__ qpos, double* __restrict__ qvel, double* __restrict__ target) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= a.n) return;
    const int N = a.n, L = 4 * N;
    const float* r = a.root + env;
    if (qpos) {
        double* q = qpos + (size_t)env * 16;
        for (int i = 0; i < 7; i++) q[i] = r[i * N];
        for (int i = 0; i < 5; i++) q[2 + i] += (double)r[(RF_LO + i) * N];          // height and quaternion are held as hi + lo
        for (int l = 0; l < 4; l++) { q[7 + 2 * l] = a.leg[LF_TH1 * L + env * 4 + l]; q[8 + 2 * l] = a.leg[LF_TH2 * L + env * 4 + l]; }
        q[15] = (double)r[RF_PHI * N] + 6.283185307179586 * (double)r[RF_TURNS * N];
    }
    if (qvel) {
        double* v = qvel + (size_t)env * 15;
        for (int i = 0; i < 6; i++) v[i] = r[(RF_V + i) * N];
        for (int l = 0; l < 4; l++) { v[6 + 2 * l] = a.leg[LF_THD1 * L + env * 4 + l]; v[7 + 2 * l] = a.leg[LF_THD2 * L + env * 4 + l]; }
        v[14] = r[RF_PHID * N];
    }
    if (target) for (int i = 0; i < 3; i++) target[(size_t)env * 3 + i] = r[(RF_TGT + i) * N];
}
__global__ void jb_import_kernel(KArgs a, const double* __restrict__ qpos, const double* __restrict__ qvel, const double* __restrict__ target) {
    int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= a.n) return;
    const int N = a.n, L = 4 * N;
    float* r = a.root + env;
    if (qpos) {
        const double* q = qpos + (size_t)env * 16;
        for (int i = 0; i < 7; i++) r[i * N] = (float)q[i];
        for (int i = 0; i < 5; i++) r[(RF_LO + i) * N] = (float)(q[2 + i] - (double)(float)q[2 + i]);
        for (int l = 0; l < 4; l++) { a.leg[LF_TH1 * L + env * 4 + l] = (float)q[7 + 2 * l]; a.leg[LF_TH2 * L + env * 4 + l] = (float)q[8 + 2 * l]; }
        double k = floor((q[15] + 3.141592653589793) / 6.283185307179586);
        r[RF_PHI * N] = (float)(q[15] - k * 6.283185307179586); r[RF_TURNS * N] = (float)k;
    }
    if (qvel) {
        const double* v = qvel + (size_t)env * 15;
        for (int i = 0; i < 6; i++) r[(RF_V + i) * N] = (float)v[i];
        for (int l = 0; l < 4; l++) { a.leg[LF_THD1 * L + env * 4 + l] = (float)v[6 + 2 * l]; a.leg[LF_THD2 * L + env * 4 + l] = (float)v[7 + 2 * l]; }
        r[RF_PHID * N] = (float)v[14];
    }
    if (target) for (int i = 0; i < 3; i++) r[(RF_TGT + i) * N] = (float)target[(size_t)env * 3 + i];
    // a teacher-forced state has no history: clear the contact-solver warm start
    for (int i = 0; i < 7; i++) r[(RF_WA + i) * N] = 0.f;
    for (int l = 0; l < 4; l++) { a.leg[LF_WJ0 * L + env * 4 + l] = 0.f; a.leg[LF_WJ1 * L + env * 4 + l] = 0.f; }
}

// ---------------------------------------------------------------------------------------------- domain randomisation (one thread per env)
static_assert(AO_COUNT == JB_NOFFSET, "offset vector layout: header and compiler disagree");
struct RndArgs {
    int n, flags, max_attempts;
    unsigned long long seed, env_offset;
    AugSigmas sd;
    double min_mass_clearance;
    const JbNominalSpec* spec;            // nominal model, device memory
    const double* offsets_in;             // [n, AO_COUNT] nullable: use these instead of drawing
    float* tables;                        // [n, LM_TABLE] lane constant tables (the kernel's per-env model)
    double* params_out;                   // [n, JB_NPARAM] nullable
    double* offsets_out;                  // [n, AO_COUNT] nullable
    int* attempts_out;                    // [n] nullable
    int* status;                          // [1]: first failure code (0 = ok)
};
__global__ __launch_bounds__(64) void jb_randomise_kernel(RndArgs a) {
    const int env = blockIdx.x * blockDim.x + threadIdx.x;
    if (env >= a.n) return;
    JbNominalSpec S;
    double P[JB_NPARAM], off[AO_COUNT];
    int attempt = 0, rc = 0;
    for (;;) {
        if (a.offsets_in) for (int i = 0; i < AO_COUNT; i++) off[i] = a.offsets_in[(size_t)env * AO_COUNT + i];
        else draw_offsets(a.seed, a.env_offset + (unsigned long long)env, (uint32_t)attempt, a.flags, a.sd, off);
        apply_offsets(*a.spec, a.flags, off, S);
        rc = compile_model(S, P);
        attempt++;
        if (rc == 0 && (a.offsets_in || a.min_mass_clearance <= 0.0 || mass_sweep_clear(P, a.min_mass_clearance))) break;
        if (a.offsets_in || attempt >= a.max_attempts) { if (rc == 0) rc = -30; break; }      // -30: no acceptable draw within max_attempts
    }
    if (rc == 0) rc = build_packed_model<float>(P, a.tables + (size_t)env * LM_TABLE);
    if (rc != 0) atomicCAS(a.status, 0, rc);
    if (a.params_out) for (int i = 0; i < JB_NPARAM; i++) a.params_out[(size_t)env * JB_NPARAM + i] = P[i];
    if (a.offsets_out) for (int i = 0; i < AO_COUNT; i++) a.offsets_out[(size_t)env * AO_COUNT + i] = off[i];
    if (a.attempts_out) a.attempts_out[env] = attempt;
}

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define JB_HIP(call)                                                                                     \
    do {                                                                                                 \
        hipError_t _e = (call);                                                                          \
        if (_e != hipSuccess) return fail(JB_E_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); \
    } while (0)

}  // namespace

struct jb_handle {
    jb_config cfg;
    KArgs ka;
    int D;
    hipStream_t stream;
    bool own_stream;
    float *d_root, *d_leg, *d_model, *d_ovc;
    int* d_step; unsigned* d_episode;
    // staging for the host-buffer entry points
    float *d_action, *d_obs, *d_reward; unsigned char *d_done, *d_mask;
    double *d_qpos, *d_qvel, *d_target;
    unsigned long long* d_wave_stats;
    unsigned long long* d_wave_clock;
    unsigned long long* d_resolve;    // [1]: KArgs::resolve_count
    float* d_capture; unsigned* d_capture_count;      // diagnostic builds (-DJB_CAPTURE)
    float *d_tape, *d_rows_stage, *d_rew_stage;       // staging of the host-buffer rollouts (jb_step_many, jb_rollout_policy): grow-only, freed in jb_destroy
    size_t tape_cap, rows_cap, rew_cap;               //   their capacities in floats
    int* d_wave_order;               // launch order of the waves (jb_wave_order_kernel), null while the device holds the whole batch at once
    int wave_slots;                  // waves the device holds at once with this handle's kernel variant
    size_t model_tables;
    EncArgs enc;          // observation encoder (n_layers = 0: none)
    PolicyParams<float> policy;      // keyword arguments of the reference's heuristic policies
    JbNominalSpec* d_spec;           // nominal (uncompiled) model for the randomiser
    void* comm;                      // RCCL communicator (jb_comm_init), null until asked for
    int comm_ranks, comm_rank;
    int comm_nmax;                   // envs of the longest shard (jb_comm_set_shards); 0: every rank holds cfg.n_envs
    // jb_step_async / jb_step_wait: pinned host staging (allocated at the first jb_step_async) and the event behind the D2H copies
    float *p_action, *p_obs, *p_reward; unsigned char* p_done;
    hipEvent_t async_done;
    bool async_pending;
    float* d_terms;
    float* d_enc_params; float* d_code;
    float *d_wit_clear, *d_wit_min; int* d_wit_pair; unsigned* d_wit_count;      // the pair witness (jb_pair_witness / JB_FLAG_PAIR_WITNESS), allocated at first use
};

// ---------------------------------------------------------------------------------------------- RCCL, bound at run time
// The library has no link-time dependency on RCCL: a host that never calls jb_comm_* never loads it, and a process that already
// holds an RCCL (PyTorch-ROCm ships its own copy) keeps exactly that one - dlopen finds the loaded image first.
namespace {
struct RcclApi {
    struct IdBlob { char b[128]; };          // ncclUniqueId (NCCL_UNIQUE_ID_BYTES = 128), passed BY VALUE to ncclCommInitRank
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, IdBlob, int) = nullptr;
    int (*CommDestroy)(void*) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
};
RcclApi g_rccl;
int load_rccl() {
    if (g_rccl.lib) return JB_OK;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void* lib = nullptr;
    for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;          // an image the process already holds
    if (!lib) for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (!lib) return fail(JB_E_HIP, std::string("RCCL is not available (librccl.so): ") + (dlerror() ? dlerror() : ""));
    auto sym = [&](const char* n) { return dlsym(lib, n); };
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))sym("ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))sym("ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))sym("ncclCommDestroy");
    g_rccl.GroupStart = (decltype(g_rccl.GroupStart))sym("ncclGroupStart");
    g_rccl.GroupEnd = (decltype(g_rccl.GroupEnd))sym("ncclGroupEnd");
    g_rccl.Send = (decltype(g_rccl.Send))sym("ncclSend");
    g_rccl.Recv = (decltype(g_rccl.Recv))sym("ncclRecv");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))sym("ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.GroupStart || !g_rccl.GroupEnd || !g_rccl.Send || !g_rccl.Recv)
        return fail(JB_E_HIP, "librccl.so lacks an expected symbol");
    g_rccl.lib = lib;
    return JB_OK;
}
#define JB_NCCL(call)                                                                                                      \
    do {                                                                                                                   \
        int _r = (call);                                                                                                   \
        if (_r != 0) return fail(JB_E_HIP, std::string(#call) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "RCCL error")); \
    } while (0)
}  // namespace

// ---------------------------------------------------------------------------------------------- roctx ranges, bound at run time
// Named ranges around step / reset / gather so that a rocprofv3 --marker-trace timeline of the N > 1 path shows where the row gather
// sits relative to the step kernel (tracing row).  Like RCCL the library is bound by dlopen: an image the process
// already holds (the profiler's, or PyTorch's libroctx64) is used as is; otherwise it is loaded only when JB_ROCTX=1 asks for it.
// Without it the ranges cost one predictable branch.
namespace {
struct RoctxApi {
    int state = 0;          // 0: not looked for yet, 1: bound, -1: unavailable
    int (*Push)(const char*) = nullptr;
    int (*Pop)() = nullptr;
};
RoctxApi g_roctx;
void load_roctx() {
    const char* names[] = {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"};
    void* lib = nullptr;
    for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    const char* want = getenv("JB_ROCTX");
    if (!lib && want && want[0] == '1') for (const char* n : names) if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    if (lib) {
        g_roctx.Push = (decltype(g_roctx.Push))dlsym(lib, "roctxRangePushA");
        g_roctx.Pop = (decltype(g_roctx.Pop))dlsym(lib, "roctxRangePop");
    }
    g_roctx.state = (g_roctx.Push && g_roctx.Pop) ? 1 : -1;
}
struct RoctxRange {          // RAII: popped on every return path
    bool on;
    explicit RoctxRange(const char* name) {
        if (g_roctx.state == 0) load_roctx();
        on = g_roctx.state == 1;
        if (on) g_roctx.Push(name);
    }
    ~RoctxRange() { if (on) g_roctx.Pop(); }
    RoctxRange(const RoctxRange&) = delete;
    RoctxRange& operator=(const RoctxRange&) = delete;
};
}  // namespace

// Every entry point that allocates or launches runs with the handle's device current (a process may hold handles on several
// GPUs, or switch devices with torch.cuda.set_device after jb_create) and hands the caller's current device back on EVERY return
// path (jb_device_guard.hpp): torch reads the current device through hipGetDevice, so a switch that leaked out of an entry
// point would send the caller's next allocation or launch to the wrong GPU.
struct HipDeviceApi {
    static int get(int* d) { return hipGetDevice(d) == hipSuccess ? 0 : 1; }
    static int set(int d) { return hipSetDevice(d) == hipSuccess ? 0 : 1; }
};
#define JB_ENTER(h)                                                                                                   \
    if (!(h)) return fail(JB_E_INVALID, "handle is NULL");                                                            \
    jb::DeviceGuard<HipDeviceApi> _jb_device_guard;                                                                   \
    if (_jb_device_guard.enter((h)->cfg.device_id) != 0) return fail(JB_E_HIP, "hipSetDevice(" + std::to_string((h)->cfg.device_id) + ") failed")

static dim3 grid_lanes(int n) { return dim3((unsigned)(((size_t)n * 4 + 63) / 64)); }

static int ensure_model_buffer(jb_handle* h, int n_tables) {
    if ((size_t)n_tables != h->model_tables) {
        if (h->d_model) JB_HIP(hipFree(h->d_model));
        h->d_model = nullptr; h->model_tables = 0;
        JB_HIP(hipMalloc(&h->d_model, (size_t)n_tables * LM_TABLE * sizeof(float)));
        h->model_tables = n_tables;
    }
    return JB_OK;
}
// Which step kernel a handle runs.  JB_FLAG_LEAN is honoured where a LEAN instantiation exists: a shared model without the pair contact
// (envs per wave 1, 2 or 4), or one model per env at four envs per wave (LEAN + PAIR, split tables).  Anything else that asks for LEAN is
// refused (JB_E_INVALID) at the call that creates the combination - never silently run as the ordinary kernel.
static int kernel_variant(const jb_handle* h) {
    const bool lean_pair = h->ka.lean && h->ka.pair && h->ka.per_env_model && h->ka.epw == 4;
    if (lean_pair) return JB_VARIANT_LEAN_PAIR;
    if (h->ka.pair) return JB_VARIANT_PAIR;
    if (h->ka.lean) return JB_VARIANT_LEAN;
    return JB_VARIANT_ORDINARY;
}
static int check_variant(const jb_handle* h) {
    if (h->ka.lean && h->ka.pair && !(h->ka.per_env_model && h->ka.epw == 4))
        return fail(JB_E_INVALID, "JB_FLAG_LEAN cannot be honoured: the pair-contact kernel has a two-waves-per-SIMD form only for one model per env at 4 envs per wave "
                                  "(this handle: " + std::string(h->ka.per_env_model ? "one model per env" : "a shared model that needs the pair contact") + ", " + std::to_string(h->ka.epw) +
                                  " envs per wave); drop JB_FLAG_LEAN, or add JB_FLAG_NO_PAIR for floor contacts only");
    return JB_OK;
}
static int upload_model(jb_handle* h, const double* params, int n_tables) {
    std::vector<float> host((size_t)n_tables * LM_TABLE);
    for (int t = 0; t < n_tables; t++) {
        int rc = build_packed_model<float>(params + (size_t)t * JB_NPARAM, host.data() + (size_t)t * LM_TABLE);
        if (rc) return fail(JB_E_MODEL, "parameter table " + std::to_string(t) + " not supported by the kernel (code " + std::to_string(rc) + ")");
    }
    // which step kernel: the PAIR variant whenever the model(s) may bring the mass against a leg (see JB_FLAG_PAIR)
    const int pair = (h->cfg.flags & JB_FLAG_NO_PAIR) ? 0 : ((h->cfg.flags & JB_FLAG_PAIR) || n_tables > 1 || !mass_sweep_clear(params, 5e-4)) ? 1 : 0;
    {   // a model that JB_FLAG_LEAN cannot run is refused BEFORE anything changes (the handle keeps the model it had)
        jb_handle probe = *h;
        probe.ka.pair = pair; probe.ka.per_env_model = n_tables > 1 ? 1 : 0;
        int rc = check_variant(&probe);
        if (rc) return rc;
    }
    { int rc = ensure_model_buffer(h, n_tables); if (rc) return rc; }
    JB_HIP(hipMemcpyAsync(h->d_model, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice, h->stream));
    JB_HIP(hipStreamSynchronize(h->stream));
    h->ka.lane_model = h->d_model;
    h->ka.per_env_model = n_tables > 1 ? 1 : 0;
    h->ka.pair = pair;
    return JB_OK;
}

extern "C" {

const char* jb_last_error(void) { return g_err.c_str(); }
int jb_abi_version(void) { return JB_ABI_VERSION; }
// sha256 of the sources (csrc/*, include/*, compiler flags) this library was built from; jitterbug_amd/build.py passes it in and reads it
// back from the file (the tag in front makes it findable without loading the library)
#ifndef JB_SRC_SHA
#define JB_SRC_SHA "0000000000000000000000000000000000000000000000000000000000000000"
#endif
static const char JB_SRC_TAG[] = "JB_SRC_SHA256=" JB_SRC_SHA;
const char* jb_source_sha256(void) { return JB_SRC_TAG + 14; }
const double* jb_default_model_params(void) { return JB_DEFAULT_PARAMS; }
int jb_obs_dim(int32_t task_id) { return (task_id >= 0 && task_id < JB_NTASK) ? obs_dim(task_id) : JB_E_INVALID; }
int jb_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
int jb_default_config(jb_config* cfg, int32_t n_envs, int32_t task_id) {
    if (!cfg) return fail(JB_E_INVALID, "cfg is NULL");
    std::memset(cfg, 0, sizeof *cfg);
    cfg->n_envs = n_envs; cfg->task_id = task_id; cfg->device_id = 0; cfg->random_pose = 1; cfg->contacts = 1;
    cfg->substeps = 50; cfg->step_limit = 1000; cfg->auto_reset = 1; cfg->max_newton = JB_DEFAULT_MAX_NEWTON; cfg->seed = 0; cfg->env_offset = 0; cfg->stream = nullptr;
    return JB_OK;
}

static int create_impl(jb_handle* h) {      // every failure returns through jb_create, which destroys the partly built handle
    const jb_config* cfg = &h->cfg;
    const size_t N = (size_t)cfg->n_envs;
    if (cfg->use_caller_stream) { h->stream = (hipStream_t)cfg->stream; h->own_stream = false; }
    else { JB_HIP(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking)); h->own_stream = true; }
    JB_HIP(hipMalloc(&h->d_root, sizeof(float) * ROOT_F * N));
    JB_HIP(hipMalloc(&h->d_leg, sizeof(float) * LEG_F * 4 * N));
    JB_HIP(hipMalloc(&h->d_step, sizeof(int) * N));
    JB_HIP(hipMalloc(&h->d_episode, sizeof(unsigned) * N));
    JB_HIP(hipMalloc(&h->d_action, sizeof(float) * N));
    JB_HIP(hipMalloc(&h->d_obs, sizeof(float) * N * h->D));
    JB_HIP(hipMalloc(&h->d_reward, sizeof(float) * N));
    JB_HIP(hipMalloc(&h->d_done, N));
    JB_HIP(hipMalloc(&h->d_mask, N));
    JB_HIP(hipMalloc(&h->d_qpos, sizeof(double) * 16 * N));
    JB_HIP(hipMalloc(&h->d_qvel, sizeof(double) * 15 * N));
    JB_HIP(hipMalloc(&h->d_target, sizeof(double) * 3 * N));
    JB_HIP(hipMalloc(&h->d_terms, sizeof(float) * 4 * N));
    JB_HIP(hipMalloc(&h->d_resolve, sizeof(unsigned long long)));
#ifdef JB_CAPTURE
    JB_HIP(hipMalloc(&h->d_capture, sizeof(float) * 64 * JB_CAPTURE_SLOTS));
    JB_HIP(hipMalloc(&h->d_capture_count, sizeof(unsigned)));
    JB_HIP(hipMemsetAsync(h->d_capture_count, 0, sizeof(unsigned), h->stream));
#endif
    JB_HIP(hipMemsetAsync(h->d_resolve, 0, sizeof(unsigned long long), h->stream));
#ifdef JB_WAVE_STATS
    JB_HIP(hipMalloc(&h->d_wave_stats, sizeof(unsigned long long) * (size_t)(16 + 64) * N));
    JB_HIP(hipMemset(h->d_wave_stats, 0, sizeof(unsigned long long) * (size_t)(16 + 64) * N));
#endif
    JB_HIP(hipMemsetAsync(h->d_root, 0, sizeof(float) * ROOT_F * N, h->stream));
    JB_HIP(hipMemsetAsync(h->d_leg, 0, sizeof(float) * LEG_F * 4 * N, h->stream));
    JB_HIP(hipMemsetAsync(h->d_step, 0, sizeof(int) * N, h->stream));
    JB_HIP(hipMemsetAsync(h->d_episode, 0, sizeof(unsigned) * N, h->stream));
    KArgs& k = h->ka;
    k.n = cfg->n_envs; k.task = cfg->task_id; k.substeps = cfg->substeps; k.step_limit = cfg->step_limit; k.auto_reset = cfg->auto_reset;
    k.contacts = cfg->contacts; k.max_newton = h->cfg.max_newton; k.random_pose = cfg->random_pose; k.per_env_model = 0;
    k.seed = cfg->seed; k.env_offset = cfg->env_offset; k.rank_one = (cfg->flags & JB_FLAG_NO_RANK_ONE) ? 0 : 1; k.spread = (cfg->flags & JB_FLAG_NO_SPREAD) ? 0 : 1;
    k.lean = (cfg->flags & JB_FLAG_LEAN) ? 1 : 0;
    {   // envs per wave: fill every SIMD of the device before filling the lanes of a wave.  The kernel holds one wave
        // per SIMD (register budget), so the device runs (CUs x 4) waves at a time; LDS (scratch is per active lane)
        // allows 4 resident waves per CU up to 8 envs per wave.
        int epw = cfg->envs_per_wave;
        if (epw <= 0) {
            hipDeviceProp_t prop;
            JB_HIP(hipGetDeviceProperties(&prop, cfg->device_id));
            const int simds = prop.multiProcessorCount * 4;
            epw = (cfg->n_envs + simds - 1) / simds;
            if (epw > 4) epw = 4;     // LDS: 4 resident waves per CU need <= 40 KB each
        }
        if (epw < 1) epw = 1;
        if (epw > 8) epw = 8;                // (a 16-env wave would need 97 KB of LDS - one wave per CU - and spilled registers: not instantiated)
        while (epw & (epw - 1)) epw++;       // 1, 2, 4 or 8 (the kernel is instantiated for these)
        if (k.lean) { if (cfg->envs_per_wave <= 0) epw = 4; if (epw > 4) epw = 4; }     // LEAN: 4 envs per wave, 20 KB of LDS each: 8 waves per CU, two per SIMD
        k.epw = epw;
    }
    k.root = h->d_root; k.leg = h->d_leg; k.step_count = h->d_step; k.episode = h->d_episode; k.wave_stats = h->d_wave_stats; k.resolve_count = h->d_resolve; k.capture = h->d_capture; k.capture_count = h->d_capture_count;
    {   // the step kernels read part of their arguments through the kernarg segment: make sure that layout is what they assume (once per
        // process; handles created from several threads at once are serialised here)
        static std::mutex probe_mutex;
        static bool probed = false;
        std::lock_guard<std::mutex> lock(probe_mutex);
        if (!probed) {
            int* d_ok = nullptr;
            JB_HIP(hipMalloc(&d_ok, sizeof(int)));
            KArgs pa = KArgs(); StepIO pio = StepIO();
            pa.n = 0x1234567; pa.seed = 0x0123456789ABCDEFull; pa.root = (float*)0x10002000; pa.step_limit = 777;
            pio.n_steps = 4242; pio.obs_out = (float*)0x30004000; pio.every_step = 5; pio.wave_clock = (unsigned long long*)0x50006000; pio.pp.angle_threshold = 0.625f;
            hipLaunchKernelGGL(jb_kernarg_probe_kernel, dim3(1), dim3(1), 0, h->stream, pa, pio, d_ok);
            int ok = 0;
            const hipError_t e1 = hipGetLastError(), e2 = hipMemcpyAsync(&ok, d_ok, sizeof(int), hipMemcpyDeviceToHost, h->stream), e3 = hipStreamSynchronize(h->stream);
            hipFree(d_ok);          // (on every path: nothing below returns before this)
            if (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) return fail(JB_E_HIP, "kernarg probe failed to run");
            if (!ok) return fail(JB_E_HIP, "the step kernels' view of the kernarg segment does not match their arguments (compiler ABI change?): rebuild is needed with the layout fixed");
            probed = true;
        }
    }
    int rc = upload_model(h, JB_DEFAULT_PARAMS, 1);
    if (rc) return rc;
    rc = jb_reset_device(h, nullptr, nullptr);     // every env starts in a valid episode-0 state
    if (rc) return rc;
    JB_HIP(hipStreamSynchronize(h->stream));
    return JB_OK;
}

int jb_create(const jb_config* cfg, jb_handle** out) {
    if (!cfg || !out) return fail(JB_E_INVALID, "cfg/out is NULL");
    *out = nullptr;
    if (cfg->n_envs < 1) return fail(JB_E_INVALID, "n_envs must be >= 1");
    if (cfg->task_id < 0 || cfg->task_id >= JB_NTASK) return fail(JB_E_INVALID, "unknown task_id");
    if (cfg->substeps < 1 || cfg->step_limit < 1) return fail(JB_E_INVALID, "substeps and step_limit must be >= 1");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(JB_E_NODEVICE, "no HIP device available (there is no CPU fallback)");
    if (cfg->device_id < 0 || cfg->device_id >= ndev) return fail(JB_E_INVALID, "device_id out of range");
    jb::DeviceGuard<HipDeviceApi> guard;              // the caller's current device is restored on every return path
    if (guard.enter(cfg->device_id) != 0) return fail(JB_E_HIP, "hipSetDevice(" + std::to_string(cfg->device_id) + ") failed");
    jb_handle* h = new (std::nothrow) jb_handle();
    if (!h) return fail(JB_E_INVALID, "out of host memory");
    std::memset(h, 0, sizeof *h);
    h->cfg = *cfg;
    if (h->cfg.max_newton <= 0) h->cfg.max_newton = JB_DEFAULT_MAX_NEWTON;
    h->D = obs_dim(cfg->task_id);
    h->policy = default_policy_params<float>();
    const int rc = create_impl(h);
    if (rc) {                               // one cleanup path: stream, every buffer allocated so far, the handle (keep the message)
        const std::string msg = g_err;
        jb_destroy(h);
        g_err = msg;
        return rc;
    }
    *out = h;
    return JB_OK;
}

int jb_destroy(jb_handle* h) {
    if (!h) return JB_OK;
    jb::DeviceGuard<HipDeviceApi> guard;
    guard.enter(h->cfg.device_id);
    if (h->stream || !h->own_stream) hipStreamSynchronize(h->stream);
    if (h->comm && g_rccl.CommDestroy) { hipDeviceSynchronize(); g_rccl.CommDestroy(h->comm); h->comm = nullptr; }      // (device-wide: exchanges may be queued on streams of the caller's)
    void* bufs[] = {h->d_tape, h->d_rows_stage, h->d_rew_stage, h->d_capture, h->d_capture_count, h->d_resolve, h->d_wave_order, h->d_wave_clock, h->d_ovc, h->d_terms, h->d_spec, h->d_root, h->d_leg, h->d_model, h->d_step, h->d_episode, h->d_action, h->d_obs, h->d_reward, h->d_done, h->d_mask, h->d_qpos, h->d_qvel, h->d_target, h->d_wave_stats, h->d_enc_params, h->d_code, h->d_wit_clear, h->d_wit_min, h->d_wit_pair, h->d_wit_count};
    for (void* b : bufs) if (b) hipFree(b);
    void* pinned[] = {h->p_action, h->p_obs, h->p_reward, h->p_done};
    for (void* b : pinned) if (b) hipHostFree(b);
    if (h->async_done) hipEventDestroy(h->async_done);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return JB_OK;
}

int jb_set_obs_encoder(jb_handle* h, int32_t n_layers, const int32_t* dims, const int32_t* acts, const float* weights, const float* biases, int32_t vae) {
    JB_ENTER(h);
    JB_HIP(hipStreamSynchronize(h->stream));
    if (h->d_enc_params) { hipFree(h->d_enc_params); h->d_enc_params = nullptr; }
    if (h->d_code) { hipFree(h->d_code); h->d_code = nullptr; }
    h->enc = EncArgs();
    if (n_layers == 0) return JB_OK;
    if (n_layers < 0 || n_layers > JB_ENC_MAX_LAYERS || !dims || !acts || !weights || !biases) return fail(JB_E_INVALID, "encoder: 1.." + std::to_string(JB_ENC_MAX_LAYERS) + " layers with dims/acts/weights/biases");
    if (dims[0] != h->D) return fail(JB_E_INVALID, "encoder: dims[0] must be the observation width " + std::to_string(h->D));
    size_t nw = 0, nb = 0;
    for (int l = 0; l < n_layers; l++) {
        if (dims[l] < 1 || dims[l] > JB_ENC_MAX_WIDTH || dims[l + 1] < 1 || dims[l + 1] > JB_ENC_MAX_WIDTH) return fail(JB_E_INVALID, "encoder: layer widths must be 1.." + std::to_string(JB_ENC_MAX_WIDTH));
        if (acts[l] < JB_ACT_LINEAR || acts[l] > JB_ACT_RELU) return fail(JB_E_INVALID, "encoder: unknown activation");
        nw += (size_t)dims[l] * dims[l + 1]; nb += (size_t)dims[l + 1];
    }
    if (vae && (dims[n_layers] & 1)) return fail(JB_E_INVALID, "encoder: a VAE head needs an even last width [mean | std]");
    EncArgs e = EncArgs();
    e.n = h->cfg.n_envs; e.n_layers = n_layers; e.vae = vae ? 1 : 0; e.in_dim = dims[0]; e.out_dim = vae ? dims[n_layers] / 2 : dims[n_layers];
    size_t wo = 0, bo = nw;
    for (int l = 0; l < n_layers; l++) {
        e.dims[l] = dims[l]; e.acts[l] = acts[l]; e.woff[l] = (int)wo; e.boff[l] = (int)bo;
        wo += (size_t)dims[l] * dims[l + 1]; bo += (size_t)dims[l + 1];
    }
    e.dims[n_layers] = dims[n_layers];
    e.seed = h->cfg.seed; e.env_offset = h->cfg.env_offset; e.call = 0;
    std::vector<float> host(nw + nb);
    std::copy(weights, weights + nw, host.begin());
    std::copy(biases, biases + nb, host.begin() + nw);
    JB_HIP(hipMalloc(&h->d_enc_params, host.size() * sizeof(float)));
    JB_HIP(hipMalloc(&h->d_code, sizeof(float) * (size_t)e.n * e.out_dim));
    JB_HIP(hipMemcpy(h->d_enc_params, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    e.params = h->d_enc_params;
    h->enc = e;
    return JB_OK;
}
int jb_encoded_dim(jb_handle* h) { return h ? h->enc.out_dim * (h->enc.n_layers > 0) : JB_E_INVALID; }
int jb_encode_device(jb_handle* h, const float* d_obs, float* d_code_out) {
    if (!h || !d_obs || !d_code_out) return fail(JB_E_INVALID, "handle/obs/out is NULL");
    JB_ENTER(h);
    if (h->enc.n_layers <= 0) return fail(JB_E_INVALID, "no observation encoder set (jb_set_obs_encoder)");
    const int N = h->cfg.n_envs;
    hipLaunchKernelGGL(jb_encode_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, h->stream, h->enc, d_obs, d_code_out);
    JB_HIP(hipGetLastError());
    h->enc.call++;
    return JB_OK;
}
int jb_encode(jb_handle* h, const float* obs, float* code_out) {
    if (!h || !obs || !code_out) return fail(JB_E_INVALID, "handle/obs/out is NULL");
    JB_ENTER(h);
    if (h->enc.n_layers <= 0) return fail(JB_E_INVALID, "no observation encoder set (jb_set_obs_encoder)");
    const size_t N = (size_t)h->cfg.n_envs;
    JB_HIP(hipMemcpyAsync(h->d_obs, obs, sizeof(float) * N * h->D, hipMemcpyHostToDevice, h->stream));
    int rc = jb_encode_device(h, h->d_obs, h->d_code);
    if (rc) return rc;
    JB_HIP(hipMemcpyAsync(code_out, h->d_code, sizeof(float) * N * h->enc.out_dim, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipStreamSynchronize(h->stream));
    return JB_OK;
}
int jb_debug_poison_lds(jb_handle* h) {
    JB_ENTER(h);
    hipDeviceProp_t prop;
    JB_HIP(hipGetDeviceProperties(&prop, h->cfg.device_id));
    const int bytes = 40 * 1024;                                   // 4 resident blocks cover a CU's 160 KB
    hipLaunchKernelGGL(jb_poison_lds_kernel, dim3((unsigned)prop.multiProcessorCount * 8), dim3(256), bytes, h->stream, bytes / 4, (float*)nullptr);
    JB_HIP(hipGetLastError());
    return JB_OK;
}
int jb_num_envs(jb_handle* h) { return h ? h->cfg.n_envs : JB_E_INVALID; }
void* jb_stream(jb_handle* h) { return h ? (void*)h->stream : nullptr; }
int jb_synchronize(jb_handle* h) {
    JB_ENTER(h);
    JB_HIP(hipStreamSynchronize(h->stream));
    return JB_OK;
}

int jb_reset_device(jb_handle* h, const uint8_t* d_mask, float* d_obs_out) {
    JB_ENTER(h);
    RoctxRange range("jb_reset");
    hipLaunchKernelGGL(jb_reset_kernel, grid_lanes(h->cfg.n_envs), dim3(64), 0, h->stream, h->ka, d_mask, d_obs_out);
    JB_HIP(hipGetLastError());
    return JB_OK;
}
static int ensure_witness(jb_handle* h) {
    if (h->d_wit_clear) return JB_OK;
    const size_t N = (size_t)h->cfg.n_envs;
    JB_HIP(hipMalloc(&h->d_wit_clear, sizeof(float) * N));
    JB_HIP(hipMalloc(&h->d_wit_min, sizeof(float) * N));
    JB_HIP(hipMalloc(&h->d_wit_pair, sizeof(int) * 2 * N));
    JB_HIP(hipMalloc(&h->d_wit_count, sizeof(unsigned) * N));
    std::vector<float> inf(N, INFINITY);
    JB_HIP(hipMemcpyAsync(h->d_wit_min, inf.data(), sizeof(float) * N, hipMemcpyHostToDevice, h->stream));
    JB_HIP(hipMemsetAsync(h->d_wit_count, 0, sizeof(unsigned) * N, h->stream));
    JB_HIP(hipStreamSynchronize(h->stream));          // (`inf` goes out of scope)
    return JB_OK;
}
static int launch_step(jb_handle* h, StepIO io, int packed_rows) {
    if (!h) return fail(JB_E_INVALID, "handle is NULL");
    if (io.n_steps < 1) return fail(JB_E_INVALID, "n_steps must be >= 1");
    if (!io.use_policy && !io.actions) return fail(JB_E_INVALID, "action buffer is NULL");
    JB_ENTER(h);
    { int rc = check_variant(h); if (rc) return rc; }
    RoctxRange range(io.n_steps > 1 ? "jb_step_many" : "jb_step");
    h->ka.packed_rows = packed_rows;
    io.pp = h->policy;
    const dim3 grid((unsigned)((h->cfg.n_envs + h->ka.epw - 1) / h->ka.epw));
    if (!h->d_wave_clock) {
        JB_HIP(hipMalloc(&h->d_wave_clock, sizeof(unsigned long long) * (size_t)h->cfg.n_envs));      // (>= the number of waves for any envs-per-wave)
        JB_HIP(hipMemsetAsync(h->d_wave_clock, 0, sizeof(unsigned long long) * (size_t)h->cfg.n_envs, h->stream));
    }
    io.wave_clock = h->d_wave_clock;
    const int variant = kernel_variant(h);
    // more waves than the device holds at once -> launch them longest first (the order comes from the previous launch's clocks)
    bool reorder = false;
    int fold_from = 0;
    if (!(h->cfg.flags & JB_FLAG_NO_REORDER)) {
        if (h->wave_slots == 0) {
            hipDeviceProp_t prop;
            JB_HIP(hipGetDeviceProperties(&prop, h->cfg.device_id));
            h->wave_slots = prop.multiProcessorCount * 4;
        }
        const int per_simd_x2 = (variant == JB_VARIANT_LEAN || variant == JB_VARIANT_LEAN_PAIR) ? 4 : 2;      // resident waves per SIMD, times two
        reorder = (long long)grid.x * 2 > (long long)h->wave_slots * per_simd_x2;
        // two waves per SIMD and the whole batch resident: no launch ORDER to choose, but who shares a SIMD with whom (jb_wave_order_kernel)
        if (per_simd_x2 == 4 && (int)grid.x > h->wave_slots && !reorder) { reorder = true; fold_from = h->wave_slots; }
        if (reorder && !h->d_wave_order) {
            JB_HIP(hipMalloc(&h->d_wave_order, sizeof(int) * (size_t)h->cfg.n_envs));
            hipLaunchKernelGGL(jb_wave_order_kernel, dim3(1), dim3(1024), 0, h->stream, h->d_wave_clock, h->d_wave_order, (int)grid.x, fold_from);      // (clocks all zero: identity)
            JB_HIP(hipGetLastError());
        }
    }
    io.wave_order = reorder ? h->d_wave_order : nullptr;
    const bool lean_pair = variant == JB_VARIANT_LEAN_PAIR, use_lean = lean_pair || variant == JB_VARIANT_LEAN;
    const bool aux_bodies = variant == JB_VARIANT_ORDINARY && h->ka.epw <= 4;      // (step_body: AUX - an aux block behind every staged table)
    const size_t lds_bytes = lean_pair ? ((size_t)SC_COUNT_LEAN_PAIR * 4 * h->ka.epw + (size_t)LM_SPLIT_RES * h->ka.epw) * sizeof(float)
                                       : ((size_t)(use_lean ? SC_COUNT_LEAN : SC_COUNT) * 4 * h->ka.epw + ((size_t)(use_lean ? LM_TABLE_BASE : LM_TABLE) + (aux_bodies ? LM_AUX : 0)) * (h->ka.per_env_model ? h->ka.epw : 1)) * sizeof(float);
#ifdef JB_DEBUG
    static const size_t extra_lds = getenv("JB_DEBUG_EXTRA_LDS") ? (size_t)atoi(getenv("JB_DEBUG_EXTRA_LDS")) : 0;      // occupancy experiments (-DJB_DEBUG builds only)
    const size_t lds_bytes_x = lds_bytes + extra_lds;
#else
    const size_t lds_bytes_x = lds_bytes;
#endif
    if (use_lean && !h->d_ovc) {      // the LEAN variant's overflow candidates (beyond the row cache): one block per wave
        const size_t waves = (size_t)grid.x, fl = waves * OVC_FLOATS_PER_LANE * 4 * h->ka.epw;
        JB_HIP(hipMalloc(&h->d_ovc, fl * sizeof(float)));
        h->ka.ovc_buf = h->d_ovc;
    }
#define JB_LAUNCH_STEP(E) hipLaunchKernelGGL(jb_step_kernel<E>, grid, dim3(64), lds_bytes_x, h->stream, h->ka, io)
#define JB_LAUNCH_LEAN(E) hipLaunchKernelGGL(jb_step_kernel_lean<E>, grid, dim3(64), lds_bytes_x, h->stream, h->ka, io)
#define JB_LAUNCH_PAIR(E) hipLaunchKernelGGL(jb_step_kernel_pair<E>, grid, dim3(64), lds_bytes_x, h->stream, h->ka, io)
#ifdef JB_DEV_ONLY4      // development builds: the one instantiation the headline runs (a fifth of the compile time)
    if (variant != JB_VARIANT_ORDINARY || h->ka.epw != 4) return fail(JB_E_INVALID, "JB_DEV_ONLY4 build: only the ordinary kernel at 4 envs per wave");
    JB_LAUNCH_STEP(4);
#else
    if (lean_pair) {
        hipLaunchKernelGGL(jb_step_kernel_lean_pair<4>, grid, dim3(64), lds_bytes_x, h->stream, h->ka, io);
    } else if (variant == JB_VARIANT_PAIR) {
        switch (h->ka.epw) {
        case 1: JB_LAUNCH_PAIR(1); break;
        case 2: JB_LAUNCH_PAIR(2); break;
        case 4: JB_LAUNCH_PAIR(4); break;
        default: JB_LAUNCH_PAIR(8); break;
        }
    } else if (use_lean) {
        switch (h->ka.epw) {
        case 1: JB_LAUNCH_LEAN(1); break;
        case 2: JB_LAUNCH_LEAN(2); break;
        default: JB_LAUNCH_LEAN(4); break;
        }
    } else {
        switch (h->ka.epw) {
        case 1: JB_LAUNCH_STEP(1); break;
        case 2: JB_LAUNCH_STEP(2); break;
        case 4: JB_LAUNCH_STEP(4); break;
        default: JB_LAUNCH_STEP(8); break;
        }
    }
#endif
#undef JB_LAUNCH_STEP
#undef JB_LAUNCH_LEAN
#undef JB_LAUNCH_PAIR
    JB_HIP(hipGetLastError());
    if (reorder) {
        hipLaunchKernelGGL(jb_wave_order_kernel, dim3(1), dim3(1024), 0, h->stream, h->d_wave_clock, h->d_wave_order, (int)grid.x, fold_from);
        JB_HIP(hipGetLastError());
    }
    if (h->cfg.flags & JB_FLAG_PAIR_WITNESS) {      // one witness pass behind every step launch (the state the launch left)
        int rc = ensure_witness(h);
        if (rc) return rc;
        hipLaunchKernelGGL(jb_witness_kernel, dim3((unsigned)((h->cfg.n_envs + 63) / 64)), dim3(64), 0, h->stream, h->ka, (float*)nullptr, (int*)nullptr, h->d_wit_min, h->d_wit_count);
        JB_HIP(hipGetLastError());
    }
    return JB_OK;
}
int jb_kernel_variant(jb_handle* h) { return h ? kernel_variant(h) : JB_E_INVALID; }
int jb_envs_per_wave(jb_handle* h) { return h ? h->ka.epw : JB_E_INVALID; }
int jb_step_device(jb_handle* h, const float* d_action, float* d_obs_out, float* d_reward_out, uint8_t* d_done_out) {
    if (!d_action) return fail(JB_E_INVALID, "handle/action is NULL");
    StepIO io = StepIO();
    io.n_steps = 1; io.actions = d_action; io.obs_out = d_obs_out; io.reward_out = d_reward_out; io.done_out = d_done_out;
    return launch_step(h, io, 0);
}
int jb_step_rows_device(jb_handle* h, const float* d_action, float* d_rows_out) {
    if (!d_action) return fail(JB_E_INVALID, "handle/action is NULL");
    if (!d_rows_out) return fail(JB_E_INVALID, "rows buffer is NULL");
    StepIO io = StepIO();
    io.n_steps = 1; io.actions = d_action; io.obs_out = d_rows_out;
    return launch_step(h, io, 1);
}
// K control steps in ONE launch (see StepIO).  d_actions [K, N], or NULL: the handle's heuristic policy, evaluated in the kernel.
// d_rows_out [K, N, D+2] nullable (packed rows of every step); d_rewards [K, N] nullable; d_obs_last [N, D] / d_done_last [N] nullable:
// the last step's observations / done flags.  Rows and the separate outputs exclude each other (one output layout per launch).
int jb_step_many_device(jb_handle* h, int32_t n_steps, const float* d_actions, float* d_rows_out, float* d_rewards, float* d_obs_last, uint8_t* d_done_last) {
    if (!h) return fail(JB_E_INVALID, "handle is NULL");
    if (n_steps < 0) return fail(JB_E_INVALID, "n_steps < 0");
    if (n_steps == 0) return JB_OK;
    if (d_rows_out && (d_rewards || d_obs_last || d_done_last)) return fail(JB_E_INVALID, "jb_step_many_device: either packed rows [K,N,D+2] or rewards / last observations / last done flags, not both");
    StepIO io = StepIO();
    io.n_steps = n_steps; io.use_policy = d_actions ? 0 : 1; io.actions = d_actions;
    if (d_rows_out) {
        io.obs_out = d_rows_out; io.every_step = 1;
        return launch_step(h, io, 1);
    }
    io.obs_out = d_obs_last; io.reward_out = d_rewards; io.done_out = d_done_last;
    io.every_step = 2;
    return launch_step(h, io, 0);
}
// grow-only device staging owned by the handle: (re)allocated only when a call needs more than any call before it
static int ensure_stage(jb_handle* h, float** buf, size_t* cap, size_t floats, const char* what) {
    if (floats <= *cap) return JB_OK;
    if (*buf) { JB_HIP(hipStreamSynchronize(h->stream)); JB_HIP(hipFree(*buf)); *buf = nullptr; *cap = 0; }
    if (hipMalloc(buf, sizeof(float) * floats) != hipSuccess) { *buf = nullptr; return fail(JB_E_HIP, std::string("out of device memory for ") + what); }
    *cap = floats;
    return JB_OK;
}
// host-buffer form of jb_step_many_device: actions [K, N] (NULL: the in-kernel heuristic policy), rows_out [K, N, D+2] (nullable);
// the device staging belongs to the handle and only ever grows (no allocation after the first call of a given size); the results are valid on return
int jb_step_many(jb_handle* h, int32_t n_steps, const float* actions, float* rows_out) {
    if (!h) return fail(JB_E_INVALID, "handle is NULL");
    if (n_steps < 0) return fail(JB_E_INVALID, "n_steps < 0");
    if (n_steps == 0) return JB_OK;
    JB_ENTER(h);
    const size_t N = (size_t)h->cfg.n_envs, K = (size_t)n_steps, W = (size_t)h->D + 2;
    int rc = JB_OK;
    if (actions) {
        rc = ensure_stage(h, &h->d_tape, &h->tape_cap, K * N, "the action tape");
        if (rc) return rc;
        JB_HIP(hipMemcpyAsync(h->d_tape, actions, sizeof(float) * K * N, hipMemcpyHostToDevice, h->stream));
    }
    if (rows_out) { rc = ensure_stage(h, &h->d_rows_stage, &h->rows_cap, K * N * W, "the rows"); if (rc) return rc; }
    rc = jb_step_many_device(h, n_steps, actions ? h->d_tape : nullptr, rows_out ? h->d_rows_stage : nullptr, nullptr, nullptr, nullptr);
    if (!rc && rows_out && hipMemcpyAsync(rows_out, h->d_rows_stage, sizeof(float) * K * N * W, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = fail(JB_E_HIP, "jb_step_many: copy of the rows failed");
    const hipError_t e = hipStreamSynchronize(h->stream);
    if (!rc && e != hipSuccess) rc = fail(JB_E_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return rc;
}
int jb_release_staging(jb_handle* h) {
    JB_ENTER(h);
    JB_HIP(hipStreamSynchronize(h->stream));
    float** bufs[] = {&h->d_tape, &h->d_rows_stage, &h->d_rew_stage};
    size_t* caps[] = {&h->tape_cap, &h->rows_cap, &h->rew_cap};
    for (int i = 0; i < 3; i++) { if (*bufs[i]) hipFree(*bufs[i]); *bufs[i] = nullptr; *caps[i] = 0; }
    return JB_OK;
}
// how long each wave of the last step launch lived, in seconds (s_memrealtime, 100 MHz): out[0 .. n_waves); returns the number of waves
int jb_wave_clocks(jb_handle* h, double* out, int32_t max_waves) {
    if (!h || !out) return fail(JB_E_INVALID, "handle/out is NULL");
    JB_ENTER(h);
    const int waves = (h->cfg.n_envs + h->ka.epw - 1) / h->ka.epw;
    if (!h->d_wave_clock) return fail(JB_E_INVALID, "no step has been launched yet");
    JB_HIP(hipStreamSynchronize(h->stream));
    std::vector<unsigned long long> t((size_t)waves);
    JB_HIP(hipMemcpy(t.data(), h->d_wave_clock, sizeof(unsigned long long) * (size_t)waves, hipMemcpyDeviceToHost));
    for (int i = 0; i < waves && i < max_waves; i++) out[i] = (double)t[(size_t)i] * 1e-8;
    return waves;
}
int jb_observe_device(jb_handle* h, float* d_obs_out, float* d_reward_out) {
    JB_ENTER(h);
    hipLaunchKernelGGL(jb_observe_kernel, grid_lanes(h->cfg.n_envs), dim3(64), 0, h->stream, h->ka, d_obs_out, d_reward_out);
    JB_HIP(hipGetLastError());
    return JB_OK;
}

int jb_reset(jb_handle* h, const uint8_t* mask, float* obs_out) {
    JB_ENTER(h);
    if (h->async_pending) return fail(JB_E_INVALID, "jb_reset: a jb_step_async is pending (jb_step_wait first)");
    const size_t N = (size_t)h->cfg.n_envs;
    if (mask) JB_HIP(hipMemcpyAsync(h->d_mask, mask, N, hipMemcpyHostToDevice, h->stream));
    int rc = jb_reset_device(h, mask ? h->d_mask : nullptr, obs_out ? h->d_obs : nullptr);
    if (rc) return rc;
    if (obs_out) JB_HIP(hipMemcpyAsync(obs_out, h->d_obs, sizeof(float) * N * h->D, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipStreamSynchronize(h->stream));
    return JB_OK;
}
int jb_step(jb_handle* h, const float* action, float* obs_out, float* reward_out, uint8_t* done_out) {
    if (!h || !action) return fail(JB_E_INVALID, "handle/action is NULL");
    JB_ENTER(h);
    if (h->async_pending) return fail(JB_E_INVALID, "jb_step: a jb_step_async is pending (jb_step_wait first)");
    RoctxRange range("jb_step_host_buffers");
    const size_t N = (size_t)h->cfg.n_envs;
    JB_HIP(hipMemcpyAsync(h->d_action, action, sizeof(float) * N, hipMemcpyHostToDevice, h->stream));
    int rc = jb_step_device(h, h->d_action, h->d_obs, h->d_reward, h->d_done);
    if (rc) return rc;
    if (obs_out) JB_HIP(hipMemcpyAsync(obs_out, h->d_obs, sizeof(float) * N * h->D, hipMemcpyDeviceToHost, h->stream));
    if (reward_out) JB_HIP(hipMemcpyAsync(reward_out, h->d_reward, sizeof(float) * N, hipMemcpyDeviceToHost, h->stream));
    if (done_out) JB_HIP(hipMemcpyAsync(done_out, h->d_done, N, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipStreamSynchronize(h->stream));
    return JB_OK;
}
// jb_step in two halves (header: the VecEnv's step_async / step_wait).  Pinned staging: a copy from or to pageable memory would make
// hipMemcpyAsync wait for the stream; with pinned buffers the call returns as soon as the four operations are queued.
int jb_step_async(jb_handle* h, const float* action) {
    if (!h || !action) return fail(JB_E_INVALID, "handle/action is NULL");
    JB_ENTER(h);
    if (h->async_pending) return fail(JB_E_INVALID, "jb_step_async: the previous step has not been waited for (jb_step_wait)");
    const size_t N = (size_t)h->cfg.n_envs;
    if (!h->p_action) {
        JB_HIP(hipHostMalloc((void**)&h->p_action, sizeof(float) * N, hipHostMallocDefault));
        JB_HIP(hipHostMalloc((void**)&h->p_obs, sizeof(float) * N * h->D, hipHostMallocDefault));
        JB_HIP(hipHostMalloc((void**)&h->p_reward, sizeof(float) * N, hipHostMallocDefault));
        JB_HIP(hipHostMalloc((void**)&h->p_done, N, hipHostMallocDefault));
        JB_HIP(hipEventCreateWithFlags(&h->async_done, hipEventDisableTiming));
    }
    RoctxRange range("jb_step_async");
    std::memcpy(h->p_action, action, sizeof(float) * N);
    JB_HIP(hipMemcpyAsync(h->d_action, h->p_action, sizeof(float) * N, hipMemcpyHostToDevice, h->stream));
    int rc = jb_step_device(h, h->d_action, h->d_obs, h->d_reward, h->d_done);
    if (rc) return rc;
    JB_HIP(hipMemcpyAsync(h->p_obs, h->d_obs, sizeof(float) * N * h->D, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipMemcpyAsync(h->p_reward, h->d_reward, sizeof(float) * N, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipMemcpyAsync(h->p_done, h->d_done, N, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipEventRecord(h->async_done, h->stream));
    h->async_pending = true;
    return JB_OK;
}
int jb_step_wait(jb_handle* h, float* obs_out, float* reward_out, uint8_t* done_out) {
    if (!h) return fail(JB_E_INVALID, "handle is NULL");
    if (!h->async_pending) return fail(JB_E_INVALID, "jb_step_wait: no jb_step_async is pending");
    JB_ENTER(h);
    h->async_pending = false;          // (whatever happens below, the pairing async -> wait is used up)
    JB_HIP(hipEventSynchronize(h->async_done));
    const size_t N = (size_t)h->cfg.n_envs;
    if (obs_out) std::memcpy(obs_out, h->p_obs, sizeof(float) * N * h->D);
    if (reward_out) std::memcpy(reward_out, h->p_reward, sizeof(float) * N);
    if (done_out) std::memcpy(done_out, h->p_done, N);
    return JB_OK;
}
int jb_step_views(jb_handle* h, const float** obs, const float** reward, const uint8_t** done) {
    if (!h) return fail(JB_E_INVALID, "handle is NULL");
    if (!h->p_action) return fail(JB_E_INVALID, "jb_step_views: no jb_step_async yet (the pinned buffers come into being with the first one)");
    if (obs) *obs = h->p_obs;
    if (reward) *reward = h->p_reward;
    if (done) *done = h->p_done;
    return JB_OK;
}
int jb_observe(jb_handle* h, float* obs_out, float* reward_out) {
    if (!h || !obs_out) return fail(JB_E_INVALID, "handle/obs_out is NULL");
    JB_ENTER(h);
    const size_t N = (size_t)h->cfg.n_envs;
    int rc = jb_observe_device(h, h->d_obs, reward_out ? h->d_reward : nullptr);
    if (rc) return rc;
    JB_HIP(hipMemcpyAsync(obs_out, h->d_obs, sizeof(float) * N * h->D, hipMemcpyDeviceToHost, h->stream));
    if (reward_out) JB_HIP(hipMemcpyAsync(reward_out, h->d_reward, sizeof(float) * N, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipStreamSynchronize(h->stream));
    return JB_OK;
}
int jb_policy_device(jb_handle* h, const float* d_obs, float* d_action) {
    if (!h || !d_obs || !d_action) return fail(JB_E_INVALID, "handle/obs/action is NULL");
    JB_ENTER(h);
    const int N = h->cfg.n_envs;
    hipLaunchKernelGGL(jb_policy_kernel, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, h->stream, N, h->cfg.task_id, h->policy, d_obs, d_action);
    JB_HIP(hipGetLastError());
    return JB_OK;
}
// K control steps of "heuristic policy -> step" (the loop of the reference's benchmarks/evaluate_policy.py:29-33, for every env of the
// batch at once) in ONE launch: the policy is evaluated in the step kernel on the observation the lanes just produced (step 0: on the
// caller's rows).  d_rewards (nullable): [K, N] per-step rewards.
int jb_rollout_policy_device(jb_handle* h, int32_t n_steps, float* d_obs_inout /*[N,D]: current observations in, last out*/,
                             float* d_rewards /*[K,N] nullable*/, uint8_t* d_done_last /*[N] nullable*/) {
    if (!h || !d_obs_inout || n_steps < 0) return fail(JB_E_INVALID, "handle/obs is NULL or n_steps < 0");
    if (n_steps == 0) return JB_OK;
    StepIO io = StepIO();
    io.n_steps = n_steps; io.use_policy = 1; io.obs_in = d_obs_inout;
    io.obs_out = d_obs_inout;
    io.reward_out = d_rewards ? d_rewards : h->d_reward; io.every_step = d_rewards ? 2 : 0;
    io.done_out = d_done_last ? d_done_last : h->d_done;
    return launch_step(h, io, 0);
}
int jb_rollout_policy(jb_handle* h, int32_t n_steps, float* rewards_out /*[K,N] host, nullable*/, float* obs_out /*[N,D] host, nullable*/) {
    if (!h || n_steps < 0) return fail(JB_E_INVALID, "handle is NULL or n_steps < 0");
    JB_ENTER(h);
    const size_t N = (size_t)h->cfg.n_envs;
    const bool want_rew = rewards_out && n_steps > 0;
    if (want_rew) { int rc0 = ensure_stage(h, &h->d_rew_stage, &h->rew_cap, N * (size_t)n_steps, "the rewards"); if (rc0) return rc0; }
    int rc = jb_observe_device(h, h->d_obs, nullptr);
    if (!rc) rc = jb_rollout_policy_device(h, n_steps, h->d_obs, want_rew ? h->d_rew_stage : nullptr, nullptr);
    if (!rc && want_rew && hipMemcpyAsync(rewards_out, h->d_rew_stage, sizeof(float) * N * (size_t)n_steps, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = fail(JB_E_HIP, "copy of rewards failed");
    if (!rc && obs_out && hipMemcpyAsync(obs_out, h->d_obs, sizeof(float) * N * h->D, hipMemcpyDeviceToHost, h->stream) != hipSuccess) rc = fail(JB_E_HIP, "copy of observations failed");
    hipError_t e = hipStreamSynchronize(h->stream);
    if (!rc && e != hipSuccess) rc = fail(JB_E_HIP, std::string("hipStreamSynchronize: ") + hipGetErrorString(e));
    return rc;
}
int jb_policy(jb_handle* h, const float* obs, float* action) {
    if (!h || !obs || !action) return fail(JB_E_INVALID, "handle/obs/action is NULL");
    JB_ENTER(h);
    const size_t N = (size_t)h->cfg.n_envs;
    JB_HIP(hipMemcpyAsync(h->d_obs, obs, sizeof(float) * N * h->D, hipMemcpyHostToDevice, h->stream));
    int rc = jb_policy_device(h, h->d_obs, h->d_action);
    if (rc) return rc;
    JB_HIP(hipMemcpyAsync(action, h->d_action, sizeof(float) * N, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipStreamSynchronize(h->stream));
    return JB_OK;
}
int jb_set_policy_params(jb_handle* h, float kick_angle, float speed, float angle_threshold) {
    if (!h) return fail(JB_E_INVALID, "handle is NULL");
    if (!(kick_angle >= 0.f) || !(angle_threshold >= 0.f) || !std::isfinite(speed)) return fail(JB_E_INVALID, "policy parameters: kick_angle, angle_threshold >= 0 and a finite speed");
    h->policy.kick_angle = kick_angle; h->policy.speed = speed; h->policy.angle_threshold = angle_threshold;
    return JB_OK;
}
int jb_reward_terms_device(jb_handle* h, float* d_terms_out) {
    if (!h || !d_terms_out) return fail(JB_E_INVALID, "handle/terms is NULL");
    JB_ENTER(h);
    hipLaunchKernelGGL(jb_reward_terms_kernel, grid_lanes(h->cfg.n_envs), dim3(64), 0, h->stream, h->ka, d_terms_out);
    JB_HIP(hipGetLastError());
    return JB_OK;
}
int jb_reward_terms(jb_handle* h, float* terms_out) {
    if (!h || !terms_out) return fail(JB_E_INVALID, "handle/terms is NULL");
    JB_ENTER(h);
    int rc = jb_reward_terms_device(h, h->d_terms);
    if (rc) return rc;
    JB_HIP(hipMemcpyAsync(terms_out, h->d_terms, sizeof(float) * 4 * (size_t)h->cfg.n_envs, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipStreamSynchronize(h->stream));
    return JB_OK;
}
int jb_get_state(jb_handle* h, double* qpos, double* qvel, double* target) {
    JB_ENTER(h);
    const size_t N = (size_t)h->cfg.n_envs;
    hipLaunchKernelGGL(jb_export_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, h->stream, h->ka, qpos ? h->d_qpos : nullptr, qvel ? h->d_qvel : nullptr,
                       target ? h->d_target : nullptr);
    JB_HIP(hipGetLastError());
    if (qpos) JB_HIP(hipMemcpyAsync(qpos, h->d_qpos, sizeof(double) * 16 * N, hipMemcpyDeviceToHost, h->stream));
    if (qvel) JB_HIP(hipMemcpyAsync(qvel, h->d_qvel, sizeof(double) * 15 * N, hipMemcpyDeviceToHost, h->stream));
    if (target) JB_HIP(hipMemcpyAsync(target, h->d_target, sizeof(double) * 3 * N, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipStreamSynchronize(h->stream));
    return JB_OK;
}
int jb_set_state(jb_handle* h, const double* qpos, const double* qvel, const double* target) {
    JB_ENTER(h);
    const size_t N = (size_t)h->cfg.n_envs;
    if (qpos) JB_HIP(hipMemcpyAsync(h->d_qpos, qpos, sizeof(double) * 16 * N, hipMemcpyHostToDevice, h->stream));
    if (qvel) JB_HIP(hipMemcpyAsync(h->d_qvel, qvel, sizeof(double) * 15 * N, hipMemcpyHostToDevice, h->stream));
    if (target) JB_HIP(hipMemcpyAsync(h->d_target, target, sizeof(double) * 3 * N, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(jb_import_kernel, dim3((unsigned)((N + 127) / 128)), dim3(128), 0, h->stream, h->ka, qpos ? h->d_qpos : nullptr, qvel ? h->d_qvel : nullptr,
                       target ? h->d_target : nullptr);
    JB_HIP(hipGetLastError());
    JB_HIP(hipStreamSynchronize(h->stream));
    return JB_OK;
}
int jb_get_counters(jb_handle* h, int32_t* step_count, uint32_t* episode, float* solver_cap_hits) {
    JB_ENTER(h);
    const size_t N = (size_t)h->cfg.n_envs;
    JB_HIP(hipStreamSynchronize(h->stream));
    if (step_count) JB_HIP(hipMemcpy(step_count, h->d_step, sizeof(int) * N, hipMemcpyDeviceToHost));
    if (episode) JB_HIP(hipMemcpy(episode, h->d_episode, sizeof(unsigned) * N, hipMemcpyDeviceToHost));
    if (solver_cap_hits) JB_HIP(hipMemcpy(solver_cap_hits, h->d_root + RF_FAIL * N, sizeof(float) * N, hipMemcpyDeviceToHost));
    return JB_OK;
}
// how often the contact solve needed its second, line-searched pass: wave-substeps since jb_create (jb_sim.hpp newton_phase<LS = true>)
int jb_solver_stats(jb_handle* h, uint64_t* resolved_wave_substeps) {
    if (!h || !resolved_wave_substeps) return fail(JB_E_INVALID, "handle/out is NULL");
    JB_ENTER(h);
    JB_HIP(hipStreamSynchronize(h->stream));
    unsigned long long v = 0;
    JB_HIP(hipMemcpy(&v, h->d_resolve, sizeof v, hipMemcpyDeviceToHost));
    *resolved_wave_substeps = (uint64_t)v;
    return JB_OK;
}
#ifdef JB_CAPTURE
// diagnostic builds only: the captured records (SimOpts::capture), out[max_records][64]; returns how many substeps stayed unconverged
int jb_debug_captured(jb_handle* h, float* out, int32_t max_records) {
    if (!h || !out) return fail(JB_E_INVALID, "NULL");
    JB_HIP(hipStreamSynchronize(h->stream));
    unsigned n = 0;
    JB_HIP(hipMemcpy(&n, h->d_capture_count, sizeof n, hipMemcpyDeviceToHost));
    const int m = (int)n < max_records ? (int)n : max_records;
    JB_HIP(hipMemcpy(out, h->d_capture, sizeof(float) * 64 * (size_t)(m < JB_CAPTURE_SLOTS ? m : JB_CAPTURE_SLOTS), hipMemcpyDeviceToHost));
    return (int)n;
}
#endif
#ifdef JB_WAVE_STATS
// diagnostic builds only: per-wave [cycles, rare-path substeps, Newton sweeps, contact substeps] of the last step launch
int jb_debug_wave_stats(jb_handle* h, unsigned long long* out, int32_t n_waves) {
    if (!h || !out) return fail(JB_E_INVALID, "NULL");
    JB_HIP(hipStreamSynchronize(h->stream));
    JB_HIP(hipMemcpy(out, h->d_wave_stats, sizeof(unsigned long long) * 16 * n_waves, hipMemcpyDeviceToHost));
    return h->ka.epw;
}
#endif
// The witness for the geom pairs the step kernels do not collide (header; jb_witness.hpp): one pass over the current state, results on the host.
int jb_pair_witness(jb_handle* h, float* clearance_out, int32_t* pairs_out, int32_t* n_touching) {
    JB_ENTER(h);
    int rc = ensure_witness(h);
    if (rc) return rc;
    const size_t N = (size_t)h->cfg.n_envs;
    hipLaunchKernelGGL(jb_witness_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, h->stream, h->ka, h->d_wit_clear, h->d_wit_pair, (float*)nullptr, (unsigned*)nullptr);
    JB_HIP(hipGetLastError());
    std::vector<float> host(N);
    JB_HIP(hipMemcpyAsync(host.data(), h->d_wit_clear, sizeof(float) * N, hipMemcpyDeviceToHost, h->stream));
    if (pairs_out) JB_HIP(hipMemcpyAsync(pairs_out, h->d_wit_pair, sizeof(int) * 2 * N, hipMemcpyDeviceToHost, h->stream));
    JB_HIP(hipStreamSynchronize(h->stream));
    int touching = 0;
    for (size_t i = 0; i < N; i++) { if (!(host[i] > 0.f)) touching++; if (clearance_out) clearance_out[i] = host[i]; }
    if (n_touching) *n_touching = touching;
    return JB_OK;
}
int jb_get_pair_witness(jb_handle* h, uint32_t* overlap_passes, float* min_clearance) {
    JB_ENTER(h);
    if (!(h->cfg.flags & JB_FLAG_PAIR_WITNESS)) return fail(JB_E_INVALID, "jb_get_pair_witness: the handle was created without JB_FLAG_PAIR_WITNESS (jb_pair_witness gives one pass on demand)");
    int rc = ensure_witness(h);
    if (rc) return rc;
    const size_t N = (size_t)h->cfg.n_envs;
    JB_HIP(hipStreamSynchronize(h->stream));
    if (overlap_passes) JB_HIP(hipMemcpy(overlap_passes, h->d_wit_count, sizeof(unsigned) * N, hipMemcpyDeviceToHost));
    if (min_clearance) JB_HIP(hipMemcpy(min_clearance, h->d_wit_min, sizeof(float) * N, hipMemcpyDeviceToHost));
    return JB_OK;
}
int jb_set_model_params(jb_handle* h, const double* params, int32_t n_tables) {
    if (!h || !params) return fail(JB_E_INVALID, "handle/params is NULL");
    JB_ENTER(h);
    if (n_tables != 1 && n_tables != h->cfg.n_envs) return fail(JB_E_INVALID, "n_tables must be 1 or n_envs");
    JB_HIP(hipStreamSynchronize(h->stream));
    return upload_model(h, params, n_tables);
}


int jb_default_randomise_config(jb_randomise_cfg* c) {
    if (!c) return fail(JB_E_INVALID, "cfg is NULL");
    std::memset(c, 0, sizeof *c);
    c->flags = JB_RND_LEGS | JB_RND_MASS;                 // what the reference's training harness switches on (benchmarks/benchmark.py:130-136)
    c->max_attempts = 64; c->seed = 0;
    c->sd_legs[0] = 0.003; c->sd_legs[1] = 0.003; c->sd_legs[2] = 0.002;                    // reference augmented_jitterbug.py:96
    c->sd_mass_pos[0] = 0.0015; c->sd_mass_pos[1] = 0.002; c->sd_mass_pos[2] = 0.001;      // :98
    c->sd_core1_density = 10.0; c->sd_core2_density = 80.0; c->sd_global_density = 200.0; c->sd_gear = 0.001;     // :100-107
    c->min_mass_clearance = 0.0;
    return JB_OK;
}
int jb_randomise_models(jb_handle* h, const jb_randomise_cfg* cfg, const double* offsets_in, double* params_out, double* offsets_out, int32_t* attempts_out) {
    if (!h || !cfg) return fail(JB_E_INVALID, "handle/cfg is NULL");
    JB_ENTER(h);
    const size_t N = (size_t)h->cfg.n_envs;
    {   // one model per env runs the PAIR kernel: refuse up front what JB_FLAG_LEAN cannot run (see check_variant)
        jb_handle probe = *h;
        probe.ka.per_env_model = 1; probe.ka.pair = (h->cfg.flags & JB_FLAG_NO_PAIR) ? 0 : 1;
        int rc0 = check_variant(&probe);
        if (rc0) return rc0;
    }
    JB_HIP(hipStreamSynchronize(h->stream));
    if (!h->d_spec) {
        JB_HIP(hipMalloc(&h->d_spec, sizeof(JbNominalSpec)));
        JB_HIP(hipMemcpy(h->d_spec, &JB_NOMINAL_SPEC, sizeof(JbNominalSpec), hipMemcpyHostToDevice));
    }
    // The tables are generated into a FRESH buffer and swapped into the handle only after the kernel succeeded: a failure on the
    // way (the 5 KB x N params_out allocation is the likeliest) leaves the handle on the model it had, never on freed or
    // half-written tables.
    float* d_tables = nullptr;
    double *d_off_in = nullptr, *d_par = nullptr, *d_off_out = nullptr; int *d_att = nullptr, *d_status = nullptr;
    int rc = JB_OK, status = 0;
    auto cleanup = [&]() { if (d_tables) hipFree(d_tables); if (d_off_in) hipFree(d_off_in); if (d_par) hipFree(d_par); if (d_off_out) hipFree(d_off_out); if (d_att) hipFree(d_att); if (d_status) hipFree(d_status); };
#define JB_TRY(call) do { hipError_t _e = (call); if (_e != hipSuccess) { cleanup(); return fail(JB_E_HIP, std::string(#call) + ": " + hipGetErrorString(_e)); } } while (0)
    JB_TRY(hipMalloc(&d_tables, N * LM_TABLE * sizeof(float)));
    JB_TRY(hipMalloc(&d_status, sizeof(int)));
    JB_TRY(hipMemset(d_status, 0, sizeof(int)));
    if (offsets_in) { JB_TRY(hipMalloc(&d_off_in, sizeof(double) * N * AO_COUNT)); JB_TRY(hipMemcpy(d_off_in, offsets_in, sizeof(double) * N * AO_COUNT, hipMemcpyHostToDevice)); }
    if (params_out) JB_TRY(hipMalloc(&d_par, sizeof(double) * N * JB_NPARAM));
    if (offsets_out) JB_TRY(hipMalloc(&d_off_out, sizeof(double) * N * AO_COUNT));
    if (attempts_out) JB_TRY(hipMalloc(&d_att, sizeof(int) * N));
    RndArgs a;
    a.n = (int)N; a.flags = cfg->flags; a.max_attempts = cfg->max_attempts > 0 ? cfg->max_attempts : 64;
    a.seed = cfg->seed; a.env_offset = h->cfg.env_offset;
    for (int i = 0; i < 3; i++) { a.sd.legs[i] = cfg->sd_legs[i]; a.sd.mass_pos[i] = cfg->sd_mass_pos[i]; }
    a.sd.core1_density = cfg->sd_core1_density; a.sd.core2_density = cfg->sd_core2_density; a.sd.global_density = cfg->sd_global_density; a.sd.gear = cfg->sd_gear;
    a.min_mass_clearance = cfg->min_mass_clearance;
    a.spec = h->d_spec; a.offsets_in = d_off_in; a.tables = d_tables; a.params_out = d_par; a.offsets_out = d_off_out; a.attempts_out = d_att; a.status = d_status;
    hipLaunchKernelGGL(jb_randomise_kernel, dim3((unsigned)((N + 63) / 64)), dim3(64), 0, h->stream, a);
    JB_TRY(hipGetLastError());
    JB_TRY(hipStreamSynchronize(h->stream));
    JB_TRY(hipMemcpy(&status, d_status, sizeof(int), hipMemcpyDeviceToHost));
    if (status == 0) {
        if (params_out) JB_TRY(hipMemcpy(params_out, d_par, sizeof(double) * N * JB_NPARAM, hipMemcpyDeviceToHost));
        if (offsets_out) JB_TRY(hipMemcpy(offsets_out, d_off_out, sizeof(double) * N * AO_COUNT, hipMemcpyDeviceToHost));
        if (attempts_out) JB_TRY(hipMemcpy(attempts_out, d_att, sizeof(int) * N, hipMemcpyDeviceToHost));
    }
#undef JB_TRY
    if (status != 0) {      // the handle keeps the model it had
        cleanup();
        return fail(JB_E_MODEL, status == -30 ? "randomise: no draw cleared min_mass_clearance within max_attempts for some env" : "randomise: a generated model is not supported by the kernel (code " + std::to_string(status) + ")");
    }
    if (h->d_model) hipFree(h->d_model);          // (the stream is idle: synchronised above)
    h->d_model = d_tables; h->model_tables = N;
    d_tables = nullptr;
    cleanup();
    h->ka.lane_model = h->d_model;
    h->ka.per_env_model = 1;
    h->ka.pair = (h->cfg.flags & JB_FLAG_NO_PAIR) ? 0 : 1;
    return rc;
}
// host-only helpers (no GPU): the native compiler / validity check on ONE model - what the device kernel runs per env
int jb_model_compile_host(const double* offsets, int32_t flags, double* params_out) {
    if (!params_out) return fail(JB_E_INVALID, "params_out is NULL");
    JbNominalSpec S;
    double zero[AO_COUNT] = {0};
    apply_offsets(JB_NOMINAL_SPEC, offsets ? flags : 0, offsets ? offsets : zero, S);
    const int rc = compile_model(S, params_out);
    return rc ? fail(JB_E_MODEL, "model does not compile (code " + std::to_string(rc) + ")") : JB_OK;
}
int jb_model_mass_clearance_ok(const double* params, double margin) {
    if (!params) return fail(JB_E_INVALID, "params is NULL");
    return mass_sweep_clear(params, margin) ? 1 : 0;
}
int jb_model_draw_offsets_host(uint64_t seed, uint64_t env, uint32_t attempt, const jb_randomise_cfg* cfg, double* offsets_out) {
    if (!cfg || !offsets_out) return fail(JB_E_INVALID, "cfg/offsets_out is NULL");
    AugSigmas sd;
    for (int i = 0; i < 3; i++) { sd.legs[i] = cfg->sd_legs[i]; sd.mass_pos[i] = cfg->sd_mass_pos[i]; }
    sd.core1_density = cfg->sd_core1_density; sd.core2_density = cfg->sd_core2_density; sd.global_density = cfg->sd_global_density; sd.gear = cfg->sd_gear;
    draw_offsets(seed, env, attempt, cfg->flags, sd, offsets_out);
    return JB_OK;
}


// ---- rows between GPUs without Python: one RCCL communicator per handle (gather of [N_local, D+2] rows to rank 0;
// on 8 MI355X that is 7 concurrent single-hop xGMI sends, not a ring)
int jb_comm_unique_id(void* id_out) {
    if (!id_out) return fail(JB_E_INVALID, "id_out is NULL");
    int rc = load_rccl();
    if (rc) return rc;
    JB_NCCL(g_rccl.GetUniqueId(id_out));
    return JB_OK;
}
int jb_comm_init(jb_handle* h, int32_t n_ranks, int32_t rank, const void* id) {
    if (!h || !id) return fail(JB_E_INVALID, "handle/id is NULL");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) return fail(JB_E_INVALID, "need 0 <= rank < n_ranks");
    if (h->comm) return fail(JB_E_INVALID, "the handle already has a communicator (jb_comm_destroy first)");
    JB_ENTER(h);
    int rc = load_rccl();
    if (rc) return rc;
    RcclApi::IdBlob blob;
    std::memcpy(blob.b, id, sizeof blob.b);
    JB_NCCL(g_rccl.CommInitRank(&h->comm, n_ranks, blob, rank));
    h->comm_ranks = n_ranks; h->comm_rank = rank; h->comm_nmax = 0;
    return JB_OK;
}
int jb_comm_set_shards(jb_handle* h, const int32_t* shard_envs) {
    if (!h || !shard_envs) return fail(JB_E_INVALID, "handle/shard_envs is NULL");
    if (!h->comm) return fail(JB_E_INVALID, "no communicator (jb_comm_init)");
    int nmax = 0;
    for (int r = 0; r < h->comm_ranks; r++) {
        if (shard_envs[r] < 1) return fail(JB_E_INVALID, "shard_envs[" + std::to_string(r) + "] < 1");
        nmax = shard_envs[r] > nmax ? shard_envs[r] : nmax;
    }
    if (shard_envs[h->comm_rank] != h->cfg.n_envs)
        return fail(JB_E_INVALID, "shard_envs[" + std::to_string(h->comm_rank) + "] = " + std::to_string(shard_envs[h->comm_rank]) + " but this handle steps " + std::to_string(h->cfg.n_envs) +
                                  " envs: the ranks do not agree on the partition");
    h->comm_nmax = nmax;
    return JB_OK;
}
int jb_comm_destroy(jb_handle* h) {
    if (!h) return fail(JB_E_INVALID, "handle is NULL");
    // the exchanges run on streams of the caller's choosing (ShardedJitterbugEnv: a side stream): wait for the whole device, not just the handle's stream
    if (h->comm) { JB_ENTER(h); hipDeviceSynchronize(); JB_NCCL(g_rccl.CommDestroy(h->comm)); h->comm = nullptr; h->comm_nmax = 0; }
    return JB_OK;
}
// the same exchange for a block of `count` floats per rank (equal on every rank): what a fused K-step rollout returns, [K, N_local, D+2]
int jb_gather_block_device(jb_handle* h, const float* d_src, float* d_all, int64_t count, void* stream, int32_t use_stream) {
    if (!h || !d_src || count < 1) return fail(JB_E_INVALID, "handle/src is NULL or count < 1");
    if (!h->comm) return fail(JB_E_INVALID, "no communicator (jb_comm_init)");
    if (h->comm_rank == 0 && !d_all) return fail(JB_E_INVALID, "rank 0 needs the receive buffer [n_ranks, count]");
    JB_ENTER(h);
    RoctxRange range("jb_gather_block");
    hipStream_t st = use_stream ? (hipStream_t)stream : h->stream;
    JB_NCCL(g_rccl.GroupStart());
    int err = 0;              // a failed Send/Recv must not leave the RCCL group open: close it, then report the first error
    if (h->comm_rank == 0)
        for (int r = 0; r < h->comm_ranks && !err; r++) err = g_rccl.Recv(d_all + (size_t)r * (size_t)count, (size_t)count, 7 /*ncclFloat*/, r, h->comm, st);
    if (!err) err = g_rccl.Send(d_src, (size_t)count, 7 /*ncclFloat*/, 0, h->comm, st);
    const int end = g_rccl.GroupEnd();
    if (err || end) return fail(JB_E_HIP, std::string("RCCL block gather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(err ? err : end) : "RCCL error"));
    return JB_OK;
}
int jb_gather_rows_device(jb_handle* h, const float* d_rows, float* d_all, void* stream, int32_t use_stream) {
    if (!h || !d_rows) return fail(JB_E_INVALID, "handle/rows is NULL");
    if (!h->comm) return fail(JB_E_INVALID, "no communicator (jb_comm_init)");
    if (h->comm_rank == 0 && !d_all) return fail(JB_E_INVALID, "rank 0 needs the receive buffer [n_ranks, N_local, D+2]");
    JB_ENTER(h);
    RoctxRange range("jb_gather_rows");
    hipStream_t st = use_stream ? (hipStream_t)stream : h->stream;
    // every rank posts blocks of the longest shard (jb_comm_set_shards; without it the shards are equal by contract)
    const size_t count = (size_t)(h->comm_nmax > 0 ? h->comm_nmax : h->cfg.n_envs) * (size_t)(h->D + 2);
    JB_NCCL(g_rccl.GroupStart());
    int err = 0;              // a failed Send/Recv must not leave the RCCL group open: close it, then report the first error
    if (h->comm_rank == 0)
        for (int r = 0; r < h->comm_ranks && !err; r++) err = g_rccl.Recv(d_all + (size_t)r * count, count, 7 /*ncclFloat*/, r, h->comm, st);
    if (!err) err = g_rccl.Send(d_rows, count, 7 /*ncclFloat*/, 0, h->comm, st);
    const int end = g_rccl.GroupEnd();
    if (err || end) return fail(JB_E_HIP, std::string("RCCL row gather: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(err ? err : end) : "RCCL error"));
    return JB_OK;
}
int jb_scatter_actions_device(jb_handle* h, const float* d_all, float* d_local, int64_t count, void* stream, int32_t use_stream) {
    if (!h || !d_local || count < 1) return fail(JB_E_INVALID, "handle/local is NULL or count < 1");
    if (!h->comm) return fail(JB_E_INVALID, "no communicator (jb_comm_init)");
    if (h->comm_rank == 0 && !d_all) return fail(JB_E_INVALID, "rank 0 needs the send buffer [n_ranks, count]");
    if (count < (int64_t)h->cfg.n_envs || (h->comm_nmax > 0 && count != (int64_t)h->comm_nmax))
        return fail(JB_E_INVALID, "jb_scatter_actions_device: count must be the longest shard's envs (" + std::to_string(h->comm_nmax > 0 ? h->comm_nmax : h->cfg.n_envs) + "), the same on every rank");
    JB_ENTER(h);
    RoctxRange range("jb_scatter_actions");
    hipStream_t st = use_stream ? (hipStream_t)stream : h->stream;
    JB_NCCL(g_rccl.GroupStart());
    int err = 0;              // (as in the gathers: the group is closed before an error is reported)
    if (h->comm_rank == 0)
        for (int r = 0; r < h->comm_ranks && !err; r++) err = g_rccl.Send(d_all + (size_t)r * (size_t)count, (size_t)count, 7 /*ncclFloat*/, r, h->comm, st);
    if (!err) err = g_rccl.Recv(d_local, (size_t)count, 7 /*ncclFloat*/, 0, h->comm, st);
    const int end = g_rccl.GroupEnd();
    if (err || end) return fail(JB_E_HIP, std::string("RCCL action scatter: ") + (g_rccl.GetErrorString ? g_rccl.GetErrorString(err ? err : end) : "RCCL error"));
    return JB_OK;
}

}  // extern "C"
