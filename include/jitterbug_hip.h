/* jitterbug_hip.h — C ABI of libjitterbug_hip.so, the MI355X-native Jitterbug stepper.
 *
 * Drop-in boundary for the reference's hot path.  The reference has no FFI of
 * its own (it is pure Python over dm_control/MuJoCo); the entry points below are
 * what a binding for that path would bind, one per reference interface:
 *
 *   jb_create / jb_destroy   suite.load('jitterbug', task, task_kwargs, environment_kwargs)
 *                            -> task factories, reference jitterbug_dmc/jitterbug.py:72-174
 *                            (Physics.from_xml_string + Jitterbug task + control.Environment
 *                            with time_limit=10, control_timestep=0.01: :84-90)
 *   jb_reset                 control.Environment.reset() -> Jitterbug.initialize_episode,
 *                            reference jitterbug.py:601-666, then get_observation :673-763
 *   jb_step                  control.Environment.step(action): 50 x Physics.step() on
 *                            jitterbug.xml, then Jitterbug.get_reward (jitterbug.py:891-925)
 *                            and get_observation (:673-763); call sites
 *                            benchmarks/evaluate_policy.py:29-33, gym_wrapper.py:59
 *   jb_get_state/jb_set_state  physics.named.data.qpos / qvel  (jitterbug.py:180-243) and the
 *                            target pose written at :621-648
 *   jb_set_model_params      per-environment model constants, the intent of
 *                            augmented_jitterbug.py:95-267 (domain randomisation)
 *   jb_obs_dim               observation widths 15/16/19/18/19, jitterbug.py:700-753
 *
 * Conventions
 *   - N environments advance in lockstep.  Arrays are C-contiguous, row major:
 *     action[N], obs[N, D], reward[N], done[N] (uint8), qpos[N,16], qvel[N,15],
 *     target[N,3] = (x, y, yaw).  qpos/qvel use MuJoCo's layout for this model
 *     (SURVEY.md §8a): qpos = [x y z | qw qx qy qz | 8 leg hinges | motor],
 *     qvel = [v_world(3) | omega_body(3) | 8 leg rates | motor rate].
 *   - The caller owns every buffer it passes; the handle owns device memory,
 *     its stream and the RNG counters.  "_device" entry points take device
 *     pointers and are asynchronous on the handle's stream; the others take
 *     host pointers and return with the results valid.
 *   - Every function returns 0 on success or a negative JB_E_* code; the
 *     message is available from jb_last_error() (thread local).  No C++
 *     exception crosses this boundary.  There is no CPU fallback: without a
 *     usable HIP device jb_create fails with JB_E_NODEVICE.
 *   - A handle is not thread safe; distinct handles (one per GPU / rank) are
 *     independent.
 */
#ifndef JITTERBUG_HIP_H
#define JITTERBUG_HIP_H

#include <stdint.h>

#include "jitterbug_model.h"

#ifdef __cplusplus
extern "C" {
#endif

#define JB_ABI_VERSION 5

#define JB_OK            0
#define JB_E_INVALID    -1   /* bad argument */
#define JB_E_NODEVICE   -2   /* no usable HIP device */
#define JB_E_HIP        -3   /* a HIP runtime call failed */
#define JB_E_MODEL      -4   /* parameter table asks for something the kernel does not implement */

typedef struct jb_handle jb_handle;

#define JB_FLAG_NO_RANK_ONE 1   /* diagnostic: every Newton pass is a full sweep + refactorisation (no rank-one passes) */
#define JB_FLAG_LEAN        2   /* the two-waves-per-SIMD kernel variant (<= 256 registers and 20 KB of LDS per four-env wave; state / system /
                                  factorisation parked in LDS).  It agrees with the ordinary kernel to fp32 rounding, NOT bit for bit (the
                                  compiler fuses multiply-adds differently in the two instantiations): give every shard of one batch the
                                  same flag.  Pays when a GPU holds more four-env waves than SIMDs (> 4096 envs; one model per env: from 8192), see
                                  jitterbug_amd.variants / DESIGN.md 4.  A combination it cannot run (a model that needs the pair contact
                                  unless that is one model per env at 4 envs per wave) is refused with JB_E_INVALID, never run silently as
                                  the ordinary kernel; jb_kernel_variant() reports what a handle launches */
#define JB_FLAG_PAIR        4   /* always run the kernel variant with the geom-geom contacts (eccentric-mass ellipsoid, and - round 5 - the motor-axis
                                  thread, against the upper-leg cylinders; reference jitterbug.xml:44-107: every jitterbug geom collides).  Without the flag the variant
                                  is chosen by the model: on for one-model-per-env batches and for a shared table whose mass comes within
                                  0.5 mm of a leg, off for the nominal model (whose mass clears the legs by 3 mm) */
#define JB_FLAG_NO_PAIR     8   /* never: floor contacts only (rounds 1-2 behaviour; diagnostic) */
#define JB_FLAG_NO_SPREAD   16  /* diagnostic: contact sweeps never in spread mode (a robot lying on a leg keeps its 8-9 contacts on that leg's four
                                   lanes, two or three rounds per sweep) */

#define JB_FLAG_NO_REORDER  32  /* diagnostic: never reorder the waves of a launch.  By default a batch with more waves than the device holds at once
                                   launches them longest-first, by the wave times the previous launch measured (results do not depend on it) */

#define JB_FLAG_PAIR_WITNESS 64 /* diagnostic, off by default: behind every step launch one pass of the PAIR WITNESS (jb_get_pair_witness below) - for users whose
                                   randomised models leave the reference's distribution */

typedef struct jb_config {
    int32_t  n_envs;        /* N >= 1 */
    int32_t  task_id;       /* JB_TASK_* */
    int32_t  device_id;     /* HIP device ordinal */
    int32_t  random_pose;   /* reference jitterbug.py:378 (default 1) */
    int32_t  contacts;      /* 1: floor contacts on; 0: MuJoCo disableflags=contact */
    int32_t  substeps;      /* physics substeps per control step (reference: 0.01/0.0002 = 50) */
    int32_t  step_limit;    /* control steps per episode (reference: 10/(0.0002*50) = 1000) */
    int32_t  auto_reset;    /* 1: VecEnv semantics - an env that reports done is reset inside the
                                  same call and its returned observation is the new episode's first */
    int32_t  max_newton;    /* checks of the active set per substep after which the contact problem goes to the line-searched Newton solve (0 -> default 12; jb_sim.hpp newton_phase) */
    int32_t  use_caller_stream; /* 1: launch on `stream` below even when it is NULL (the legacy default stream) */
    int32_t  envs_per_wave; /* environments per 64-lane wavefront (4 lanes each): 1, 2, 4 or 8 (larger requests run as 8), 0 = choose so that the batch
                               spreads over all SIMDs of the device (small batches use partially filled waves) */
    int32_t  flags;         /* JB_FLAG_* bits, 0 by default */
    uint64_t seed;          /* RNG key */
    uint64_t env_offset;    /* global index of env 0 (sharding: results do not depend on the split) */
    void*    stream;        /* hipStream_t to run on when use_caller_stream=1; otherwise the handle owns a stream */
} jb_config;

/* fills cfg with the reference defaults for `task_id` and `n_envs` */
int jb_default_config(jb_config* cfg, int32_t n_envs, int32_t task_id);

int jb_create(const jb_config* cfg, jb_handle** out);
int jb_destroy(jb_handle* h);

/* host-buffer entry points (synchronous) */
int jb_reset(jb_handle* h, const uint8_t* mask /*[N] nullable = all*/, float* obs_out /*[N,D] nullable*/);
int jb_step(jb_handle* h, const float* action /*[N]*/, float* obs_out /*[N,D]*/, float* reward_out /*[N]*/, uint8_t* done_out /*[N]*/);
/* The same step in two halves - the step_async / step_wait split of the VecEnv the reference trains against (stable-baselines
 * SubprocVecEnv, reference benchmarks/benchmark.py:146-171: the workers step while the caller goes on).  jb_step_async copies the actions
 * into pinned staging owned by the handle and queues H2D copy -> step kernel -> D2H copies of obs / reward / done into pinned buffers ->
 * an event on the handle's stream, and returns at once; jb_step_wait waits for that event and copies the results into the caller's buffers
 * (each nullable).  jb_step_views hands out the pinned result buffers themselves (valid until the next jb_step_async): a caller that
 * reads them in place needs no copy at all.  Exactly one jb_step_wait per jb_step_async; results are bit-identical to jb_step's. */
int jb_step_async(jb_handle* h, const float* action /*[N]*/);
int jb_step_wait(jb_handle* h, float* obs_out /*[N,D] nullable*/, float* reward_out /*[N] nullable*/, uint8_t* done_out /*[N] nullable*/);
int jb_step_views(jb_handle* h, const float** obs /*[N,D]*/, const float** reward /*[N]*/, const uint8_t** done /*[N]*/);
int jb_observe(jb_handle* h, float* obs_out /*[N,D]*/, float* reward_out /*[N] nullable*/);
int jb_get_state(jb_handle* h, double* qpos /*[N,16]*/, double* qvel /*[N,15]*/, double* target /*[N,3]*/);
int jb_set_state(jb_handle* h, const double* qpos, const double* qvel, const double* target);   /* any may be NULL = keep */
/* solver_cap_hits: +1 for every substep whose contact solve did NOT converge - neither the active-set iteration within max_newton checks
 * nor the line-searched Newton solve that takes over then (never expected: MuJoCo's own solver, reference jitterbug.xml:18 defaults) -,
 * +1000 for every control step that ended in a non-finite state (never expected; the env stays non-finite until its next reset) -
 * cumulative over the handle's life */
int jb_get_counters(jb_handle* h, int32_t* step_count /*[N]*/, uint32_t* episode /*[N]*/, float* solver_cap_hits /*[N]*/);
/* how often the plain active-set iteration did not settle within max_newton checks and the substep's contact problem was solved again with
 * the exact line search: wave-substeps (a wave = jb_envs_per_wave envs) since jb_create - a few in ten million */
int jb_solver_stats(jb_handle* h, uint64_t* resolved_wave_substeps);
/* The pair witness.  Every jitterbug geom has contype = conaffinity = 1 (reference jitterbug_dmc/jitterbug.xml:44-107): MuJoCo tests the 160
 * geom pairs whose bodies differ and are not parent and child.  The step kernels collide every geom with the floor and the eight pairs
 * that do occur on randomised models (mass ellipsoid and motor-axis thread against the upper-leg cylinders); the other 152 never touch on
 * the reference's model or under the reference's randomisation (measured: tests/test_gpu_clearance.py) - but a caller's sigmas may be
 * larger.  The witness says so instead of letting bodies pass through each other silently: per env, the exact (GJK, fp64) smallest
 * distance over the 152 unsimulated pairs in the current state, from the constant tables the step kernel reads; 0 = a pair interpenetrates.
 *   jb_pair_witness      one pass now: clearance_out [N] metres (nullable), pairs_out [N, 2] geom indices in the reference XML's order
 *                        (nullable), *n_touching = envs with clearance 0 (nullable).  Synchronous.
 *   jb_get_pair_witness  handles created with JB_FLAG_PAIR_WITNESS run a pass behind every step launch: overlap_passes [N] = passes that
 *                        found a pair interpenetrating, min_clearance [N] = smallest clearance seen, both since jb_create (nullable). */
int jb_pair_witness(jb_handle* h, float* clearance_out, int32_t* pairs_out, int32_t* n_touching);
int jb_get_pair_witness(jb_handle* h, uint32_t* overlap_passes, float* min_clearance);
/* n_tables is 1 (shared) or N (one table per env); each table is JB_NPARAM doubles (jitterbug_model.h) */
int jb_set_model_params(jb_handle* h, const double* params, int32_t n_tables);

/* Domain randomisation ON THE DEVICE, one model per environment (reference augment_Jitterbug,
 * jitterbug_dmc/augmented_jitterbug.py:95-267): per env, Philox draws keyed (cfg.seed, GLOBAL env index, attempt) -> offsets of the
 * leg ends / motor axis / densities / gear with the reference's distributions -> perturbed geometry, hinge axes recomputed
 * (:182-186, :212-215) -> geom masses, body COM / inertia, body_invweight0 (what MuJoCo's compiler derives) -> the kernel's lane
 * constant tables, all in HBM.  The handle then simulates N models (like jb_set_model_params with N tables).
 *   offsets_in  (host, [N, JB_NOFFSET], nullable): compile THESE offsets instead of drawing (tests; replaying a saved set)
 *   params_out  (host, [N, JB_NPARAM],  nullable): the compiled parameter tables (5 KB per env: ask only when needed)
 *   offsets_out (host, [N, JB_NOFFSET], nullable), attempts_out (host, [N], nullable)
 * cfg.min_mass_clearance > 0 re-draws (attempt 1, 2, ...) any model whose eccentric mass cannot turn at the rest pose without
 * coming within that distance of a leg: the reference's sigmas produce such robots in 3.6 % of the draws.  MuJoCo simulates a
 * mass-leg contact there and so does this simulator (the PAIR kernel variant, chosen automatically for one-model-per-env
 * batches), so the default 0 keeps every draw; a positive margin is for callers who want buildable robots only. */
#define JB_NOFFSET 31   /* [global density, coreBody1 density, coreBody2 density | 4 legs x (upper far end xyz, foot end xyz) | motor xyz | gear] */
#define JB_RND_LEGS 1
#define JB_RND_MASS 2
#define JB_RND_CORE1_DENSITY 4
#define JB_RND_CORE2_DENSITY 8
#define JB_RND_GLOBAL_DENSITY 16
#define JB_RND_GEAR 32
typedef struct jb_randomise_cfg {
    int32_t  flags;               /* JB_RND_* (reference keyword arguments modify_legs, modify_mass, ...) */
    int32_t  max_attempts;        /* re-draws allowed per env (0 -> 64) */
    uint64_t seed;
    double   sd_legs[3];          /* reference :96  (0.003, 0.003, 0.002) m */
    double   sd_mass_pos[3];      /* reference :98  (0.0015, 0.002, 0.001) m; y and z are clipped below at -1 mm like the reference */
    double   sd_core1_density, sd_core2_density, sd_global_density, sd_gear;      /* reference :100-107 */
    double   min_mass_clearance;  /* metres; <= 0: accept every draw */
} jb_randomise_cfg;
int jb_default_randomise_config(jb_randomise_cfg* cfg);
int jb_randomise_models(jb_handle* h, const jb_randomise_cfg* cfg, const double* offsets_in, double* params_out, double* offsets_out, int32_t* attempts_out);
/* host-only helpers (no GPU needed): the same native code on ONE model - compile offsets (NULL = nominal) to a parameter table,
 * the mass-clearance check on a parameter table (returns 1 / 0), and the draw of (seed, env, attempt) */
int jb_model_compile_host(const double* offsets /*[JB_NOFFSET] nullable*/, int32_t flags, double* params_out /*[JB_NPARAM]*/);
int jb_model_mass_clearance_ok(const double* params /*[JB_NPARAM]*/, double margin);
int jb_model_draw_offsets_host(uint64_t seed, uint64_t env, uint32_t attempt, const jb_randomise_cfg* cfg, double* offsets_out /*[JB_NOFFSET]*/);

/* device-buffer entry points (asynchronous on the handle's stream) */
int jb_reset_device(jb_handle* h, const uint8_t* d_mask /*nullable*/, float* d_obs_out /*nullable*/);
int jb_step_device(jb_handle* h, const float* d_action, float* d_obs_out, float* d_reward_out, uint8_t* d_done_out);
/* same step, one packed float row per env: rows[N, D+2] = [obs(D) | reward | done (0.0/1.0)] - the unit the north-star's
 * per-step gather moves between GPUs, written by the step kernel itself */
int jb_step_rows_device(jb_handle* h, const float* d_action, float* d_rows_out /*[N, D+2]*/);
int jb_observe_device(jb_handle* h, float* d_obs_out, float* d_reward_out /*nullable*/);
/* heuristic bang-bang policies of the reference (heuristic_policies.py:6-136) for the handle's task, evaluated on
 * observation rows [N,D] -> actions [N]; the device form lets a rollout chain observe -> act -> step without leaving HBM */
int jb_policy_device(jb_handle* h, const float* d_obs, float* d_action);
int jb_policy(jb_handle* h, const float* obs, float* action);
/* the reference policies' keyword arguments (heuristic_policies.py:28 kick_angle = 45 deg, speed = 0.3; :64,:81,:98
 * angle_threshold = 20 deg); the defaults are set at jb_create */
int jb_set_policy_params(jb_handle* h, float kick_angle, float speed, float angle_threshold);
/* the four reward terms of every env, terms[N,4] = [position, heading, velocity, upright]: Jitterbug.position_reward /
 * heading_reward / velocity_reward / upright_reward, reference jitterbug.py:840-889 (whatever the handle's task) */
int jb_reward_terms_device(jb_handle* h, float* d_terms_out /*[N,4]*/);
int jb_reward_terms(jb_handle* h, float* terms_out /*[N,4]*/);
/* n_steps of (policy -> step) chained on the stream, the loop of benchmarks/evaluate_policy.py:29-33 for the whole batch;
 * d_obs_inout [N,D] holds the current observations on entry and the last ones on return; d_rewards [n_steps,N] nullable */
int jb_rollout_policy_device(jb_handle* h, int32_t n_steps, float* d_obs_inout, float* d_rewards, uint8_t* d_done_last);
/* host-buffer form: starts from the handle's current state; rewards_out [n_steps,N] and obs_out [N,D] are nullable */
int jb_rollout_policy(jb_handle* h, int32_t n_steps, float* rewards_out, float* obs_out);
/* K control steps in ONE kernel launch (SURVEY.md 3.3 "one kernel launch per control step or per K control steps", 8(f)1 "policies fused
 * ahead of the step kernel"; the loops it serves: benchmarks/evaluate_policy.py:29-33, the async step_async / step_wait split of
 * benchmarks/benchmark.py:146-171).  Every wave keeps its environments' state in registers / LDS across the K steps and never waits for
 * another wave, so the launch lasts as long as its slowest wave's SUM over the steps, not the sum of every step's slowest wave.
 *   d_actions   [K, N] action tape (step k applies row k), or NULL: the handle's heuristic policy (jb_set_policy_params) evaluated in the
 *               kernel on the observation the lanes just produced (step 0: on the current state's observation)
 *   d_rows_out  [K, N, D+2] nullable: the packed rows [obs | reward | done] of every step (what jb_step_rows_device writes per step)
 *   d_rewards   [K, N] nullable; d_obs_last [N, D] nullable; d_done_last [N] nullable: per-step rewards, the last step's observations
 *               and done flags.  Rows and these three exclude each other.
 * jb_step_device / jb_step_rows_device ARE this kernel with K = 1: K single-step calls and one K-step call give bit-identical
 * states, rows and rewards (tests/test_gpu_rollout.py), in-kernel auto-reset included. */
int jb_step_many_device(jb_handle* h, int32_t n_steps, const float* d_actions, float* d_rows_out, float* d_rewards, float* d_obs_last, uint8_t* d_done_last);
/* host-buffer form (synchronous): actions [K, N] or NULL = in-kernel policy, rows_out [K, N, D+2] nullable; the device staging belongs to the
 * handle and only grows (no allocation from the second call of a size on) */
int jb_step_many(jb_handle* h, int32_t n_steps, const float* actions, float* rows_out);
/* frees the device staging jb_step_many / jb_rollout_policy have grown (K x N x (D+2) floats: 4-5 GB after one rollout(1000) of 65 536 envs);
 * the next call of either allocates what it needs again */
int jb_release_staging(jb_handle* h);
/* seconds each wave of the LAST step launch was alive (one wave = jb_envs_per_wave envs), out[0 .. min(n_waves, max_waves)); returns the
 * number of waves.  Mean against maximum is the load imbalance of the launch (roofline). */
int jb_wave_clocks(jb_handle* h, double* out, int32_t max_waves);
/* Rows between GPUs without Python: one RCCL communicator per handle, RCCL bound at run time (dlopen: no link-time
 * dependency; a process that already holds an RCCL, e.g. PyTorch-ROCm's, keeps that one).  Rank 0 makes the id, the host distributes
 * its JB_COMM_ID_BYTES to every rank by its own means, every rank calls jb_comm_init (collective).  jb_gather_rows_device sends this
 * rank's packed rows [N_local, D+2] (what jb_step_rows_device wrote; N_local equal on every rank, or jb_comm_set_shards) to rank 0, which receives
 * [n_ranks, N_local, D+2] - grouped ncclSend / ncclRecv: on 8 MI355X seven concurrent single-hop xGMI transfers, no ring.  It is
 * asynchronous on `stream` (use_stream = 1) or on the handle's stream; issue it from a side stream one step late, like
 * jitterbug_amd.distributed.ShardedJitterbugEnv does, so that it overlaps the next step kernel instead of delaying it. */
#define JB_COMM_ID_BYTES 128
int jb_comm_unique_id(void* id_out /*[JB_COMM_ID_BYTES]*/);
int jb_comm_init(jb_handle* h, int32_t n_ranks, int32_t rank, const void* id);
int jb_comm_destroy(jb_handle* h);
/* Uneven shards (ABI 5): the number of envs EVERY rank holds, the same list on every rank (the host knows the partition: contiguous ranges
 * of a global batch).  The exchanges below then move blocks of the LONGEST shard - rows buffers are [n_max, D+2], rank 0 receives
 * [n_ranks, n_max, D+2] and reads shard r's first shard_envs[r] rows; action blocks are [n_max] - so that every rank posts transfers of the
 * same size whatever its own shard.  JB_E_INVALID when shard_envs[rank] is not this handle's n_envs: a partition the ranks do not agree
 * on is refused HERE, on the host, instead of hanging RCCL in the first exchange.  Without this call every rank must hold the same number
 * of envs (ABI 4's contract). */
int jb_comm_set_shards(jb_handle* h, const int32_t* shard_envs /*[n_ranks]*/);
int jb_gather_rows_device(jb_handle* h, const float* d_rows /*[n_max, D+2]*/, float* d_all /*rank 0: [n_ranks, n_max, D+2]; others: NULL*/, void* stream, int32_t use_stream);
/* The action half of the per-step round trip (SURVEY.md 8e "actions flow the other way as a scatter of [N_local] fp32"; the reference sends
 * every worker its actions every step, benchmarks/benchmark.py:161-171 SubprocVecEnv.step_async): rank 0 holds the actions of every shard,
 * d_all [n_ranks, count] (shard r's block at r * count), and every rank receives its block into d_local [count] - grouped ncclSend x n_ranks
 * on rank 0, one ncclRecv everywhere: seven concurrent single-hop xGMI transfers on 8 MI355X.  count must be the same on every rank (n_max
 * with uneven shards); asynchronous on `stream` (use_stream = 1) or on the handle's stream. */
int jb_scatter_actions_device(jb_handle* h, const float* d_all /*rank 0: [n_ranks, count]; others: NULL*/, float* d_local /*[count]*/, int64_t count, void* stream, int32_t use_stream);
/* the same exchange for `count` floats per rank (equal on every rank): the [K, N_local, D+2] block a fused K-step rollout returns (jb_step_many_device) */
int jb_gather_block_device(jb_handle* h, const float* d_src, float* d_all /*rank 0: [n_ranks, count]; others: NULL*/, int64_t count, void* stream, int32_t use_stream);

/* Observation-encoder hook (reference jitterbug.py:760-761 -> encode_obs :927-993): a tiny dense network applied to every
 * observation row on the GPU.  n_layers <= JB_ENC_MAX_LAYERS dense layers; dims[0] must be the task's observation width and
 * every width <= JB_ENC_MAX_WIDTH; acts[l] in JB_ACT_*; weights are the layers' [in][out] matrices concatenated, biases
 * likewise.  vae = 1: the last layer's outputs are [mean(L) | std(L)] and the code is mean + std * eps with eps ~ N(0,1)
 * (reference benchmarks/VAE.py:134-146; the reference draws eps from torch's global RNG, here it is a Philox stream keyed
 * (seed, global env, call number)).  The autoencoder of benchmarks/autoencoder.py:71-106 is one tanh layer.  The reference's
 * trained weight files are not in its repository; the caller supplies weights.  n_layers = 0 removes the encoder. */
#define JB_ENC_MAX_LAYERS 4
#define JB_ENC_MAX_WIDTH  32
#define JB_ACT_LINEAR 0
#define JB_ACT_TANH   1
#define JB_ACT_RELU   2
int jb_set_obs_encoder(jb_handle* h, int32_t n_layers, const int32_t* dims /*[n_layers+1]*/, const int32_t* acts /*[n_layers]*/,
                       const float* weights, const float* biases, int32_t vae);
int jb_encoded_dim(jb_handle* h);              /* width of an encoded row, 0 if no encoder is set */
int jb_encode_device(jb_handle* h, const float* d_obs /*[N,D]*/, float* d_code_out /*[N, jb_encoded_dim]*/);
int jb_encode(jb_handle* h, const float* obs, float* code_out);
/* diagnostic: fill the device's LDS with NaN patterns on the handle's stream, so that a kernel reading scratch it did not write
 * in the same launch fails deterministically (tests/test_gpu_parity.py::test_no_uninitialised_scratch_is_read) */
int jb_debug_poison_lds(jb_handle* h);
int jb_synchronize(jb_handle* h);
void* jb_stream(jb_handle* h);                 /* the hipStream_t the handle launches on */

/* queries */
int jb_obs_dim(int32_t task_id);               /* 15/16/19/18/19, or JB_E_INVALID */
int jb_num_envs(jb_handle* h);
/* the step kernel a handle launches (it depends on the flags, the model and envs_per_wave; see JB_FLAG_LEAN / JB_FLAG_PAIR) */
#define JB_VARIANT_ORDINARY  0   /* one wave per SIMD, floor contacts */
#define JB_VARIANT_PAIR      1   /* one wave per SIMD, floor contacts + the mass / upper-leg geom-geom contact */
#define JB_VARIANT_LEAN      2   /* two waves per SIMD, floor contacts */
#define JB_VARIANT_LEAN_PAIR 3   /* two waves per SIMD, one model per env (split tables), floor + pair contact */
int jb_kernel_variant(jb_handle* h);
int jb_envs_per_wave(jb_handle* h);
int jb_device_count(void);
int jb_abi_version(void);
const char* jb_source_sha256(void);            /* sha256 (hex) of the sources this library was built from (jitterbug_amd/build.py) */
const double* jb_default_model_params(void);   /* the compiled nominal model, JB_NPARAM doubles */
const char* jb_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
