"""Which step-kernel variant a batch should run - the one place the thresholds live (JitterbugVecEnv, ShardedJitterbugEnv and bench.py
all resolve `variant="auto"` here).

The reference's harness builds a vectorised env with no knobs (benchmarks/benchmark.py:146-171); so does `variant="auto"`:

  ordinary   one four-env wave per SIMD (256 VGPRs + 256 AGPRs, 26.6 KB of LDS; profiles/r06_kernel_resources.txt): the fastest kernel while a GPU holds at
             most one wave per SIMD
  lean       two four-env waves per SIMD (256 VGPRs, no AGPRs - most of what it spills is used in its cold second contact solve, profiles/r06_asm_spills.txt -,
             <= 20.4 KB of LDS: eight waves per CU, JB_FLAG_LEAN): pays as soon as a GPU has
             more waves than SIMDs, i.e. from 4097 envs with a shared model - the one-wave kernel would need a second round for the
             rest.  While the batch still fits the device at once (up to two waves per SIMD) the library chooses WHO shares a SIMD with
             whom: the hardware puts workgroups b and b + 1024 on one SIMD, and the launch order is folded so that the longest wave of
             the previous launch runs alone or with the shortest (jb_wave_order_kernel).  Measured on MI355X, round 4, per step:
             4608 envs 5.38 -> 6.47 M env-steps/s, 6144: 6.88 -> 7.73 M, 8192: 7.82 -> 9.68 M, 65 536: 14.1 M.  One model per env (LEAN +
             PAIR kernel, split tables with a per-substep overlay): from 8192 envs on (8192: 5.95 -> 6.28 M per step with the folded
             pairing, 12 288: 6.74 -> 7.86 M; at 6144 the one-wave kernel is still ahead, 5.36 against 4.85 M); DESIGN.md 4

The two kernels agree to fp32 rounding, not bit for bit, so every shard of one batch must run the same one: a sharded env resolves
"auto" from the GLOBAL batch and the world size (the per-GPU count every rank computes identically), never from its own shard length.
"""
LEAN_MIN_ENVS_PER_GPU = 4097                  # shared model: more envs than one four-env wave per SIMD holds (1024 SIMDs)
LEAN_MIN_ENVS_PER_GPU_PER_ENV_MODEL = 8192    # one model per env (LEAN + PAIR kernel, split constant tables)
FLAG_LEAN = 2                                 # include/jitterbug_hip.h JB_FLAG_LEAN
VARIANTS = ("auto", "ordinary", "lean")


def envs_per_gpu(n_global, world=1):
    """The per-GPU batch every rank of a sharded run agrees on (shards differ by at most one env: the ceiling)."""
    return -(-int(n_global) // max(1, int(world)))


def resolve(variant, n_envs_per_gpu, per_env_model=False):
    """'auto' | 'ordinary' | 'lean'  ->  'ordinary' | 'lean' for a GPU that steps n_envs_per_gpu envs."""
    if variant not in VARIANTS:
        raise ValueError("variant must be one of %s, not %r" % (VARIANTS, variant))
    if variant != "auto":
        return variant
    threshold = LEAN_MIN_ENVS_PER_GPU_PER_ENV_MODEL if per_env_model else LEAN_MIN_ENVS_PER_GPU
    return "lean" if int(n_envs_per_gpu) >= threshold else "ordinary"


def flags_for(variant, n_envs_per_gpu, per_env_model=False, flags=0):
    """jb_config.flags with the resolved variant's bit set / cleared (other bits pass through)."""
    lean = resolve(variant, n_envs_per_gpu, per_env_model) == "lean"
    return (int(flags) | FLAG_LEAN) if lean else (int(flags) & ~FLAG_LEAN)
