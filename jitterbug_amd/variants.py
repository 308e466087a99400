"""Which step-kernel variant a batch should run - the one place the thresholds live (JitterbugVecEnv, ShardedJitterbugEnv and bench.py
all resolve `variant="auto"` here).

The reference's harness builds a vectorised env with no knobs (benchmarks/benchmark.py:146-171); so does `variant="auto"`:

  ordinary   one four-env wave per SIMD (445 registers, 25.6 KB of LDS): the fastest kernel while a GPU holds at most one wave per SIMD
  lean       two four-env waves per SIMD (<= 256 registers, <= 20.4 KB of LDS: eight waves per CU, JB_FLAG_LEAN): pays from 2048 waves per
             GPU on, i.e. 8192 envs with a shared model (measured on MI355X, round 4: 8192 envs 7.50 -> 8.75 M env-steps/s per step and
             8.8 -> 11.8 M as one fused 1000-step launch, 16 384: 8.6 -> 11.5 M, 65 536: 14.1 M).  One model per env (LEAN + PAIR kernel,
             split tables with a per-substep overlay): from 16 384 envs on (8.2 M per step, 9.2 M in fused 100-step launches); at 8192
             the one-wave kernel with its waves launched longest-first is ahead per step (5.55 against 5.10 M) although the LEAN kernel
             wins as ONE fused 1000-step launch (6.76 against 5.56 M) - per step is what "auto" optimises; DESIGN.md 4

The two kernels agree to fp32 rounding, not bit for bit, so every shard of one batch must run the same one: a sharded env resolves
"auto" from the GLOBAL batch and the world size (the per-GPU count every rank computes identically), never from its own shard length.
"""
LEAN_MIN_ENVS_PER_GPU = 8192                  # shared model
LEAN_MIN_ENVS_PER_GPU_PER_ENV_MODEL = 16384   # one model per env (LEAN + PAIR kernel, split constant tables)
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
