"""ctypes binding of libjitterbug_hip.so (C ABI: include/jitterbug_hip.h).

There is no CPU fallback: if the library is missing or no HIP device is usable the
import of the library / creation of a handle raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("JITTERBUG_HIP_LIB") or os.path.join(HERE, "libjitterbug_hip.so")   # env override: A/B builds when profiling

JB_OK = 0
ERR_NAMES = {-1: "JB_E_INVALID", -2: "JB_E_NODEVICE", -3: "JB_E_HIP", -4: "JB_E_MODEL"}

EXPORTS = ["jb_default_config", "jb_create", "jb_destroy", "jb_reset", "jb_step", "jb_observe", "jb_get_state", "jb_set_state",
           "jb_get_counters", "jb_solver_stats", "jb_set_model_params", "jb_policy", "jb_policy_device", "jb_set_policy_params", "jb_reward_terms", "jb_reward_terms_device", "jb_default_randomise_config", "jb_randomise_models", "jb_model_compile_host", "jb_model_mass_clearance_ok", "jb_model_draw_offsets_host", "jb_comm_unique_id", "jb_comm_init", "jb_comm_destroy", "jb_comm_set_shards", "jb_gather_rows_device", "jb_gather_block_device", "jb_scatter_actions_device", "jb_step_async", "jb_step_wait", "jb_step_views", "jb_release_staging", "jb_pair_witness", "jb_get_pair_witness", "jb_rollout_policy_device", "jb_rollout_policy", "jb_step_many_device", "jb_step_many", "jb_wave_clocks", "jb_kernel_variant", "jb_envs_per_wave", "jb_reset_device", "jb_step_device", "jb_step_rows_device", "jb_observe_device", "jb_debug_poison_lds", "jb_set_obs_encoder", "jb_encoded_dim", "jb_encode_device", "jb_encode", "jb_synchronize",
           "jb_stream", "jb_obs_dim", "jb_num_envs", "jb_device_count", "jb_abi_version", "jb_source_sha256", "jb_default_model_params", "jb_last_error"]


class JitterbugHipError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("n_envs", C.c_int32), ("task_id", C.c_int32), ("device_id", C.c_int32), ("random_pose", C.c_int32),
                ("contacts", C.c_int32), ("substeps", C.c_int32), ("step_limit", C.c_int32), ("auto_reset", C.c_int32),
                ("max_newton", C.c_int32), ("use_caller_stream", C.c_int32), ("envs_per_wave", C.c_int32), ("flags", C.c_int32), ("seed", C.c_uint64), ("env_offset", C.c_uint64),
                ("stream", C.c_void_p)]


class RandomiseConfig(C.Structure):
    _fields_ = [("flags", C.c_int32), ("max_attempts", C.c_int32), ("seed", C.c_uint64), ("sd_legs", C.c_double * 3), ("sd_mass_pos", C.c_double * 3),
                ("sd_core1_density", C.c_double), ("sd_core2_density", C.c_double), ("sd_global_density", C.c_double), ("sd_gear", C.c_double),
                ("min_mass_clearance", C.c_double)]


VARIANT_NAMES = {0: "ordinary", 1: "pair", 2: "lean", 3: "lean_pair"}      # jb_kernel_variant (JB_VARIANT_*)
FLAG_NO_RANK_ONE, FLAG_LEAN, FLAG_PAIR, FLAG_NO_PAIR, FLAG_NO_SPREAD, FLAG_NO_REORDER, FLAG_PAIR_WITNESS = 1, 2, 4, 8, 16, 32, 64
NOFFSET = 31
RND_LEGS, RND_MASS, RND_CORE1_DENSITY, RND_CORE2_DENSITY, RND_GLOBAL_DENSITY, RND_GEAR = 1, 2, 4, 8, 16, 32

_lib = None


def _preload_torch_hip_runtime():
    """If PyTorch-ROCm is installed it ships its own libamdhip64; a process must end up with ONE HIP runtime whatever the
    import order, so bind to torch's copy (the configuration bench.py and torch.distributed run in) before loading ours.
    Does not import torch."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except Exception:
        pass


def load():
    """Load the shared library (once). Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise JitterbugHipError(
            "libjitterbug_hip.so is not built (%s). Build it with `python -m jitterbug_amd.build` "
            "(needs hipcc); there is no CPU fallback." % LIB_PATH)
    if os.environ.get("JITTERBUG_HIP_LIB") is None:
        # the in-tree library must be the build of the sources next to it (a stale .so beside newer sources is refused, whatever the
        # file times say); JITTERBUG_HIP_LIB - an A/B build named by hand - is taken as it is
        from . import build as _build
        want, have = _build.source_sha256(), _build.embedded_sha256(LIB_PATH)
        if want is not None and have != want:
            raise JitterbugHipError("libjitterbug_hip.so was built from other sources (embedded sha256 %s..., sources now %s...): rebuild with "
                                    "`python -m jitterbug_amd.build`, or name a library explicitly with JITTERBUG_HIP_LIB" % (str(have)[:12], want[:12]))
    _preload_torch_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, fp, dp, u8p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    L.jb_default_config.argtypes = [C.POINTER(Config), C.c_int32, C.c_int32]
    L.jb_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.jb_destroy.argtypes = [vp]
    L.jb_reset.argtypes = [vp, u8p, fp]
    L.jb_step.argtypes = [vp, fp, fp, fp, u8p]
    L.jb_observe.argtypes = [vp, fp, fp]
    L.jb_get_state.argtypes = [vp, dp, dp, dp]
    L.jb_set_state.argtypes = [vp, dp, dp, dp]
    L.jb_get_counters.argtypes = [vp, vp, vp, vp]
    L.jb_set_model_params.argtypes = [vp, dp, C.c_int32]
    L.jb_reset_device.argtypes = [vp, u8p, fp]
    L.jb_step_device.argtypes = [vp, fp, fp, fp, u8p]
    L.jb_step_rows_device.argtypes = [vp, fp, fp]
    L.jb_observe_device.argtypes = [vp, fp, fp]
    L.jb_set_obs_encoder.argtypes = [vp, C.c_int32, vp, vp, fp, fp, C.c_int32]
    L.jb_encoded_dim.argtypes = [vp]
    L.jb_debug_poison_lds.argtypes = [vp]
    L.jb_encode_device.argtypes = [vp, fp, fp]
    L.jb_encode.argtypes = [vp, fp, fp]
    L.jb_policy.argtypes = [vp, fp, fp]
    if os.environ.get("JITTERBUG_HIP_LIB") is None or hasattr(L, "jb_set_policy_params"):     # (an A/B build of an older ABI may lack these)
        L.jb_set_policy_params.argtypes = [vp, C.c_float, C.c_float, C.c_float]
        L.jb_reward_terms.argtypes = [vp, fp]
        L.jb_reward_terms_device.argtypes = [vp, fp]
    L.jb_policy_device.argtypes = [vp, fp, fp]
    L.jb_rollout_policy_device.argtypes = [vp, C.c_int32, fp, fp, u8p]
    L.jb_rollout_policy.argtypes = [vp, C.c_int32, fp, fp]
    if os.environ.get("JITTERBUG_HIP_LIB") is None or hasattr(L, "jb_step_many_device"):
        L.jb_step_many_device.argtypes = [vp, C.c_int32, fp, fp, fp, fp, u8p]
        L.jb_step_many.argtypes = [vp, C.c_int32, fp, fp]
        L.jb_wave_clocks.argtypes = [vp, dp, C.c_int32]
        L.jb_kernel_variant.argtypes = [vp]
        L.jb_envs_per_wave.argtypes = [vp]
    if os.environ.get("JITTERBUG_HIP_LIB") is None or hasattr(L, "jb_randomise_models"):
        L.jb_default_randomise_config.argtypes = [C.POINTER(RandomiseConfig)]
        L.jb_randomise_models.argtypes = [vp, C.POINTER(RandomiseConfig), dp, dp, dp, vp]
        L.jb_model_compile_host.argtypes = [dp, C.c_int32, dp]
        L.jb_model_mass_clearance_ok.argtypes = [dp, C.c_double]
        L.jb_model_draw_offsets_host.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(RandomiseConfig), dp]
    if os.environ.get("JITTERBUG_HIP_LIB") is None or hasattr(L, "jb_comm_init"):
        L.jb_comm_unique_id.argtypes = [vp]
        L.jb_comm_init.argtypes = [vp, C.c_int32, C.c_int32, vp]
        L.jb_comm_destroy.argtypes = [vp]
        L.jb_gather_rows_device.argtypes = [vp, fp, fp, vp, C.c_int32]
        if hasattr(L, "jb_gather_block_device"):
            L.jb_gather_block_device.argtypes = [vp, fp, fp, C.c_int64, vp, C.c_int32]
    if os.environ.get("JITTERBUG_HIP_LIB") is None or hasattr(L, "jb_step_async"):      # ABI 5
        L.jb_comm_set_shards.argtypes = [vp, vp]
        L.jb_scatter_actions_device.argtypes = [vp, fp, fp, C.c_int64, vp, C.c_int32]
        L.jb_step_async.argtypes = [vp, fp]
        L.jb_step_wait.argtypes = [vp, fp, fp, u8p]
        L.jb_step_views.argtypes = [vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)]
        L.jb_release_staging.argtypes = [vp]
        L.jb_pair_witness.argtypes = [vp, fp, vp, vp]
        L.jb_get_pair_witness.argtypes = [vp, vp, fp]
    if os.environ.get("JITTERBUG_HIP_LIB") is None or hasattr(L, "jb_solver_stats"):
        L.jb_solver_stats.argtypes = [vp, vp]
    L.jb_synchronize.argtypes = [vp]
    L.jb_stream.argtypes = [vp]
    L.jb_stream.restype = vp
    L.jb_obs_dim.argtypes = [C.c_int32]
    L.jb_num_envs.argtypes = [vp]
    L.jb_default_model_params.restype = C.POINTER(C.c_double)
    if hasattr(L, "jb_source_sha256"):
        L.jb_source_sha256.restype = C.c_char_p
    L.jb_last_error.restype = C.c_char_p
    _lib = L
    return L


def check(rc):
    if rc != JB_OK:
        msg = load().jb_last_error().decode("utf-8", "replace")
        raise JitterbugHipError("%s: %s" % (ERR_NAMES.get(rc, str(rc)), msg))


def ptr(a):
    return None if a is None else a.ctypes.data


def default_model_params():
    from . import model
    p = load().jb_default_model_params()
    return np.ctypeslib.as_array(p, shape=(model.NPARAM,)).copy()
