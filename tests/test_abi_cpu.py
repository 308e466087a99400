"""No-GPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/jitterbug_hip.h
declares, answers the device-free queries, and refuses to create a handle without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from jitterbug_amd import _lib, model

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from jitterbug_amd import build
    build.build()
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    hdr = open(os.path.join(ROOT, "include", "jitterbug_hip.h")).read()
    declared = set(re.findall(r"\b(jb_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_device_free_queries(lib):
    assert lib.jb_abi_version() == 5
    assert [lib.jb_obs_dim(i) for i in range(5)] == [15, 16, 19, 18, 19]
    assert lib.jb_obs_dim(5) == -1 and lib.jb_obs_dim(-1) == -1
    np.testing.assert_array_equal(_lib.default_model_params(), model.default_params())
    cfg = _lib.Config()
    assert lib.jb_default_config(C.byref(cfg), 4096, 0) == 0
    assert (cfg.n_envs, cfg.substeps, cfg.step_limit, cfg.contacts, cfg.random_pose, cfg.auto_reset) == (4096, 50, 1000, 1, 1, 1)


def test_config_struct_matches_header():
    hdr = open(os.path.join(ROOT, "include", "jitterbug_hip.h")).read()
    body = hdr[hdr.index("typedef struct jb_config {"):hdr.index("} jb_config;")]
    fields = re.findall(r"^\s*(?:int32_t|uint64_t|void\*)\s+(\w+);", body, flags=re.M)
    assert fields == [f[0] for f in _lib.Config._fields_]


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from jitterbug_amd.vec_env import JitterbugVecEnv
    with pytest.raises(_lib.JitterbugHipError, match="JB_E_NODEVICE"):
        JitterbugVecEnv(4)
    # argument validation happens before the device probe
    cfg = _lib.Config()
    lib.jb_default_config(C.byref(cfg), 0, 0)
    h = C.c_void_p()
    assert lib.jb_create(C.byref(cfg), C.byref(h)) == -1 and b"n_envs" in lib.jb_last_error()


def test_task_factories_validate_like_the_reference():
    from jitterbug_amd import jitterbug, suite
    with pytest.raises(AssertionError, match="Invalid task"):
        jitterbug.Jitterbug(task="fly")
    with pytest.raises(ValueError):
        suite.load("jitterbug", "fly")
    with pytest.raises(ValueError):
        suite.load("hopper", "stand")
    from jitterbug_amd.vec_env import compute_n_substeps
    assert compute_n_substeps(0.01) == 50
    with pytest.raises(ValueError):
        compute_n_substeps(0.00031)


def test_host_only_entry_points_validate_arguments(lib):
    """the device-free helpers of the native model compiler: argument errors come back as JB_E_INVALID with a message, never a crash"""
    assert lib.jb_default_randomise_config(None) == -1 and b"NULL" in lib.jb_last_error()
    assert lib.jb_model_compile_host(None, 0, None) == -1
    assert lib.jb_model_mass_clearance_ok(None, 0.001) == -1
    assert lib.jb_model_draw_offsets_host(0, 0, 0, None, None) == -1
    cfg = _lib.RandomiseConfig()
    assert lib.jb_default_randomise_config(cfg) == 0 and cfg.max_attempts == 64 and cfg.min_mass_clearance == 0.0
    P = np.zeros(model.NPARAM)
    assert lib.jb_model_compile_host(None, 0, _lib.ptr(P)) == 0 and lib.jb_model_mass_clearance_ok(_lib.ptr(P), 0.001) == 1
    # handle-taking entry points refuse a NULL handle before touching any device
    for call in (lambda: lib.jb_randomise_models(None, cfg, None, None, None, None), lambda: lib.jb_set_policy_params(None, 0.5, 0.3, 0.3),
                 lambda: lib.jb_reward_terms(None, None), lambda: lib.jb_comm_init(None, 1, 0, None), lambda: lib.jb_comm_destroy(None),
                 lambda: lib.jb_gather_rows_device(None, None, None, None, 0)):
        assert call() == -1
    assert lib.jb_comm_unique_id(None) == -1


def test_entry_points_hand_the_callers_device_back():
    """an entry point makes the HANDLE's device current and must restore the caller's on every return path (torch reads
    the current device through hipGetDevice).  The guard the library instantiates with hipGetDevice / hipSetDevice
    (csrc/jb_device_guard.hpp) is run here against a recording stub; with two real GPUs tests/test_gpu_surface.py checks the library."""
    import ctypes as C
    import tests.build_harness as bh
    lib = C.CDLL(bh.build())
    out = (C.c_int * 4)()
    lib.jbh_device_guard_probe(2, 5, 0, 0, out)          # caller on device 2, handle on 5
    assert list(out) == [0, 5, 2, 2]                      # ran on 5, back on 2: one switch in, one out
    lib.jbh_device_guard_probe(2, 5, 0, 1, out)          # the entry point fails half way: still restored
    assert list(out) == [-3, 5, 2, 2]
    lib.jbh_device_guard_probe(3, 3, 0, 0, out)          # already current: no hipSetDevice at all
    assert list(out) == [0, 3, 3, 0]
    lib.jbh_device_guard_probe(1, 4, 1, 0, out)          # the switch itself fails: error out, nothing to restore
    assert out[0] != 0 and out[1] == -1 and out[2] == 1 and out[3] == 1
    src = open(bh.os.path.join(bh.HERE, "..", "jitterbug_amd", "csrc", "jb_api.hip")).read()
    assert "DeviceGuard<HipDeviceApi>" in src and "JB_HIP(hipSetDevice" not in src      # no entry point switches without the guard


def test_build_is_keyed_on_the_source_hash_not_on_file_times(tmp_path):
    """the library carries the sha256 of the sources it was built from; needs_build() compares that with the sources
    as they are (a comment edited in a copy of csrc/ makes it true, restoring the text makes it false again - whatever the mtimes), and
    _lib.load() refuses an in-tree library that was built from other sources."""
    import shutil
    from jitterbug_amd import _lib, build as B
    assert os.path.exists(B.OUT), "library not built"
    have = B.embedded_sha256(B.OUT)
    assert have is not None and len(have) == 64
    assert have == B.source_sha256() and not B.needs_build()
    assert _lib.load().jb_source_sha256().decode() == have
    csrc, inc = tmp_path / "csrc", tmp_path / "include"
    shutil.copytree(B.CSRC, csrc)
    shutil.copytree(B.INCLUDE, inc)
    assert B.source_sha256(str(csrc), str(inc)) == have and not B.needs_build(B.OUT, str(csrc), str(inc))       # a copy (new mtimes) is not a change
    f = csrc / "jb_task.hpp"
    text = f.read_text()
    f.write_text(text.replace("// jb_task.hpp", "// jb_task.hpp (edited)", 1))
    assert B.needs_build(B.OUT, str(csrc), str(inc))
    f.write_text(text)
    os.utime(f, (1, 1))                                                                                            # ... and an old mtime does not hide one
    assert not B.needs_build(B.OUT, str(csrc), str(inc))
    (inc / "jitterbug_hip.h").write_text((inc / "jitterbug_hip.h").read_text() + "\n/* edited */\n")
    assert B.needs_build(B.OUT, str(csrc), str(inc))
    # the loader's guard: same check, against the real tree
    stale = tmp_path / "libstale.so"
    data = open(B.OUT, "rb").read()
    stale.write_bytes(data.replace(b"JB_SRC_SHA256=" + have.encode(), b"JB_SRC_SHA256=" + b"0" * 64))
    assert B.embedded_sha256(str(stale)) == "0" * 64 and B.needs_build(str(stale))
