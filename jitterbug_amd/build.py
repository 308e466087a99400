"""Builds libjitterbug_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m jitterbug_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels
with the working tree to the GPU box."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libjitterbug_hip.so")
SOURCES = ["jb_api.hip"]
DEPS = ["jb_api.hip", "jb_sim.hpp", "jb_step.hpp", "jb_task.hpp", "jb_lane.hpp", "jb_model_build.hpp", "jb_model_compile.hpp", "jb_model_compile_types.h", "jb_nominal_spec.h", "jb_default_params.h", "jb_device_guard.hpp",
        os.path.join("..", "..", "include", "jitterbug_hip.h"), os.path.join("..", "..", "include", "jitterbug_model.h")]
# -fno-slp-vectorize: the SLP vectoriser turns the small fixed-size linear algebra into v_pk_fma_f32 fed by hundreds of
# register-shuffling v_mov (a third of the contact loop); scalar v_fma code is ~10 % shorter and has no such moves.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize", "-Wno-unused-value"]


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found (need ROCm): cannot build libjitterbug_hip.so")
    return p


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    cmd = [hipcc_path()] + FLAGS + ["-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
