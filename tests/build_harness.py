"""Builds tests/_build/libjb_hostsim.so: the simulator source compiled for the host (test infrastructure)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libjb_hostsim.so")
SRC = os.path.join(HERE, "host_harness.cpp")
DEPS = [SRC] + [os.path.join(HERE, "..", "jitterbug_amd", "csrc", f) for f in ("jb_sim.hpp", "jb_lane.hpp", "jb_step.hpp", "jb_task.hpp", "jb_model_build.hpp", "jb_device_guard.hpp", "jb_witness.hpp")]


def build(row_k=None):
    """row_k: build with a row cache of that many contacts (-DJB_ROW_K): a tiny cache sends almost every contact through the
    beyond-the-cache path (candidate kept, rows recomputed per pass), which ordinary rollouts hardly ever reach."""
    out = OUT if row_k is None else OUT.replace(".so", "_rowk%d.so" % row_k)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if os.path.exists(out) and all(os.path.getmtime(d) <= os.path.getmtime(out) for d in DEPS):
        return out
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas"] + (["-DJB_ROW_K=%d" % row_k] if row_k is not None else []) + ["-o", out, SRC])
    return out


if __name__ == "__main__":
    print(build())
