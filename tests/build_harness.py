"""Builds tests/_build/libjb_hostsim.so: the simulator source compiled for the host (test infrastructure)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build", "libjb_hostsim.so")
SRC = os.path.join(HERE, "host_harness.cpp")
DEPS = [SRC] + [os.path.join(HERE, "..", "jitterbug_amd", "csrc", f) for f in ("jb_sim.hpp", "jb_lane.hpp", "jb_model_build.hpp", "jb_device_guard.hpp")]


def build():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    if os.path.exists(OUT) and all(os.path.getmtime(d) <= os.path.getmtime(OUT) for d in DEPS):
        return OUT
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wno-unknown-pragmas", "-o", OUT, SRC])
    return OUT


if __name__ == "__main__":
    print(build())
