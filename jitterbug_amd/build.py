"""Builds libjitterbug_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m jitterbug_amd.build [--force]

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but travels with the working tree to the GPU box.

The library carries the sha256 of the sources it was built from (`jb_source_sha256()`, fed by -DJB_SRC_SHA): `needs_build()` compares
THAT with the hash of csrc/* + include/* as they are now - not file times, which a fresh checkout or a copied tree resets - and
`jitterbug_amd._lib.load()` refuses a library whose embedded hash differs from the sources next to it."""
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include")
OUT = os.path.join(HERE, "libjitterbug_hip.so")
SOURCES = ["jb_api.hip"]
SRC_EXT = (".hip", ".hpp", ".h")
# -fno-slp-vectorize: the SLP vectoriser turns the small fixed-size linear algebra into v_pk_fma_f32 fed by hundreds of
# register-shuffling v_mov (a third of the contact loop); scalar v_fma code is ~10 % shorter and has no such moves.
# -mllvm -disable-vector-combine: the packed-fp32 pairs of jb_lane.hpp's Pk2 are built from scalars that come out of small arrays (the
# preloaded constants); VectorCombine widens such a scalar load into a 2-wide load of TWO NEIGHBOURING array elements and overwrites one of
# them, overlapping loads that keep the array on the stack (72 scratch operations in the substep loop).  The pass has nothing else to do here.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-fno-slp-vectorize", "-mllvm", "-disable-vector-combine", "-Wno-unused-value"]
_TAG = b"JB_SRC_SHA256="


def hipcc_path():
    p = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(p):
        raise RuntimeError("hipcc not found (need ROCm): cannot build libjitterbug_hip.so")
    return p


def source_files(csrc=CSRC, include=INCLUDE):
    """Every file the library is compiled from: csrc/* and include/* (sorted, by name relative to its directory)."""
    out = []
    for tag, d in (("csrc", csrc), ("include", include)):
        if os.path.isdir(d):
            out += [(tag + "/" + f, os.path.join(d, f)) for f in sorted(os.listdir(d)) if f.endswith(SRC_EXT)]
    return out


def source_sha256(csrc=CSRC, include=INCLUDE):
    """sha256 over (name, content) of every source file, and of the compiler flags; None when the sources are not there."""
    files = source_files(csrc, include)
    if not files:
        return None
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for name, path in files:
        h.update(b"\0" + name.encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()


def embedded_sha256(lib=OUT):
    """The source hash a built library carries (read from the file, nothing is loaded); None if it has none."""
    try:
        data = open(lib, "rb").read()
    except OSError:
        return None
    m = re.search(re.escape(_TAG) + rb"([0-9a-f]{64})", data)
    return m.group(1).decode() if m else None


def needs_build(lib=OUT, csrc=CSRC, include=INCLUDE):
    if not os.path.exists(lib):
        return True
    return embedded_sha256(lib) != source_sha256(csrc, include)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return OUT
    sha = source_sha256()
    cmd = [hipcc_path()] + FLAGS + ['-DJB_SRC_SHA="%s"' % sha, "-o", OUT] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    assert embedded_sha256(OUT) == sha, "the built library does not carry the source hash it was given"
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
