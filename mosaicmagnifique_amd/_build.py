"""In-tree build of the HIP engine for gfx950 (explicit hipcc, no JIT cache).

``build()`` produces ``mosaicmagnifique_amd/libmosaic_hip.so`` next to this
file; the file is git-ignored but travels with the repo snapshot to the GPU box.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmosaic_hip.so")
SOURCES = [os.path.join(CSRC, "engine.hip")]
HEADERS = [os.path.join(CSRC, "kernels.hpp"), os.path.join(CSRC, "colour_diff.hpp"),
           os.path.join(HERE, "..", "include", "mosaic_hip.h")]

# -ffp-contract=off: the arbiter kernels must not fuse a*b+c -- the reference's
# CPU path is an SSE2 build and summation / rounding order is part of the
# parity contract (csrc/colour_diff.hpp).
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-ffp-contract=off", "-Wall", "-Wextra"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP engine cannot be built")
    return exe


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS if os.path.exists(p))


def build(force=False, verbose=False):
    """Compile the engine if the library is missing or older than its sources."""
    if not force and not stale():
        return LIB
    cmd = [hipcc()] + HIPCC_FLAGS + ["-o", LIB + ".tmp"] + SOURCES
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n%s\n%s" % (res.stdout, res.stderr))
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
