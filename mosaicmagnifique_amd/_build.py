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
# Two translation units with different floating-point rules:
#  engine.hip  -ffp-contract=off : the fp64 arbiter kernels must not fuse a*b+c -- the
#              reference's CPU path is an SSE2 build and summation / rounding order is
#              part of the parity contract (csrc/colour_diff.hpp);
#  filter.hip  default contraction: the fp32 filter pass wants fused multiply-adds;
#              -fno-slp-vectorize: on gfx950 a v_pk_*_f32 costs two issue slots, so SLP
#              packing buys nothing and adds register-pairing moves (10 % of the kernel).
#  group.hip   host code only (one process, several GPUs): the RCCL all-gather of the candidate lists.
UNITS = [("engine.hip", ["-ffp-contract=off"]),
         ("filter.hip", ["-ffp-contract=fast", "-fno-slp-vectorize"]),
         ("group.hip", [])]
ROCM_LIB = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib")
LINK_FLAGS = ["-L" + ROCM_LIB, "-lrccl", "-Wl,-rpath," + ROCM_LIB, "-lpthread"]
SOURCES = [os.path.join(CSRC, name) for name, _ in UNITS]
HEADERS = [os.path.join(CSRC, h) for h in ("kernels.hpp", "colour_diff.hpp", "colour_diff_fast.hpp", "colour_diff_fp16.hpp",
                                           "filter_pretier.hpp", "launch.hpp")] + \
          [os.path.join(HERE, "..", "include", h) for h in ("mosaic_hip.h", "mosaic_hip_tuning.h")]

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra"]


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
    objs = []
    for name, extra in UNITS:
        obj = os.path.join(CSRC, name + ".o")
        _run([hipcc()] + HIPCC_FLAGS + extra + ["-c", os.path.join(CSRC, name), "-o", obj], verbose)
        objs.append(obj)
    _run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp"] + objs + LINK_FLAGS, verbose)
    os.replace(LIB + ".tmp", LIB)
    return LIB


def _run(cmd, verbose):
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n%s\n%s" % (res.stdout, res.stderr))


if __name__ == "__main__":
    print(build(force=True, verbose=True))
