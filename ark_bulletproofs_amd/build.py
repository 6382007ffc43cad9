"""Builds the HIP engine in-tree: ark_bulletproofs_amd/libarkbp_hip.so (gfx950 only).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only container."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libarkbp_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["arkbp.hip"]
HEADERS = ["arkbp_params.h", "fp29.cuh", "ec.cuh", "msm.cuh", "host_math.hpp", "keccak_unrolled.inc", "r1cs_host.inc", os.path.join("..", "..", "include", "arkbp.h")]


def _deps():
    out = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    out += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".hpp", ".hip", ".h", ".inc"))]
    return sorted(set(out))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _deps())


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-Xarch_host", "-march=x86-64-v3",
           "-Wno-unused-result"] + (["-DARKBP_MSM_CH=" + os.environ["ARKBP_MSM_CH"]] if os.environ.get("ARKBP_MSM_CH") else []) + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
