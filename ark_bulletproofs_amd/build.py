"""Builds the HIP engine in-tree: ark_bulletproofs_amd/libarkbp_hip.so (gfx950 only).
hipcc cross-compiles without a GPU, so this also runs in the CPU-only container.
Two translation units (compiled side by side, linked into one library): arkbp.hip — context, MSM / IPA / R1CS kernels, host
orchestration, the C ABI — and vfe.hip — the verifier front end (codec, transcript sponge, challenge arithmetic)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libarkbp_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["arkbp.hip", "vfe.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Xarch_host", "-march=x86-64-v3", "-Wno-unused-result", "-Wno-c++20-extensions"]
# what each unit includes beyond the shared field / curve headers (a change there rebuilds only that unit)
ONLY = {"arkbp.hip": {"host_proto.hpp", "host_math.hpp", "keccak_unrolled.inc", "r1cs_host.inc", "pedersen.cuh", "glv.cuh", "ecq.cuh"}, "vfe.hip": set()}


def _headers():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".cuh", ".hpp", ".h", ".inc")))


def _deps(src):
    mine = [os.path.join(CSRC, src), os.path.join(HERE, "..", "include", "arkbp.h")]
    for h in _headers():
        if any(h in ONLY[o] for o in ONLY if o != src):
            continue
        mine.append(os.path.join(CSRC, h))
    return mine


def _obj(src):
    return os.path.join(OBJ, src.replace(".hip", ".o"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def needs_build():
    # the library against every source (the object files are not shipped to the GPU box: a current library needs none of them)
    return _stale(LIB, [d for s in SOURCES for d in _deps(s)])


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    extra = ["-DARKBP_MSM_CH=" + os.environ["ARKBP_MSM_CH"]] if os.environ.get("ARKBP_MSM_CH") else []
    procs = []
    for s in SOURCES:
        if force or _stale(_obj(s), _deps(s)):
            cmd = [HIPCC] + FLAGS + extra + ["-c", os.path.join(CSRC, s), "-o", _obj(s)]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise subprocess.CalledProcessError(p.returncode, cmd)
    if force or procs or _stale(LIB, [_obj(s) for s in SOURCES]):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fno-gpu-rdc", "-o", LIB] + [_obj(s) for s in SOURCES]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
