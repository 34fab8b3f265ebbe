"""Builds the native libraries in-tree (no pip, no JIT cache):

  zkcnn_amd/lib/libzkcnn_hip.so    HIP kernels + C-ABI of include/zkcnn_hip.h   (hipcc, gfx950)
  zkcnn_amd/lib/libzkcnn_host.so   C++14 host side: prover / verifier / circuit generator + include/zkcnn_api.h
  oracle/liboracle.so              CPU checker (test infrastructure; built by oracle/Makefile)

hipcc cross-compiles gfx950 without a GPU, so this runs in the CPU-only container too.
"""
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zkcnn_amd", "csrc")
LIB = os.path.join(ROOT, "zkcnn_amd", "lib")
OBJ = os.path.join(ROOT, "zkcnn_amd", "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

HIP_SRCS = ["hip/context.hip", "hip/upload.hip", "hip/sumcheck.hip", "hip/witness.hip", "hip/verifier.hip", "hip/hyrax.hip"]
HOST_SRCS = ["host/circuit.cpp", "host/utils.cpp", "host/neuralNetwork.cpp", "host/models.cpp", "host/prover.cpp",
             "host/polyProver.cpp", "host/api.cpp"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def _all_headers():
    out = []
    for base, _, files in os.walk(CSRC):
        out += [os.path.join(base, f) for f in files if f.endswith((".hpp", ".h", ".cuh"))]
    inc = os.path.join(ROOT, "include")
    out += [os.path.join(inc, f) for f in os.listdir(inc)]
    return out


def _run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def _run_all(cmds):
    """independent compilations side by side (one translation unit takes 0.5-2 minutes)"""
    from concurrent.futures import ThreadPoolExecutor
    if cmds:
        with ThreadPoolExecutor(max_workers=min(len(cmds), os.cpu_count() or 1)) as ex:
            list(ex.map(_run, cmds))


def build_hip(force=False):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _all_headers()
    objs, cmds = [], []
    for s in HIP_SRCS:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.basename(s) + ".o")
        if force or _newer(obj, [src] + hdrs):
            cmds.append([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-I" + CSRC, "-c", src, "-o", obj])
        objs.append(obj)
    _run_all(cmds)
    out = os.path.join(LIB, "libzkcnn_hip.so")
    if force or _newer(out, objs):
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


def build_host(force=False):
    os.makedirs(LIB, exist_ok=True)
    os.makedirs(OBJ, exist_ok=True)
    hdrs = _all_headers()
    cxx = os.environ.get("CXX", "g++")
    objs = []
    for s in HOST_SRCS:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, os.path.basename(s) + ".o")
        if force or _newer(obj, [src] + hdrs):
            _run([cxx, "-O3", "-DNDEBUG", "-std=c++14", "-fPIC", "-pthread", "-Wall", "-Wno-unused-function", "-I" + CSRC,
                  "-I" + os.path.join(CSRC, "host"), "-c", src, "-o", obj])
        objs.append(obj)
    out = os.path.join(LIB, "libzkcnn_host.so")
    if force or _newer(out, objs + [os.path.join(LIB, "libzkcnn_hip.so")]):
        _run([cxx, "-shared", "-pthread", "-o", out] + objs + ["-L" + LIB, "-lzkcnn_hip", "-Wl,-rpath,$ORIGIN", "-Wl,-Bsymbolic"])
    return out


def build_oracle(force=False):
    if force:
        shutil.rmtree(os.path.join(ROOT, "oracle", "build"), ignore_errors=True)
    _run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-j8"])
    return os.path.join(ROOT, "oracle", "liboracle.so")


def build_all(force=False):
    build_hip(force)
    build_host(force)
    build_oracle(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
