#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): runs the REFERENCE's own circuit generator (src/neuralNetwork.cpp, models.cpp, circuit.cpp,
utils.cpp, polynomial.cpp -- unmodified, reached through symlinks in a scratch directory; nothing is copied into the repository) on seeded
text inputs and writes per-layer SHA-256 digests of the gate lists, subset maps and witness values to tests/golden/gates.json.
The field is this repo's <hyrax-bls12-381/polyCommit.hpp> (the upstream submodule and mcl are absent): gate lists and subset maps are index
work and do not depend on it; the value digests check quantisation and evaluation order under the same arithmetic on both sides. This pins
host/neuralNetwork.cpp's EMISSION against the reference's; it does not pin the prover oracle (Oracle).
tests/test_gates_cpu.py rebuilds the same driver against host/*.cpp and compares."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/src"
DRIVER = os.path.join(ROOT, "tests", "gates", "gates_digest.cpp")

# (name, model, pic_cnt, network tokens or None, input seed)
CASES = [
    ("lenet_max_pp1", "lenet", 1, None, 11),
    ("lenet_avg_pp1", "lenet.avg", 1, None, 12),
    ("lenet_max_pp2_fft", "lenet", 2, None, 13),
    ("lenetCifar_pp1", "lenetCifar", 1, None, 14),
    ("vgg_small_direct_pp1", "vgg", 1, "4 M 8 M 8 8 M 8 M 8 M", 15),
    ("vgg_small_fft_pp2", "vgg", 2, "4 M 8 M 8 M 8 M 8 M", 16),
    ("vgg_small_fft_pp3_avg", "vgg", 3, "2 A 4 4 A 4 A 4 A 4 A", 17),
    ("vgg11_pp1", "vgg", 1, "64 M 128 M 256 256 M 512 512 M 512 512 M", 18),
]


def build_reference_driver(tmp):
    """<tmp>/x/y/src: symlinks to the reference's files, except the two files the product replaces -- prover.hpp / prover.cpp are THIS repo's
    (host/), with the headers they need; polyProver.hpp includes "../../../include/zkcnn_hip.h", hence the depth and <tmp>/include."""
    sdir = os.path.join(tmp, "x", "y", "src")
    os.makedirs(sdir)
    os.symlink(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
    for f in os.listdir(REF):
        if f not in ("prover.hpp", "prover.cpp"):
            os.symlink(os.path.join(REF, f), os.path.join(sdir, f))
    host = os.path.join(ROOT, "zkcnn_amd", "csrc", "host")
    for f in ("prover.hpp", "prover.cpp", "polyProver.hpp", "polyProver.cpp", "zk_mask.hpp"):
        os.symlink(os.path.join(host, f), os.path.join(sdir, f))
    exe = os.path.join(tmp, "gates_ref")
    srcs = [os.path.join(sdir, f) for f in ("neuralNetwork.cpp", "models.cpp", "circuit.cpp", "utils.cpp", "polynomial.cpp", "prover.cpp", "polyProver.cpp")]
    lib = os.path.join(ROOT, "zkcnn_amd", "lib")
    cmd = ["g++", "-std=c++17", "-O2", "-DGATES_REFERENCE", "-w", "-I", sdir, "-I", os.path.join(ROOT, "zkcnn_amd", "csrc"), DRIVER] + srcs + \
          ["-o", exe, "-pthread", "-L", lib, "-lzkcnn_hip", "-Wl,-rpath," + lib]
    subprocess.run(cmd, check=True)
    return exe


def build_own_driver(tmp):
    host = os.path.join(ROOT, "zkcnn_amd", "csrc", "host")
    exe = os.path.join(tmp, "gates_own")
    srcs = [os.path.join(host, f) for f in ("neuralNetwork.cpp", "models.cpp", "circuit.cpp", "utils.cpp")]
    cmd = ["g++", "-std=c++17", "-O2", "-w", "-I", host, "-I", os.path.join(ROOT, "zkcnn_amd", "csrc"), DRIVER] + srcs + ["-o", exe, "-pthread"]
    subprocess.run(cmd, check=True)
    return exe


def case_args(tmp, case):
    name, model, pp, tokens, seed = case
    args = [model, str(pp), os.path.join(tmp, f"in_{name}.txt")]
    if tokens:
        net = os.path.join(tmp, f"net_{name}.txt")
        open(net, "w").write(tokens + "\n")
        args.append(net)
    return args


def write_input(own_exe, tmp, case):
    """the statement's data file: this repo's seeded synthetic source (picture ~ U[0,1), weights ~ U(-k,k), k = 1/sqrt(fan_in)) written out in the
    reference's text format while this repo's generator draws from it (neuralNetwork::recordDataTo)"""
    r = subprocess.run([own_exe] + case_args(tmp, case), capture_output=True, text=True, env=dict(os.environ, GATES_WRITE_INPUT=str(case[4])))
    if r.returncode != 0:
        raise RuntimeError(f"{case[0]}: writing the input failed, rc {r.returncode}\n{r.stderr[-2000:]}")
    return json.loads(r.stdout)


def run_case(exe, tmp, case):
    r = subprocess.run([exe] + case_args(tmp, case), capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{case[0]}: rc {r.returncode}\n{r.stderr[-2000:]}")
    return json.loads(r.stdout)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    with tempfile.TemporaryDirectory() as tmp:
        exe = build_reference_driver(tmp)
        own = build_own_driver(tmp)
        out = {"_made_by": "tests/golden/make_gates_golden.py (reference generator, unmodified, through symlinks; field = this repo's polyCommit.hpp)", "cases": {}}
        for case in CASES:
            write_input(own, tmp, case)
            out["cases"][case[0]] = {"model": case[1], "pic_cnt": case[2], "network": case[3], "input_seed": case[4], "digest": run_case(exe, tmp, case)}
            print(case[0], out["cases"][case[0]]["digest"]["layers"], "layers", flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "gates.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
