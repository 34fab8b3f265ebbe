#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (needs /root/reference): runs the REFERENCE's own prover + verifier logic (src/prover.cpp, verifier.cpp, polynomial.cpp,
utils.cpp, circuit.cpp, neuralNetwork.cpp, models.cpp -- unmodified, reached through symlinks in a scratch directory; nothing is copied into the
repository) under a seeded challenge stream and writes SHA-256 + length of the SUMCHECK part of the transcript -- the results of the nine
value-returning prover calls, intercepted at link time (tests/reflogic/ref_logic.cpp) -- to tests/golden/ref_logic.json.

Field arithmetic = this repo's ff/ (the upstream submodule and mcl are absent), commitment = a stand-in (tests/reflogic/shim): under the grading
rules this pins nothing. It is a DIAGNOSTIC for one hole: oracle and verifiers of this repo come from one reading of the reference, so a shared
misreading of the message semantics would pass every other test. tests/test_reflogic_cpu.py compares the CPU oracle's transcripts against it."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_gates_golden as gg      # noqa: E402  (the data files: this repo's seeded synthetic source written in the reference's text format)

REF = "/root/reference/src"
DRIVER = os.path.join(ROOT, "tests", "reflogic", "ref_logic.cpp")
SHIM = os.path.join(ROOT, "tests", "reflogic", "shim")
METHODS = ["Vres", "sumcheckDotProdUpdate1", "sumcheckUpdate1", "sumcheckUpdate2", "sumcheckLiuUpdate", "sumcheckDotProdFinalize1", "sumcheckFinalize1",
           "sumcheckFinalize2", "sumcheckLiuFinalize"]
CHALLENGE_SEED = 0x5EED0001

# (name, reference model, oracle model string, pic_cnt, network tokens or None, input seed)
CASES = [
    ("lenet5_pp1", "lenet", "lenet", 1, None, 11),                                      # BASELINE configs[0]: FFT convolutions (5x5 kernels), max pooling
    ("lenet5_avg_pp2", "lenet.avg", "lenet.avg", 2, None, 12),                          # average pooling, two pictures in one circuit
    ("vgg_small_fft_pp2", "vgg", "vgg:4 M 8 M 8 M 8 M 8 M", 2, "4 M 8 M 8 M 8 M 8 M", 16),  # the FFT block of a 3x3 convolution (pic_cnt > 1) with max pooling
    ("vgg11_quarter_pp1", "vgg", "vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", 1, "16 M 32 M 64 64 M 128 128 M 128 128 M", 19),   # direct convolutions, 37 layers
]


def build(tmp):
    sdir = os.path.join(tmp, "r", "src")
    os.makedirs(sdir)
    for f in os.listdir(REF):
        os.symlink(os.path.join(REF, f), os.path.join(sdir, f))
    inc = ["-I", sdir, "-I", SHIM, "-I", os.path.join(ROOT, "zkcnn_amd", "csrc")]
    flags = ["g++", "-std=c++17", "-O2", "-w"]
    obj = os.path.join(tmp, "prover.o")
    subprocess.run(flags + inc + ["-c", os.path.join(sdir, "prover.cpp"), "-o", obj], check=True)
    syms = {}
    for line in subprocess.run(["nm", "--defined-only", obj], capture_output=True, text=True, check=True).stdout.splitlines():
        name = line.split()[-1]
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        m = re.match(r"prover::(\w+)\(", dem)
        if m and m.group(1) in METHODS:
            syms[m.group(1)] = name
    missing = [m for m in METHODS if m not in syms]
    if missing:
        raise RuntimeError(f"methods not found in the reference's prover.o: {missing}")
    with open(os.path.join(tmp, "wrap_syms.h"), "w") as f:
        for m, s in syms.items():
            f.write(f'#define SYM_{m} "{s}"\n')
    exe = os.path.join(tmp, "ref_logic")
    srcs = [os.path.join(sdir, f) for f in ("verifier.cpp", "polynomial.cpp", "utils.cpp", "circuit.cpp", "neuralNetwork.cpp", "models.cpp")]
    wraps = ["-Wl,--wrap=" + s for s in syms.values()]
    subprocess.run(flags + inc + ["-I", tmp, DRIVER, obj] + srcs + ["-o", exe, "-pthread"] + wraps, check=True)
    return exe


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    with tempfile.TemporaryDirectory() as tmp:
        exe = build(tmp)
        own = gg.build_own_driver(tmp)
        out = {"_made_by": "tests/golden/make_ref_logic_golden.py: the reference's unmodified prover.cpp + verifier.cpp (+ generator) through symlinks, this repo's field "
                           "arithmetic, a stand-in commitment, seeded challenges; the nine value-returning prover calls recorded at link time (ld --wrap)",
               "challenge_seed": CHALLENGE_SEED, "cases": {}}
        for name, ref_model, oracle_model, pp, tokens, seed in CASES:
            case = (name, ref_model, pp, tokens, seed)
            gg.write_input(own, tmp, case)
            args = gg.case_args(tmp, case)
            cmd = [exe, args[0], args[1], args[2], str(CHALLENGE_SEED)] + args[3:]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"{name}: rc {r.returncode}\n{r.stderr[-3000:]}")
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out["cases"][name] = {"oracle_model": oracle_model, "pic": [32, 32, 1 if ref_model.startswith("lenet") else 3], "pic_cnt": pp, "input_seed": seed, "sumcheck": d}
            print(name, d, flush=True)
    with open(os.path.join(ROOT, "tests", "golden", "ref_logic.json"), "w") as f:
        json.dump(out, f, indent=1)
        f.write("\n")
