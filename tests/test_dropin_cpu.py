"""Drop-in check of the boundary: the reference's UNMODIFIED callers of the hot path -- verifier.cpp, neuralNetwork.cpp,
models.cpp, the two main()s and the helpers they pull in -- must type-check against THIS repo's `prover.hpp` (the HIP-backed class with
the public surface of reference src/prover.hpp:18-49) and THIS repo's `<hyrax-bls12-381/polyCommit.hpp>` (the header the reference
takes its field, curve, timer and commitment types from; upstream submodule is empty).

Only runs where /root/reference exists (this container); the reference sources never travel: the test builds a scratch directory of
SYMLINKS in which `prover.hpp` points at our header and everything else at the reference's files, and runs `g++ -fsyntax-only`.
Nothing is compiled to an object, nothing is copied."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/src"
CALLERS = ["verifier.cpp", "neuralNetwork.cpp", "models.cpp", "main_demo_lenet.cpp", "main_demo_vgg.cpp", "circuit.cpp", "utils.cpp",
           "polynomial.cpp"]


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference sources are only present in the build container")
@pytest.mark.parametrize("src", CALLERS)
def test_reference_callers_typecheck_against_our_prover_header(tmp_path, src):
    # <scratch>/x/y/src/<symlinks>; our polyProver.hpp includes "../../../include/zkcnn_hip.h", hence the depth and <scratch>/include
    sdir = tmp_path / "x" / "y" / "src"
    sdir.mkdir(parents=True)
    os.symlink(os.path.join(ROOT, "include"), tmp_path / "include")
    for f in os.listdir(REF):
        if f in ("prover.hpp", "prover.cpp"):
            continue                                     # the two files the product replaces
        os.symlink(os.path.join(REF, f), sdir / f)
    host = os.path.join(ROOT, "zkcnn_amd", "csrc", "host")
    os.symlink(os.path.join(host, "prover.hpp"), sdir / "prover.hpp")
    os.symlink(os.path.join(host, "polyProver.hpp"), sdir / "polyProver.hpp")
    os.symlink(os.path.join(host, "zk_mask.hpp"), sdir / "zk_mask.hpp")        # additive zero-knowledge mixin; compiles against the reference's circuit.h / polynomial.h
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-I", str(sdir), "-I", os.path.join(ROOT, "zkcnn_amd", "csrc"), str(sdir / src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_prover_header_keeps_the_reference_method_list():
    """travels to the GPU box too: the public method names of reference src/prover.hpp:18-49 are all declared in our header"""
    text = open(os.path.join(ROOT, "zkcnn_amd", "csrc", "host", "prover.hpp")).read()
    for name in ["init", "sumcheckInitAll", "sumcheckInit", "sumcheckDotProdInitPhase1", "sumcheckInitPhase1", "sumcheckInitPhase2",
                 "sumcheckDotProdUpdate1", "sumcheckUpdate1", "sumcheckUpdate2", "Vres", "sumcheckDotProdFinalize1", "sumcheckFinalize1",
                 "sumcheckFinalize2", "sumcheckLiuFinalize", "sumcheckLiuInit", "sumcheckLiuUpdate", "commitInput", "proveTime",
                 "proofSize", "polyProverTime", "polyProofSize", "prove_timer", "layeredCircuit C", "vector<vector<F>> val"]:
        assert name in text, name
