"""Independent check of the circuit generator (reference src/neuralNetwork.cpp:60-650, src/circuit.cpp:4-88).

tests/golden/gates.json holds, per layer, SHA-256 digests of the gate lists (emission order), the subset maps `initSubset` derives from them,
the layer shapes and the witness values, as the REFERENCE's own unmodified generator emits them (tests/golden/make_gates_golden.py: build container
only; the reference sources are reached through symlinks and never enter the repository). This test compiles the same driver
(tests/gates/gates_digest.cpp) against THIS repo's host/neuralNetwork.cpp / models.cpp / circuit.cpp / utils.cpp and compares every field.

What it pins: gate emission, first-use numbering, sizes and bit lengths (index work, independent of the field), and -- under this repo's field
arithmetic on both sides -- quantisation, truncation and evaluation order of the witness. What it does NOT pin: the prover oracle (the reference's
prover.cpp never ran here: hyrax-bls12-381 / mcl are absent); DESIGN.md, "Oracle", says so."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("make_gates_golden", os.path.join(ROOT, "tests", "golden", "make_gates_golden.py"))
mk = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mk)
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "gates.json")))


@pytest.fixture(scope="module")
def own_driver(tmp_path_factory):
    tmp = str(tmp_path_factory.mktemp("gates"))
    return mk.build_own_driver(tmp), tmp


def test_golden_covers_the_generator_cases():
    assert set(GOLD["cases"]) == {c[0] for c in mk.CASES}
    for c in mk.CASES:
        g = GOLD["cases"][c[0]]
        assert (g["model"], g["pic_cnt"], g["network"], g["input_seed"]) == (c[1], c[2], c[3], c[4])
    kinds = {L["ty"] for g in GOLD["cases"].values() for L in g["digest"]["layer"]}
    # INPUT, FFT, IFFT, ADD_BIAS, RELU, AVG_POOL (8), MAX_POOL (7), DOT_PROD (9), PADDING (10), FCONN (11), NCONV (12)
    assert {0, 1, 2, 3, 4, 7, 8, 9, 10, 11, 12} <= kinds


@pytest.mark.parametrize("case", mk.CASES, ids=[c[0] for c in mk.CASES])
def test_generator_emits_what_the_reference_generator_emits(own_driver, case):
    exe, tmp = own_driver
    synth = mk.write_input(exe, tmp, case)          # also: the circuit built straight from the synthetic source
    got = mk.run_case(exe, tmp, case)               # ... and from the data file it wrote
    want = GOLD["cases"][case[0]]["digest"]
    assert got["layers"] == want["layers"] and got["two_mul"] == want["two_mul"]
    for a, b in zip(got["layer"], want["layer"]):
        diff = {k: (a[k], b[k]) for k in b if a[k] != b[k]}
        assert not diff, f"layer {b['i']} (type {b['ty']}): {diff}"
    assert synth == got, "the data file does not reproduce the synthetic source's circuit"
