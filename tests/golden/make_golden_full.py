"""Digests of the CPU oracle's transcripts for the two per-GPU circuits of BASELINE.json configs 4 / 5 at FULL size -- vgg11 with pic_cnt = 8 and
vgg16 with pic_cnt = 4 as ONE circuit each (reference src/models.cpp:43-146; FFT convolutions because pic_cnt > 1, src/neuralNetwork.cpp:46;
layer 0 = 2^26 entries) -- kept in tests/golden/full_size.json. tests/test_timed_path_gpu.py compares the GPU prover's
transcripts with them byte for byte (through SHA-256 + length); the oracle needs minutes and tens of GB per circuit, which is why this is a
committed fixture and not a test-time computation.

    python tests/golden/make_golden_full.py [vgg11_pp8] [vgg16_pp4]        (data seed 20260928; challenge seeds as in the test)

Modes: 0 = interactive with fresh seeded generators (drive-only: the transcript is the one the full verifier sees, test asserts that on the GPU side);
2 = ZKCNN_MODE_REUSE_GENS (public hash-to-curve generators)."""
import hashlib
import json
import os
import resource
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_ffi  # noqa: E402

# vgg16_pp32 = BASELINE configs[4] as the reference itself would run it (ONE circuit over 32 pictures, layer 0 = 2^28 entries): ~40 minutes and > 100 GB on
# one core -- made on the GPU box's host (`GOLDEN_RUNS=reuse_gens GOLDEN_OUT=gpurun_out/full_size_pp32.json python tests/golden/make_golden_full.py vgg16_pp32`
# through gpurun: the oracle is this repo's code and travels) and merged into full_size.json by hand
CASES = {"vgg11_pp8": ("vgg11", (32, 32, 3), 8), "vgg16_pp4": ("vgg16", (32, 32, 3), 4), "vgg16_pp32": ("vgg16", (32, 32, 3), 32)}
RUNS = [("interactive", 0x5EED0001, 0), ("reuse_gens", 0x5EED0007, 2)]
if os.environ.get("GOLDEN_RUNS"):
    RUNS = [r for r in RUNS if r[0] in os.environ["GOLDEN_RUNS"].split(",")]
DRIVE = 1

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "full_size.json")
out_path = os.environ.get("GOLDEN_OUT", path)
for key in (sys.argv[1:] or ["vgg11_pp8", "vgg16_pp4"]):
    model, pic, pp = CASES[key]
    t0 = time.time()
    entry = {"model": model, "pic": list(pic), "pic_cnt": pp, "data_seed": 20260928}
    with oracle_ffi.OracleSession(model, pic, pp, data_seed=20260928) as o:
        for name, seed, mode in RUNS:
            res, tr = o.prove(seed=seed, mode=mode | DRIVE)
            entry[name] = {"challenge_seed": seed, "mode": mode, "sha256": hashlib.sha256(tr).hexdigest(), "transcript_len": len(tr)}
            entry.update(n_layers=res.n_layers, input_bits=res.input_bits, n_rounds=res.n_rounds)
            print(key, name, entry[name]["sha256"][:16], len(tr), f"{time.time() - t0:.0f} s", flush=True)
    entry["oracle_wall_s"] = round(time.time() - t0)
    entry["oracle_peak_rss_gb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, 1)
    doc = json.load(open(out_path)) if os.path.exists(out_path) else {}
    doc.setdefault("full_size", {})[key] = entry
    json.dump(doc, open(out_path, "w"), indent=1, sort_keys=True)
