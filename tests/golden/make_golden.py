"""Regenerates tests/golden/transcripts.json from the CPU oracle (oracle/liboracle.so).

The reference ships no tests, fixtures or golden vectors for this path and cannot be built in this image
(its field/curve/commitment dependency is an empty submodule), so these are regression vectors of THIS
repo's restatement ("parity unpinned"): they pin the protocol bytes across refactors and are what
the GPU transcripts are compared with on a box that has no /root/reference.

    python tests/golden/make_golden.py
"""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import oracle_ffi  # noqa: E402

CASES = {
    "fc_only": ("custom:F8 F4", (4, 4, 1), 1),
    "naive_conv_maxpool": ("custom:C2:3:1:s M F4", (4, 4, 1), 1),
    "naive_conv_mul_add_avgpool": ("custom:C2:3:1:n A F4", (4, 4, 2), 1),
    "fft_conv_block_8x8_batch2": ("custom:C2:3:1:f M F4", (8, 8, 1), 2),
    "mixed_22_layers": ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2),
    "lenet5_pic1": ("lenet", (32, 32, 1), 1),
}

out = {}
for key, (model, pic, pp) in CASES.items():
    with oracle_ffi.OracleSession(model, pic, pp, data_seed=20260928) as o:
        res, tr = o.prove(seed=0x5EED0001)
        fres, ftr = o.prove(seed=0, mode=32)           # ZKCNN_MODE_FIAT_SHAMIR: challenges = SHA-256 of statement + messages
    assert res.accepted == 1 and fres.accepted == 1, key
    out[key] = {"model": model, "pic": list(pic), "pic_cnt": pp, "data_seed": 20260928, "challenge_seed": 0x5EED0001,
                "n_layers": res.n_layers, "input_size": res.input_size, "n_rounds": res.n_rounds,
                "transcript_len": len(tr), "sha256": hashlib.sha256(tr).hexdigest(),
                "fiat_shamir_sha256": hashlib.sha256(ftr).hexdigest()}
    print(key, out[key]["sha256"][:16], len(tr))
json.dump(out, open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "transcripts.json"), "w"), indent=1, sort_keys=True)
