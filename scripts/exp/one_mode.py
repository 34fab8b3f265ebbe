"""a few single-stream vgg11 proofs in one mode (for rocprofv3 runs): usage one_mode.py <mode bits, e.g. 0x63> [proofs]"""
import sys

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zkcnn_amd

mode = int(sys.argv[1], 0)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5
with zkcnn_amd.Session("vgg11", (32, 32, 3), 1) as s:
    for k in range(n):
        r, _ = s.prove(seed=0x5EED0100 + k, mode=mode, want_transcript=False)
        print("proof", k, "ms %.2f" % (1e3 * (r.prove_s + r.poly_prove_s)), flush=True)
