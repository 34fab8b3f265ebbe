"""Lone-proof latency against the hand-over size of the resident tail (ZKCNN_SOLO_EXPORT=<log>, -1 = the kernel runs every phase to its end): vgg11 and a small FFT-convolution
model, transcripts' SHA-256 (must not depend on the setting).   usage: ZKCNN_SOLO_EXPORT=6 python scripts/exp/solo_export_probe.py"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zkcnn_amd as M  # noqa: E402

mode = M.MODE_REUSE_GENS | M.MODE_DRIVE_ONLY
for model, pic, pp in (("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2), ("lenet", (32, 32, 1), 1), ("vgg11", (32, 32, 3), 1)):
    with M.Session(model, pic, pp) as s:
        ok = s.prove(seed=7, mode=M.MODE_REUSE_GENS)[0].accepted
        best, sha = 1e9, None
        for k in range(6):
            r, tr = s.prove(seed=11, mode=mode)
            best = min(best, r.prove_s + r.poly_prove_s)
            sha = hashlib.sha256(tr).hexdigest()[:16]
        print(f"ZKCNN_SOLO_EXPORT={os.environ.get('ZKCNN_SOLO_EXPORT', 'policy')} {model[:12]:12s} accepted {ok} best {1e3 * best:.2f} ms (sumcheck {1e3 * r.prove_s:.2f}) sha {sha} host rounds {s.host_tail_rounds()}", flush=True)
