#!/usr/bin/env python3
"""Single-stream proofs of vgg11 in the REFERENCE's semantics (fresh random generators per proof, argument down to length 1) with the
per-class profile of the last one: fresh_mode.py [proofs]. Run under rocprofv3 --kernel-trace --stats for the per-kernel table."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import zkcnn_amd as M  # noqa: E402
torch.cuda.init()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
s = M.Session(os.environ.get("MODEL", "vgg11"), (32, 32, 3), 1)
drive = M.MODE_DRIVE_ONLY
for k in range(2):
    s.prove(seed=1 + k, mode=drive | M.MODE_REUSE_GENS, want_transcript=False)
for mode, name in ((drive | M.MODE_REUSE_GENS | M.MODE_FULL_IPA, "session generators, full argument"), (drive | M.MODE_FULL_IPA, "fresh generators, full argument")):
    best = None
    for k in range(n):
        r, _ = s.prove(seed=100 + k, mode=mode, want_transcript=False)
        ms = 1e3 * (r.prove_s + r.poly_prove_s)
        if best is None or ms < best[0]:
            best = (ms, 1e3 * r.prove_s, 1e3 * r.poly_prove_s)
    print(f"[{name}] prover {best[0]:.2f} ms = sumcheck {best[1]:.2f} + commitment/opening {best[2]:.2f}", flush=True)
    if os.environ.get("CLASSES", "1") == "1":
        s.profile("all")
        s.profile_report(reset=True)
        r, _ = s.prove(seed=200, mode=mode, want_transcript=False)
        print(f"  with HIP events on every launch: {1e3 * (r.prove_s + r.poly_prove_s):.2f} ms; classes:", flush=True)
        rep = s.profile_report(reset=True)
        for c, v in sorted(rep.items(), key=lambda kv: -(kv[1].get("ms", 0) if isinstance(kv[1], dict) else 0)):
            print("   ", c, json.dumps(v), flush=True)
        s.profile(None)
s.close()
