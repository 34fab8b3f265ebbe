"""k_msm_codes time per vgg11 proof with and without the subset-sum path for rows of bits (ZKCNN_MSM_BITS)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zkcnn_amd

with zkcnn_amd.Session("vgg11", (32, 32, 3), 1) as s:
    mode = zkcnn_amd.MODE_REUSE_GENS | zkcnn_amd.MODE_DRIVE_ONLY
    for k in range(3):
        s.prove(seed=k, mode=mode, want_transcript=False)
    s.profile(["msm_planes", "msm_finish"])
    for k in range(3):
        r, _ = s.prove(seed=10 + k, mode=mode, want_transcript=False)
    rep = s.profile_report()
    print({k: v for k, v in rep.items() if v["launches"]}, "commit ms %.2f" % (1e3 * r.poly_prove_s))
