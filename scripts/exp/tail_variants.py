import sys, time
sys.path.insert(0, ".")
import zkcnn_amd
D, R, HT, FS = zkcnn_amd.MODE_DRIVE_ONLY, zkcnn_amd.MODE_REUSE_GENS, zkcnn_amd.MODE_HOST_TAIL, zkcnn_amd.MODE_FIAT_SHAMIR
with zkcnn_amd.Session("vgg11", (32, 32, 3), 1) as s:
    for _ in range(3): s.prove(seed=1, mode=D | R, want_transcript=False)
    for name, mode in (("gpu rounds", D | R), ("hybrid tail", D | R | HT), ("FS device", FS | D), ("FS hybrid", FS | D | HT)):
        best = min((lambda r: (1e3 * (r.prove_s + r.poly_prove_s), 1e3 * r.prove_s))(s.prove(seed=3 + k, mode=mode, want_transcript=False)[0]) for k in range(4))
        print(f"{name:12s} prover {best[0]:.1f} ms (sumcheck {best[1]:.1f})")
