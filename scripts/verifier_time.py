"""Full verification time of one vgg11 proof: wiring predicates and the opening's MSMs on the GPU (default) vs the reference's host loops."""
import sys
import time
sys.path.insert(0, ".")
import zkcnn_amd

with zkcnn_amd.Session("vgg11", (32, 32, 3), 1) as s:
    for name, mode in (("GPU predicates + GPU MSMs", 0), ("again", 0), ("cross-checked against the host", zkcnn_amd.MODE_CROSS_PRED), ("host loops (reference)", zkcnn_amd.MODE_HOST_PRED)):
        t0 = time.time()
        r, _ = s.prove(seed=1, mode=mode | zkcnn_amd.MODE_REUSE_GENS, want_transcript=False)
        print(f"{name:34s} accepted {r.accepted}  verifier {r.verify_s:.3f} s + opening check {r.poly_verify_s:.3f} s   (wall {time.time() - t0:.2f} s)")
