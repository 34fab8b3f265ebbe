"""verifier wall time with the wiring predicates on the GPU vs on the host (reference behaviour), one circuit. Usage: verifier_time.py [model]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkcnn_amd
model = sys.argv[1] if len(sys.argv) > 1 else "vgg11"
pic = (32, 32, 1) if model.startswith("lenet") and "Cifar" not in model else (32, 32, 3)
with zkcnn_amd.Session(model, pic, 1) as s:
    s.prove(seed=1, mode=zkcnn_amd.MODE_REUSE_GENS)
    for name, mode in (("gpu predicates", 0), ("host predicates", zkcnn_amd.MODE_HOST_PRED)):
        r, _ = s.prove(seed=2, mode=mode | zkcnn_amd.MODE_REUSE_GENS)
        print(f"{model} {name}: accepted={r.accepted} verifier {r.verify_s:.3f} s (+ opening check {r.poly_verify_s:.3f} s), prover {r.prove_s + r.poly_prove_s:.3f} s, wall {r.wall_s:.3f} s")
