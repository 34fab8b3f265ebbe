"""Soak test: K sessions prove concurrently, EVERY proof fully verified (no drive-only). Usage: stress_verify.py [proofs per stream] [K]
STRESS_NEW_IMAGE=1: every other proof is preceded by zkcnn_session_new_image (witness recomputed in HBM)."""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkcnn_amd
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
models = [("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1), ("lenet", (32, 32, 1), 1), ("custom:C3:3:1:f A C2:3:1:f M F6 F3", (8, 8, 1), 3),
          ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1)]
bad = []
def work(i):
    m = models[i % len(models)]
    with zkcnn_amd.Session(m[0], m[1], m[2], data_seed=500 + i) as s:
        for k in range(n):
            if os.environ.get("STRESS_NEW_IMAGE") and k % 2:          # every other proof is of a new picture; pictures whose scales differ are skipped
                ok = False
                for p in range(60):
                    if s.new_image(7000 + 1000 * i + 60 * k + p)[0] == 0:
                        ok = True
                        break
                if not ok:
                    print(f"stream {i}: no picture with this circuit's scales among 60, stopping at proof {k}")
                    return
            r, _ = s.prove(seed=1000 * i + k, mode=zkcnn_amd.MODE_REUSE_GENS, want_transcript=False)
            if r.accepted != 1:
                bad.append((i, k, r.message.decode()))
t0 = time.time()
th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
[t.start() for t in th]; [t.join() for t in th]
print(f"{K} streams x {n} fully verified proofs in {time.time() - t0:.1f} s, rejected: {len(bad)} {bad[:3]}")
sys.exit(1 if bad else 0)
