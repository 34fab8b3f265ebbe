"""Throughput with K proofs in flight on ONE GPU (one session, host thread and HIP stream per proof).
Usage: concurrent_proofs.py [model] [K ...]"""
import hashlib, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkcnn_amd

model = sys.argv[1] if len(sys.argv) > 1 else "vgg11"
ks = [int(x) for x in sys.argv[2:]] or [1, 2, 4]
pic = (32, 32, 1) if model == "lenet" else (32, 32, 3)
kmax = max(ks)
sessions = [None] * kmax


def build(i):
    sessions[i] = zkcnn_amd.Session(model, pic, 1, data_seed=20260928 + i)


t0 = time.time()
th = [threading.Thread(target=build, args=(i,)) for i in range(kmax)]
[t.start() for t in th]; [t.join() for t in th]
print(f"{kmax} sessions built in {time.time() - t0:.1f} s")
mode = zkcnn_amd.MODE_DRIVE_ONLY | zkcnn_amd.MODE_REUSE_GENS
ref = []
for s in sessions:                       # warm-up + single-stream reference transcripts
    s.prove(seed=7, mode=zkcnn_amd.MODE_REUSE_GENS)
    r, t = s.prove(seed=8, mode=mode)
    ref.append((hashlib.sha256(t).hexdigest(), (r.prove_s + r.poly_prove_s) * 1e3))
print("single-stream prover ms:", [round(x[1], 1) for x in ref])
for k in ks:
    out = [None] * k
    def work(i, steps):
        time.sleep(i * float(os.environ.get("STAGGER_MS", "0")) * 1e-3)
        for _ in range(steps):
            out[i] = sessions[i].prove(seed=8, mode=mode)
    steps = 6
    t0 = time.time()
    th = [threading.Thread(target=work, args=(i, steps)) for i in range(k)]
    [t.start() for t in th]; [t.join() for t in th]
    dt = time.time() - t0
    same = all(hashlib.sha256(out[i][1]).hexdigest() == ref[i][0] for i in range(k))
    lat = [round((out[i][0].prove_s + out[i][0].poly_prove_s) * 1e3, 1) for i in range(k)]
    print(f"K={k}: {k * steps / dt:.2f} proofs/s  ({dt / steps * 1e3:.1f} ms per batch of {k}), per-proof prover ms {lat}, transcripts identical to single-stream: {same}")
for s in sessions: s.close()
