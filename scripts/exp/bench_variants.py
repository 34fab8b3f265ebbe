import hashlib, os, sys, threading, time, queue
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zkcnn_amd
K = int(os.environ.get("KSTREAMS", "8"))
variant = sys.argv[1]
sessions = [None] * K
def build(i): sessions[i] = zkcnn_amd.Session("vgg11", (32, 32, 3), 1, data_seed=20260928 + i)
import numpy as np
def prewarm(n):
    """n contexts created and used side by side, then closed: the hardware queues exist before the sessions' streams are made"""
    def one(_):
        hc = zkcnn_amd.HipContext(0)
        a = np.ones((64, 4), dtype=np.uint64)
        hc.fr_binop("add", a, a)
        time.sleep(0.2)
        hc.close()
    tt = [threading.Thread(target=one, args=(j,)) for j in range(n)]
    [t.start() for t in tt]; [t.join() for t in tt]
if variant == "prewarm_allseq":
    prewarm(8)
    for i in range(K): build(i)
    th = []
elif variant == "allseq":
    for i in range(K): build(i)
    th = []
elif variant == "seqthreads":          # one builder thread per session, but one after the other
    for i in range(K):
        t = threading.Thread(target=build, args=(i,)); t.start(); t.join()
    th = []
elif variant in ("pairs", "quads"):    # two / four at a time
    g = 2 if variant == "pairs" else 4
    for i in range(0, K, g):
        tt = [threading.Thread(target=build, args=(j,)) for j in range(i, min(K, i + g))]
        [t.start() for t in tt]; [t.join() for t in tt]
    th = []
elif variant in ("seqfirst", "queue", "notranscript"):
    build(0)
    th = [threading.Thread(target=build, args=(i,)) for i in range(1, K)]
else:
    th = [threading.Thread(target=build, args=(i,)) for i in range(K)]
[t.start() for t in th]; [t.join() for t in th]
mode = zkcnn_amd.MODE_DRIVE_ONLY | zkcnn_amd.MODE_REUSE_GENS
def warm(i):
    sessions[i].prove(seed=7, mode=zkcnn_amd.MODE_REUSE_GENS, want_transcript=False)
    sessions[i].prove(seed=8, mode=mode, want_transcript=False)
th = [threading.Thread(target=warm, args=(i,)) for i in range(K)]
[t.start() for t in th]; [t.join() for t in th]
steps = 10
wt = variant != "notranscript"
done = [queue.Queue() for _ in range(K)]
def work(i):
    for k in range(steps):
        r = sessions[i].prove(seed=100 + k, mode=mode, want_transcript=wt)
        if variant in ("queue", "notranscript"): done[i].put(r)
t0 = time.time()
th = [threading.Thread(target=work, args=(i,)) for i in range(K)]
[t.start() for t in th]
if variant in ("queue", "notranscript"):
    for k in range(steps):
        batch = [done[i].get() for i in range(K)]
[t.join() for t in th]
dt = time.time() - t0
print(variant, f"{K * steps / dt:.2f} proofs/s")
