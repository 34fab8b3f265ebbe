"""experiment: does a small kernel kept running on another stream (GPU clocks stay up) shorten the interactive rounds of one proof?"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zkcnn_amd
s = zkcnn_amd.Session("vgg11", (32, 32, 3), 1)
mode = zkcnn_amd.MODE_DRIVE_ONLY | zkcnn_amd.MODE_REUSE_GENS
for _ in range(3): s.prove(seed=1, mode=mode, want_transcript=False)
def lat(n=6):
    t = 0
    for k in range(n):
        r, _ = s.prove(seed=10 + k, mode=mode, want_transcript=False)
        t += r.prove_s + r.poly_prove_s
    return 1e3 * t / n
print("alone: %.2f ms per proof" % lat())
stop = False
def spin(threads):
    hc = zkcnn_amd.HipContext(0)
    while not stop:
        hc.bench_fr_mul(threads, 20000, 1)
    hc.close()
for threads in (64, 64 * 64, 64 * 1024):
    stop = False
    th = threading.Thread(target=spin, args=(threads,)); th.start()
    time.sleep(0.5)
    print("with a spinner of %d threads: %.2f ms per proof" % (threads, lat()))
    stop = True; th.join()
s.close()
