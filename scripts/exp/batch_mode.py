#!/usr/bin/env python3
"""A few batch proofs of vgg11 in one mode (for profiler runs): batch_mode.py <mode bits (int, e.g. 129 = DRIVE|FULL_IPA)> [lanes] [steps] [batches]"""
import os
import sys
import threading
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import zkcnn_amd as M  # noqa: E402
torch.cuda.init()
mode = int(sys.argv[1], 0)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 8
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nb = int(sys.argv[4]) if len(sys.argv) > 4 else 1
model, pic = os.environ.get("MODEL", "vgg11"), (32, 32, 3)
n = k * nb
ss = [None] * n


def build(i):
    ss[i] = M.Session(model, pic, 1)
th = [threading.Thread(target=build, args=(i,)) for i in range(n)]
[t.start() for t in th]
[t.join() for t in th]
for i, s in enumerate(ss[1:], 1):
    for ps in range(1000 * i, 1000 * i + 64):
        if s.new_image(ps)[0] == 0:
            break
for s in ss:
    s.prove(seed=1, mode=M.MODE_REUSE_GENS | M.MODE_DRIVE_ONLY, want_transcript=False)
    s.prove(seed=2, mode=M.MODE_REUSE_GENS | M.MODE_DRIVE_ONLY, want_transcript=False)
r, _ = ss[0].prove(seed=3, mode=mode, want_transcript=False)
print(f"[mode {mode:#x}] single stream {1e3 * (r.prove_s + r.poly_prove_s):.1f} ms", flush=True)
B = [M.BatchSession(ss[j * k:(j + 1) * k]) for j in range(nb)]
for b in B:
    b.prove(seeds=[10 + i for i in range(k)], mode=mode, want_transcript=False)


def run(j):
    for s in range(steps):
        B[j].prove(seeds=[1000 * j + 100 * s + i for i in range(k)], mode=mode, want_transcript=False)
torch.cuda.synchronize()
t = time.perf_counter()
th = [threading.Thread(target=run, args=(j,)) for j in range(nb)]
[x.start() for x in th]
[x.join() for x in th]
torch.cuda.synchronize()
dt = time.perf_counter() - t
print(f"[mode {mode:#x}] {nb} x {k} lanes: {1e3 * dt / steps:.1f} ms per step = {n * steps / dt:.1f} proofs/s", flush=True)
for b in B:
    b.close()
