#!/usr/bin/env python3
"""Fusion report (ZKCNN_BATCH_TRACE=1) of lock-step batches in the reference's semantics, made the way bench.py makes them (clones of one session, a picture each,
solo warm-up proofs first, then B batches on B threads): which kernels were NOT fused over all lanes.  usage: batch_trace_fresh.py [batches] [steps]"""
import os
import sys
import threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ZKCNN_BATCH_TRACE"] = "1"
import torch  # noqa: E402
import zkcnn_amd as M  # noqa: E402
torch.cuda.init()
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
k = 8
first = M.Session("vgg11", (32, 32, 3), 1)
ss = [first] + [first.clone() for _ in range(nb * k - 1)]
for i, s in enumerate(ss[1:], 1):
    for ps in range(1000 * i, 1000 * i + 64):
        if s.new_image(ps)[0] == 0:
            break
mode = M.MODE_DRIVE_ONLY | M.MODE_FULL_IPA
for i, s in enumerate(ss):
    s.prove(seed=(1 << 20) | (i << 10), mode=mode, want_transcript=False)
    s.prove(seed=(1 << 20) | (i << 10) | 1, mode=mode, want_transcript=False)
B = [M.BatchSession(ss[j * k:(j + 1) * k]) for j in range(nb)]


def run(j):
    for st in range(steps + 1):
        B[j].prove(seeds=[(2 << 20) | ((j * k + i) << 10) | st for i in range(k)], mode=mode, want_transcript=True)
        if st == 0:
            print(f"batch {j}: stats after the first batch proof", B[j].stats(), flush=True)
th = [threading.Thread(target=run, args=(j,)) for j in range(nb)]
[t.start() for t in th]
[t.join() for t in th]
for j in range(nb):
    print(f"batch {j}: stats at the end", B[j].stats(), flush=True)
    B[j].close()
