#!/usr/bin/env python3
"""A/B of an environment switch of the HIP library INSIDE one process: 8 lock-step batches of 8 lanes in the reference's semantics (bench.py's default shape), the
steps alternate between the two settings -- runs of bench.py on one box differ by more than most switches do (profiles/r06_variance.md), steps of one process do not.
usage: ab_inprocess.py VAR A B [repeats] [batches]      (a value "-" = variable unset)"""
import os
import sys
import threading
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch  # noqa: E402
import zkcnn_amd as M  # noqa: E402
torch.cuda.init()
var, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
nb = int(sys.argv[5]) if len(sys.argv) > 5 else 8
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
B = M.BatchSession.group([ss[j * k:(j + 1) * k] for j in range(nb)])
step = [0]


def one_step():
    st = step[0]
    step[0] += 1
    def run(j):
        B[j].prove(seeds=[(2 << 20) | ((j * k + i) << 10) | st for i in range(k)], mode=mode, want_transcript=True)
    th = [threading.Thread(target=run, args=(j,)) for j in range(nb)]
    t0 = time.perf_counter()
    [t.start() for t in th]
    [t.join() for t in th]
    return nb * k / (time.perf_counter() - t0)


def setv(v):
    if v == "-":
        os.environ.pop(var, None)
    else:
        os.environ[var] = v

for v in (va, vb):
    setv(v)
    one_step()
res = {va: [], vb: []}
for r in range(reps):
    for v in ((va, vb) if r % 2 == 0 else (vb, va)):
        setv(v)
        res[v].append(one_step())
for v in (va, vb):
    xs = sorted(res[v])
    print(f"{var}={v}: median {xs[len(xs) // 2]:.1f} proofs/s, all {[round(x, 1) for x in res[v]]}", flush=True)
for b in B:
    b.close()
