#!/usr/bin/env python3
"""Time against size within the profiler's kernel classes: one batch proof (8 lanes, alone on the GPU) with ZKCNN_PROF_DUMP, then per class a table
of launches by log2(algorithmic bytes): count, total ms, GB/s.  launch_sizes.py [model] [pic_cnt] [lanes] [classes,comma]"""
import collections
import math
import os
import sys
import tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
dump = tempfile.mktemp(prefix="zk_prof_", suffix=".txt")
os.environ["ZKCNN_PROF_DUMP"] = dump
import torch  # noqa: E402
import zkcnn_amd as M  # noqa: E402
torch.cuda.init()
model = sys.argv[1] if len(sys.argv) > 1 else "vgg11"
pp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
k = int(sys.argv[3]) if len(sys.argv) > 3 else 8
want = sys.argv[4].split(",") if len(sys.argv) > 4 else ["round_quad", "round_cubic", "fold", "gate_reduce", "round_fine"]
mode = int(os.environ["MODE_BITS"], 0) if os.environ.get("MODE_BITS") else (M.MODE_REUSE_GENS | M.MODE_DRIVE_ONLY | M.MODE_SEEDED)      # e.g. MODE_BITS=0x81: fresh generators, full argument
ss = [M.Session(model, (32, 32, 3), pp) for _ in range(k)]
for i, s in enumerate(ss[1:], 1):
    for ps in range(1000 * i, 1000 * i + 64):
        if s.new_image(ps)[0] == 0:
            break
for s in ss:
    s.prove(seed=1, mode=mode, want_transcript=False)
B = M.BatchSession(ss) if k > 1 else None
prove = (lambda sd: B.prove(seeds=[sd + i for i in range(k)], mode=mode, want_transcript=False)) if B else (lambda sd: ss[0].prove(seed=sd, mode=mode, want_transcript=False))
prove(10)
prove(20)
ss[0].profile("all")
ss[0].profile_report(reset=True)
open(dump, "w").close()
import time
t = time.perf_counter()
prove(30)
torch.cuda.synchronize()
dt = time.perf_counter() - t
ss[0].profile_report(reset=True)
rows = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0, 0.0]))
longest = collections.defaultdict(list)
for line in open(dump):
    c, b, ms = line.split()
    b, ms = float(b), float(ms)
    longest[c].append(ms)
    e = rows[c][int(math.log2(b)) if b > 0 else -1]
    e[0] += 1; e[1] += ms; e[2] += b
print(f"{model} pic_cnt={pp}, {k} lanes, one batch proof alone: {1e3 * dt:.1f} ms wall (HIP events on every launch)")
print("\nall classes: " + ", ".join(f"{c} {sum(e[1] for e in rows[c].values()):.2f} ms / {sum(e[0] for e in rows[c].values())}" for c in sorted(rows, key=lambda c: -sum(e[1] for e in rows[c].values()))))
for c in sorted(longest):
    ls = sorted(longest[c], reverse=True)
    print(f"{c}: longest launches (us): " + " ".join(f"{1e3 * x:.0f}" for x in ls[:16]) + f" | {sum(1 for x in ls if x < 0.03)} launches under 30 us = {sum(x for x in ls if x < 0.03):.2f} ms")
for c in want:
    if c not in rows:
        continue
    tot = sum(e[1] for e in rows[c].values())
    print(f"\n{c}: {sum(e[0] for e in rows[c].values())} launches, {tot:.3f} ms")
    print("| log2 bytes | launches | ms | avg us | GB/s |\n|---|---|---|---|---|")
    for lg in sorted(rows[c]):
        n, ms, b = rows[c][lg]
        print(f"| {lg} | {n} | {ms:.3f} | {1e3 * ms / n:.1f} | {b / ms / 1e6 if ms else 0:.0f} |")
if B:
    B.close()
os.unlink(dump)
