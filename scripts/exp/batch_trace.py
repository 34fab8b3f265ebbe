#!/usr/bin/env python3
"""Fusion report of one small batch (ZKCNN_BATCH_TRACE=1): which kernels were fused over how many lanes.  usage: batch_trace.py [model] [k]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["ZKCNN_BATCH_TRACE"] = "1"
import torch  # noqa: E402
import zkcnn_amd as M  # noqa: E402
torch.cuda.init()
model = sys.argv[1] if len(sys.argv) > 1 else "custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5"
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
pic = (32, 32, 3) if model.startswith("vgg") else (8, 8, 2)
first = M.Session(model, pic, 1)
ss = [first] + [M.Session(model, pic, 1, calibrated=first.statement()) for _ in range(k - 1)]
for i, s in enumerate(ss[1:], 1):
    for ps in range(1000 * i, 1000 * i + 64):
        if s.new_image(ps)[0] == 0:
            break
mode = M.MODE_REUSE_GENS | M.MODE_DRIVE_ONLY
for s in ss:
    s.prove(seed=1, mode=mode)
    s.prove(seed=2, mode=mode)
B = M.BatchSession(ss)
B.prove(seeds=list(range(k)), mode=mode)
print("stats after 1 proof", B.stats(), flush=True)
B.close()
B = M.BatchSession(ss)
B.prove(seeds=list(range(k)), mode=mode)
B.close()
