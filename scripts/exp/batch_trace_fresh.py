import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["ZKCNN_BATCH_TRACE"] = "1"
import torch, zkcnn_amd as M
torch.cuda.init()
k = 8
first = M.Session("vgg11", (32, 32, 3), 1)
ss = [first] + [first.clone() for _ in range(k - 1)]
for i, s in enumerate(ss[1:], 1):
    for ps in range(1000 * i, 1000 * i + 64):
        if s.new_image(ps)[0] == 0:
            break
for s in ss:
    s.prove(seed=1, mode=M.MODE_REUSE_GENS | M.MODE_DRIVE_ONLY)
mode = M.MODE_DRIVE_ONLY | M.MODE_FULL_IPA
B = M.BatchSession(ss)
B.prove(seeds=[100 + i for i in range(k)], mode=mode)
B.close()
B = M.BatchSession(ss)
B.prove(seeds=[200 + i for i in range(k)], mode=mode)
B.close()
