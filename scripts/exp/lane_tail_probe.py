"""Probe of the lanes' resident tail (ZKCNN_LANE_RESIDENT=1): k lanes of a small model prove in a batch; transcripts against the solo proofs, time, fusion report
(ZKCNN_BATCH_TRACE=1 prints the per-kernel launch counts when the batch is destroyed).   usage: python scripts/exp/lane_tail_probe.py [k] [model]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zkcnn_amd as M  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 2
model = sys.argv[2] if len(sys.argv) > 2 else "custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5"
pic = (32, 32, 3) if model.startswith("vgg") else (8, 8, 2)
REUSE = M.MODE_REUSE_GENS
a = M.Session(model, pic, 1)
ss = [a] + [a.clone() for _ in range(k - 1)]
seeds = [100 + i for i in range(k)]
solo = [s.prove(seed=seeds[i], mode=REUSE)[1] for i, s in enumerate(ss)]
solo = [s.prove(seed=seeds[i], mode=REUSE)[1] for i, s in enumerate(ss)]
with M.BatchSession(ss) as B:
    for rep in range(3):
        t0 = time.time()
        got = B.prove(seeds=seeds, mode=REUSE)
        dt = time.time() - t0
        ok = all(got[i][0].accepted == 1 and got[i][1] == solo[i] for i in range(k))
        print(f"batch proof {rep}: {1e3 * dt:.1f} ms, transcripts equal to the solo proofs: {ok}, rounds {got[0][0].n_rounds}, stats {B.stats()}", flush=True)
for s in ss:
    s.close()
