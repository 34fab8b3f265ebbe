#!/usr/bin/env python3
"""Soak test of the resident round kernels: one session per model, N iterations of
   interactive proof on the resident kernels (fully verified)  ==  the same seed with a launch per round (ZKCNN_MODE_HOST_ROUNDS)
   Fiat-Shamir proof with device-side rounds (MODE_FS_DEVICE)  ==  the same with host-driven rounds  ==  the default (host hash, resident kernels)
   a proof with a corrupted message (the verifier stops calling in the middle of a phase: the resident kernel is sent home)
and, in a second part, K sessions proving side by side (a launch per round by policy) against their single-stream transcripts.
usage: soak_resident.py [iterations] [K]     exit code 1 on the first difference"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("ZKCNN_TEST_HOOKS", "1")
import zkcnn_amd  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 20
K = int(sys.argv[2]) if len(sys.argv) > 2 else 4
REUSE, DRIVE, HOST, FS, TAMPER = (zkcnn_amd.MODE_REUSE_GENS, zkcnn_amd.MODE_DRIVE_ONLY, zkcnn_amd.MODE_HOST_ROUNDS, zkcnn_amd.MODE_FIAT_SHAMIR,
                                   zkcnn_amd.MODE_TAMPER)
MODELS = [("lenet", (32, 32, 1), 1), ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1), ("vgg11", (32, 32, 3), 1)]
t0 = time.time()
for model, pic, pp in MODELS:
    with zkcnn_amd.Session(model, pic, pp) as s:
        n_msg = None
        for it in range(N if model != "vgg11" else max(3, N // 4)):
            seed = 0x5EED7000 + it
            res, live = s.prove(seed=seed, mode=REUSE)
            assert res.accepted == 1, (model, it, res.message.decode())
            n_msg = res.n_messages
            _, host = s.prove(seed=seed, mode=REUSE | DRIVE | HOST)
            if live != host:
                sys.exit(f"{model} iteration {it}: resident kernels and launch-per-round transcripts differ")
            _, fs_dev = s.prove(mode=FS | zkcnn_amd.MODE_FS_DEVICE | DRIVE)
            _, fs_host = s.prove(mode=FS | DRIVE | HOST)
            if fs_dev != fs_host:
                sys.exit(f"{model} iteration {it}: Fiat-Shamir device rounds and host rounds differ")
            _, fs_res = s.prove(mode=FS | DRIVE)          # the default: challenges hashed on the host, rounds in the resident kernels
            if fs_res != fs_host:
                sys.exit(f"{model} iteration {it}: Fiat-Shamir over the resident kernels and host rounds differ")
            bad, _ = s.prove(seed=seed, mode=REUSE | TAMPER | (((it * 37) % max(n_msg - 4, 1) + 2) << 8))
            _, again = s.prove(seed=seed, mode=REUSE | DRIVE)
            if again != live:
                sys.exit(f"{model} iteration {it}: transcript after an aborted proof differs")
        print(f"{model}: {it + 1} iterations ok, resident rounds so far {s.fs_stats()}", flush=True)

# K sessions side by side on one shared circuit
model, pic, pp = MODELS[1]
first = zkcnn_amd.Session(model, pic, pp)
want = first.prove(seed=0x5EED7100, mode=REUSE | DRIVE)[1]
sessions = [first] + [zkcnn_amd.Session(model, pic, pp) for _ in range(K - 1)]
errs = []


def run(s):
    try:
        for it in range(N):
            if s.prove(seed=0x5EED7100, mode=REUSE | DRIVE)[1] != want:
                errs.append("transcript differs under concurrency")
                return
    except BaseException as e:      # noqa: BLE001
        errs.append(repr(e))


th = [threading.Thread(target=run, args=(s,)) for s in sessions]
[t.start() for t in th]
[t.join() for t in th]
for s in sessions:
    s.close()
if errs:
    sys.exit(f"{K} sessions side by side: {errs[0]}")
print(f"{K} sessions x {N} proofs side by side ok; sharing {zkcnn_amd.sharing_stats()}; {time.time() - t0:.0f} s")
