#!/usr/bin/env python3
"""Experiment: proofs/s of B lock-step batches of K lanes each (one host thread and one HIP stream per batch) on one GPU.
   python scripts/exp/batch_throughput.py --workload vgg11 --configs 1x8,2x8,3x8,4x4 --steps 6"""
import argparse
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")
import torch  # noqa: E402  (its HIP runtime first, as bench.py does)
import zkcnn_amd  # noqa: E402

WORKLOADS = {"vgg11": ("vgg11", (32, 32, 3), 1), "lenet": ("lenet", (32, 32, 1), 1),
             "vgg11_quarter": ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1)}


def in_threads(fn, n):
    errs = []

    def run(i):
        try:
            fn(i)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    th = [threading.Thread(target=run, args=(i,)) for i in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    if errs:
        raise errs[0]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="vgg11")
    ap.add_argument("--configs", default="1x8,2x8")
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--singles", type=int, default=0, help="also time this many independent sessions on threads (the round-3 shape)")
    args = ap.parse_args()
    torch.cuda.init()
    model, pic, pp = WORKLOADS[args.workload]
    configs = [tuple(int(x) for x in c.split("x")) for c in args.configs.split(",")]
    n = max(max(b * k for b, k in configs), args.singles)
    sessions = [None] * n
    t0 = time.time()

    def build(i):
        sessions[i] = zkcnn_amd.Session(model, pic, pp)
    in_threads(build, n)

    def pictures(i):
        if i == 0:
            return
        for p in range(1, 64):
            if sessions[i].new_image(100000 * (i + 1) + p)[0] == 0:
                return
        raise RuntimeError("no fitting picture")
    in_threads(pictures, n)
    print(f"[exp] {n} sessions in {time.time() - t0:.1f} s", flush=True)
    drive = zkcnn_amd.MODE_DRIVE_ONLY | zkcnn_amd.MODE_REUSE_GENS

    def warm(i):
        r, _ = sessions[i].prove(seed=1, mode=zkcnn_amd.MODE_REUSE_GENS, want_transcript=False)
        assert r.accepted == 1
        sessions[i].prove(seed=2, mode=drive, want_transcript=False)
    in_threads(warm, n)
    r, _ = sessions[0].prove(seed=3, mode=drive, want_transcript=False)
    print(f"[exp] single stream: {1e3 * (r.prove_s + r.poly_prove_s):.2f} ms per proof", flush=True)
    if args.singles:
        torch.cuda.synchronize()
        t = time.perf_counter()
        in_threads(lambda i: [sessions[i].prove(seed=10 + k, mode=drive, want_transcript=False) for k in range(args.steps)], args.singles)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        print(f"[exp] {args.singles} independent sessions: {args.singles * args.steps / dt:.1f} proofs/s", flush=True)
    for b, k in configs:
        batches = [zkcnn_amd.BatchSession(sessions[j * k:(j + 1) * k]) for j in range(b)]
        try:
            in_threads(lambda j: batches[j].prove(seeds=list(range(k)), mode=drive, want_transcript=False), b)      # warm
            s0 = batches[0].stats()
            torch.cuda.synchronize()
            t = time.perf_counter()
            walls = [[] for _ in range(b)]

            def run(j):
                for s in range(args.steps):
                    batches[j].prove(seeds=[100 * s + i for i in range(k)], mode=drive, want_transcript=False)
                    walls[j].append(batches[j].wall_s)
            in_threads(run, b)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t
            s1 = batches[0].stats()
            fused = (s1["fused_launches"] - s0["fused_launches"]) / args.steps
            print(f"[exp] {b} batches x {k} lanes: {b * k * args.steps / dt:.1f} proofs/s; one batch proof {1e3 * sum(walls[0]) / len(walls[0]):.1f} ms "
                  f"({1e3 * sum(walls[0]) / len(walls[0]) / k:.2f} ms per proof); {fused:.0f} fused launches per batch proof, "
                  f"{(s1['driver_passes'] - s0['driver_passes']) / args.steps:.0f} driver passes", flush=True)
        finally:
            for x in batches:
                x.close()
    for s in sessions:
        s.close()


if __name__ == "__main__":
    main()
