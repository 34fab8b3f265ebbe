#!/usr/bin/env python3
"""How busy is the GPU while lock-step batches run?  From a rocprofv3 --kernel-trace database: between the 30th and the 70th percentile of the fused launches (k_run_b),
the share of time with 0 / 1 / 2 / ... kernels running, per-queue busy shares, and per kernel the summed duration and the time it ran ALONE.
usage: trace_concurrency.py <db or dir>"""
import glob
import os
import sqlite3
import re
import sys
from collections import defaultdict


def short(name):
    ks = [k for k in re.findall(r"k_[A-Za-z0-9_]+(?:<[0-9a-z, ]*>)?", name) if not k.startswith(("k_run_b", "k_run_p", "k_run<", "k_run"))]
    wrap = "b:" if "k_run_b" in name else ("p:" if "k_run_p" in name else "")
    return wrap + (ks[0] if ks else name.split("(")[0][:50])

path = sys.argv[1]
db = path if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
con = sqlite3.connect(db)
cols = [c[1] for c in con.execute("pragma table_info(kernels)")]
print("columns:", cols)
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
scol = "stream_id" if "stream_id" in cols else None
sel = "name, start, end" + (f", {qcol}" if qcol else ", 0") + (f", {scol}" if scol else ", 0") + ", grid_x, grid_y, grid_z, workgroup_x"
rows = list(con.execute(f"select {sel} from kernels order by start"))
fused = [r for r in rows if "k_run_b" in r[0]]
starts = sorted(r[1] for r in fused)           # (most fused launches belong to the timed steps: the window is the middle of them BY COUNT)
w0, w1 = starts[int(0.3 * len(starts))], starts[int(0.7 * len(starts))]
inside = [r for r in rows if r[2] > w0 and r[1] < w1]
ev = []
for i, r in enumerate(inside):
    ev.append((max(r[1], w0), 1, i))
    ev.append((min(r[2], w1), -1, i))
ev.sort()
hist = defaultdict(float)
alone = defaultdict(float)
running = set()
prev = w0
for t, d, i in ev:
    n = len(running)
    hist[min(n, 8)] += t - prev
    if n == 1:
        alone[short(inside[next(iter(running))][0])] += t - prev
    prev = t
    if d == 1:
        running.add(i)
    else:
        running.discard(i)
span = w1 - w0
print(f"window {span / 1e6:.1f} ms, {len(inside)} dispatches")
print("kernels running at once -> share of time:", {k: round(v / span, 3) for k, v in sorted(hist.items())})
byq = defaultdict(float)
for r in inside:
    byq[(r[3], r[4])] += min(r[2], w1) - max(r[1], w0)
print("busy share per (queue, stream):", {str(k): round(v / span, 3) for k, v in sorted(byq.items(), key=lambda kv: str(kv[0]))})
tot = defaultdict(lambda: [0, 0.0, 0.0])
for r in inside:
    nm = short(r[0])
    wg = max(r[8], 1)
    blocks = (r[5] // wg) * r[6] * r[7]
    tot[nm][0] += 1
    tot[nm][1] += r[2] - r[1]
    tot[nm][2] += blocks
allsum = sum(v[1] for v in tot.values())
print(f"sum of durations / window = {allsum / span:.2f}")
print("| kernel | calls | sum ms | share of summed durations | avg us | avg workgroups | ran alone ms |")
for nm, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:32]:
    al = alone.get(nm, 0.0)
    print(f"| {nm} | {v[0]} | {v[1] / 1e6:.1f} | {v[1] / allsum:.3f} | {v[1] / v[0] / 1e3:.1f} | {v[2] / v[0]:.0f} | {al / 1e6:.1f} |")

# ---- one batch's stream: the gaps between its consecutive dispatches (what the stream waits for: its host thread, or the other stream of its hardware queue)
streams = defaultdict(list)
for r in inside:
    if "k_run_b" in r[0] or "k_run_p" in r[0]:
        streams[r[4]].append(r)
sid = max(streams, key=lambda k: len(streams[k]))
seq = sorted([r for r in inside if r[4] == sid], key=lambda r: r[1])
others_same_queue = sorted([r for r in inside if r[3] == seq[0][3] and r[4] != sid], key=lambda r: r[1])
edges = [0, 10, 20, 40, 80, 160, 320, 640, 1e9]
hist_n, hist_t = [0] * (len(edges) - 1), [0.0] * (len(edges) - 1)
before = defaultdict(lambda: [0, 0.0, 0.0])
import bisect
o_starts = [r[1] for r in others_same_queue]
tot_gap = tot_mate = 0.0
for a, b in zip(seq, seq[1:]):
    gap = max(0.0, (b[1] - a[2]) / 1e3)
    # how much of the gap was the queue-mate's kernels running?
    mate = 0.0
    i = bisect.bisect_left(o_starts, a[2] - 50_000_000)
    while i < len(others_same_queue) and others_same_queue[i][1] < b[1]:
        o = others_same_queue[i]
        mate += max(0.0, min(o[2], b[1]) - max(o[1], a[2])) / 1e3
        i += 1
    k = next(j for j in range(len(edges) - 1) if gap < edges[j + 1])
    hist_n[k] += 1
    hist_t[k] += gap
    nm = short(b[0])
    before[nm][0] += 1
    before[nm][1] += gap
    before[nm][2] += mate
    tot_gap += gap
    tot_mate += mate
busy = sum((r[2] - r[1]) / 1e3 for r in seq)
print(f"\nstream {sid} (queue {seq[0][3]}): {len(seq)} dispatches, kernels {busy / 1e3:.1f} ms, gaps {tot_gap / 1e3:.1f} ms of which the queue-mate's kernels ran {tot_mate / 1e3:.1f} ms")
print("gap us  -> count, total ms:", {f"<{int(edges[j + 1])}": (hist_n[j], round(hist_t[j] / 1e3, 1)) for j in range(len(edges) - 1)})
print("| next kernel | launches | avg gap before it us | of which queue-mate busy us | total gap ms |")
for nm, v in sorted(before.items(), key=lambda kv: -kv[1][1])[:16]:
    print(f"| {nm} | {v[0]} | {v[1] / v[0]:.1f} | {v[2] / v[0]:.1f} | {v[1] / 1e3:.1f} |")

print("\nlong gaps of that stream (> 640 us): previous kernel -> next kernel, gap ms")
shown = 0
for a, b in zip(seq, seq[1:]):
    gap = (b[1] - a[2]) / 1e6
    if gap > 0.64 and shown < 48:
        print(f"  {short(a[0])[:28]:28s} -> {short(b[0])[:34]:34s} {gap:8.2f} ms   at +{(a[2] - w0) / 1e6:8.1f} ms")
        shown += 1
