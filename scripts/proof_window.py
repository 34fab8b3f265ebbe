#!/usr/bin/env python3
"""Kernel census of the LAST proof in a rocprofv3 --kernel-trace database: per kernel launches / busy time, and how much of the proof's
span no kernel was running (host turn-arounds, launch latency). A proof starts with the commitment's k_scalar_codes launch.
usage: proof_window.py <db or dir> [top]"""
import collections
import glob
import os
import sqlite3
import sys

path = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
db = path if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
rows = list(sqlite3.connect(db).execute("select name, start, end from kernels order by start"))
starts = [i for i, r in enumerate(rows) if r[0].startswith("k_scalar_codes(") or r[0].startswith("k_scalar_codes ") or r[0] == "k_scalar_codes"]
if not starts:
    starts = [i for i, r in enumerate(rows) if "k_scalar_codes" in r[0] and "wide" not in r[0]]
win = rows[starts[-1]:]
span = (win[-1][2] - win[0][1]) / 1e3
busy = 0.0
cur_end = win[0][1]
per = collections.defaultdict(lambda: [0, 0.0])
for name, st, en in win:
    k = name.split("(")[0].replace("void ", "")[:48]
    per[k][0] += 1
    per[k][1] += (en - st) / 1e3
    if en > cur_end:
        busy += (en - max(st, cur_end)) / 1e3
        cur_end = en
print(f"last proof: {len(win)} launches, span {span / 1e3:.2f} ms, some kernel running {busy / 1e3:.2f} ms, idle {(span - busy) / 1e3:.2f} ms")
print("| kernel | launches | total us | avg us |\n|---|---|---|---|")
for k, (n, us) in sorted(per.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"| {k} | {n} | {us:.0f} | {us / n:.1f} |")
