#!/usr/bin/env python3
"""Per-dispatch timeline of selected kernels from a rocprofv3 --kernel-trace database (the .db under gpurun_out/<tag>/):
start offset, duration and grid of the last N dispatches whose name matches.  usage: kernel_timeline.py <db or dir> <regex> [N]"""
import glob
import os
import re
import sqlite3
import sys

path, pat = sys.argv[1], re.compile(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
db = path if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*.db"), recursive=True)[0]
rows = [r for r in sqlite3.connect(db).execute("select name, start, end, grid_x, grid_y, grid_z, workgroup_x from kernels order by start") if pat.search(r[0])]
seq = rows[-n:]
t0 = seq[0][1]
prev_end = t0
for name, st, en, gx, gy, gz, wx in seq:
    print(f"{name.split('(')[0][:44]:44s} +{(st - t0) / 1e3:10.1f} us  gap {(st - prev_end) / 1e3:8.1f}  dur {(en - st) / 1e3:8.1f} us  grid {gx // max(wx, 1)}x{gy}x{gz} wg {wx}")
    prev_end = en
