"""GPU experiment: where does the time of the commitment MSM go? (not a test)"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkcnn_amd
from zkcnn_amd import to_mont, R_MOD
from tests import oracle_ffi
o = oracle_ffi.load()
h = zkcnn_amd.HipContext(0)
cols = 4096
bases = o.generators(cols, 1)
rng = np.random.default_rng(1)
def run(name, rows, gen):
    sc = np.concatenate([gen(r) for r in range(rows)])
    h.commit_rows(sc, bases, rows, cols)
    h.profile(0xFFFFFFFF); h.profile_report()
    t = time.perf_counter(); h.commit_rows(sc, bases, rows, cols); dt = time.perf_counter() - t
    rep = h.profile_report(); h.profile(0)
    print(name, rows, f"{dt*1e3:.2f} ms", {k: (round(v['ms'], 2), v['launches']) for k, v in rep.items() if v['launches']})
byte = lambda r: to_mont([int(x) % R_MOD for x in rng.integers(-255, 256, cols)])
wide = lambda r: to_mont([int(x) % R_MOD for x in rng.integers(-(1 << 19), 1 << 19, cols)])
run("byte", 256, byte)
run("wide24", 24, wide)
run("wide24+byte", 256, lambda r: wide(r) if r < 24 else byte(r))
