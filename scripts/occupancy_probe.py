"""Fr-multiply throughput against the number of resident waves per SIMD (dependent product chains per thread)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkcnn_amd
hc = zkcnn_amd.HipContext(0)
simds = 256 * 4
for waves in (1, 2, 3, 4, 6, 8):
    n = simds * waves * 64
    sec = hc.bench_fr_mul(n, 512, 3)
    print(f"{waves} waves/SIMD: {n * 512 / sec / 1e9:7.1f} G Fr-mul/s")
hc.close()
