"""Isolated launches of the dominant sumcheck kernel (k_round_quad: fold + round polynomial) on device-resident
tables of 2^log_n elements, for rocprofv3 (kernel trace / PMC) and for the ALU / HBM ceilings next to it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkcnn_amd
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
h = zkcnn_amd.HipContext(0)
sec, nbytes = h.bench_round_quadratic(log_n, iters)
mul = h.bench_fr_mul(1 << 20, 256, 3)
cp = h.bench_copy(1 << 30, 10)
print(f"round_quad 2^{log_n}: {sec*1e3:.4f} ms/launch, algorithmic {nbytes/1e6:.1f} MB -> {nbytes/sec/1e9:.0f} GB/s; "
      f"fr_mul ceiling {(1<<20)*256/mul/1e9:.1f} G/s; copy {2*(1<<30)/cp/1e9:.0f} GB/s")
