"""zkcnn_session_new_image on full vgg11: wall-clock per call (first call uploads the witness program) and the kernel classes behind it."""
import json
import sys
import time

import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zkcnn_amd

s = zkcnn_amd.Session("vgg11", (32, 32, 3), 1, data_seed=4242, picture_seed=1)
for p in range(2, 10):
    rc, ms = s.new_image(p)
    print("picture", p, "rc", rc, "ms %.2f" % ms, flush=True)
s.profile("all")
rc, ms = s.new_image(2)
print("profiled call: rc", rc, "ms %.2f" % ms)
print({k: v for k, v in s.profile_report().items() if v["launches"]})
s.profile(None)
t0 = time.time()
res, tr = s.prove(seed=1, mode=zkcnn_amd.MODE_REUSE_GENS)
print("prove after new_image: accepted", res.accepted, "prove_s %.4f" % (res.prove_s + res.poly_prove_s))
s.close()
