"""How many pictures fit a vgg11 circuit: scales of one picture as they are, and with one bit of head-room on every scale (a calibrated session)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import zkcnn_amd

with zkcnn_amd.Session("vgg11", (32, 32, 3), 1, data_seed=4242, picture_seed=1) as s:
    scales = s.statement()
    ok = sum(s.new_image(100 + p)[0] == 0 for p in range(40))
    print("scales of picture 1 as they are:", ok, "of 40 pictures accepted")
head = [max(x - 1, 0) for x in scales]
with zkcnn_amd.Session("vgg11", (32, 32, 3), 1, data_seed=4242, picture_seed=1, calibrated=head) as s:
    ok = sum(s.new_image(100 + p)[0] == 0 for p in range(40))
    res, _ = s.prove(seed=1, mode=zkcnn_amd.MODE_REUSE_GENS)
    print("one bit of head-room on every scale:", ok, "of 40 pictures accepted; proof of the last one accepted:", res.accepted,
          "prover ms %.1f" % (1e3 * (res.prove_s + res.poly_prove_s)))
