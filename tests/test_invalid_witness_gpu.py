"""An INVALID witness on the GPU: one value of a resident layer overwritten (test hook zk_poke_layer_value). The verifier must reject,
and the seeded transcript must still equal the CPU oracle's proof of the same corrupted witness -- which is what pins the round kernel's
live-prefix logic: the zero tails it skips are MEASURED, so a non-zero constraint row simply moves the bound; nothing is assumed
about the witness being valid. Includes the table in front of a factored convolution, whose M ends long before its V does."""
import numpy as np
import pytest

import zkcnn_amd
from tests import oracle_ffi

pytestmark = pytest.mark.gpu
RELU, MAX_POOL = 4, 7
REUSE = zkcnn_amd.MODE_REUSE_GENS

MODELS = [
    ("custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5", (8, 8, 2), 1),                 # factored convolutions behind a RELU and behind a max pooling
    ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),            # tables above 2^16 entries: the large-table round kernel
    ("lenet", (32, 32, 1), 1),                                                # FFT convolutions
]


def _layers(sess):
    out, i = [], 0
    while True:
        size, ty = sess.layer_size(i)
        if size < 0:
            return out
        out.append((i, size, ty))
        i += 1


@pytest.mark.parametrize("model,pic,pp", MODELS)
def test_corrupted_witness_rejected_and_identical_to_oracle(built, oracle, model, pic, pp):
    one = [int(v) for v in oracle.from_canonical(np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]]
    big = [int(v) for v in oracle.random(1, 77)[0]]
    with zkcnn_amd.Session(model, pic, pp) as s:
        layers = _layers(s)
    nonlin = [l for l in layers if l[2] in (RELU, MAX_POOL) and l[0] + 1 < len(layers)]
    # the last constraint row of the first and of the last RELU / pooling layer that something reads, and an activation in the middle
    cases = [(nonlin[0][0], nonlin[0][1] - 1, one), (nonlin[-1][0], nonlin[-1][1] - 1, big), (nonlin[len(nonlin) // 2][0], 3, big)]
    for layer, index, value in cases:
        with oracle_ffi.OracleSession(model, pic, pp) as o:
            o.poke(layer, index, value)
            ores, want = o.prove(seed=0x5EED0041, mode=REUSE)
            assert ores.accepted == 0
        with zkcnn_amd.Session(model, pic, pp) as s:
            s.poke(layer, index, value)
            res, got = s.prove(seed=0x5EED0041, mode=REUSE)
            assert res.accepted == 0, f"layer {layer} entry {index} corrupted but the GPU proof verifies"
            assert got == want, f"layer {layer} entry {index}: GPU transcript of the corrupted witness differs from the oracle's"
