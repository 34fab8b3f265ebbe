"""Soundness end to end, CPU side: a witness with ONE wrong value must not verify. The oracle session's witness is corrupted after it was
built (test hook zkcnn_session_poke / oracle_session_poke) -- a non-zero constraint row of a RELU layer, a changed activation, a changed
output -- and the full verifier must reject every such proof."""
import numpy as np
import pytest

from tests import oracle_ffi

RELU, MAX_POOL, NCONV = 4, 7, 12


def _mont(oracle, x):
    return [int(v) for v in oracle.from_canonical(np.array([[x, 0, 0, 0]], dtype=np.uint64))[0]]


def _layers(sess):
    out, i = [], 0
    while True:
        size, ty = sess.layer_size(i)
        if size < 0:
            return out
        out.append((i, size, ty))
        i += 1


@pytest.mark.parametrize("model,pic,pp", [("custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5", (8, 8, 2), 1), ("lenet", (32, 32, 1), 1)])
def test_one_wrong_value_is_rejected(oracle, model, pic, pp):
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        assert o.prove(seed=5)[0].accepted == 1
        layers = _layers(o)
    one, seven = _mont(oracle, 1), _mont(oracle, 7)
    relu = [l for l in layers if l[2] in (RELU, MAX_POOL) and l[0] + 1 < len(layers)][0]
    cases = [(relu[0], relu[1] - 1, one),                 # last constraint row of a RELU / pooling layer: must be zero
             (relu[0], 0, seven),                         # an activation
             (layers[-1][0], 0, seven)]                   # an output of the network
    for layer, index, value in cases:
        with oracle_ffi.OracleSession(model, pic, pp) as o:
            o.poke(layer, index, value)
            res, _ = o.prove(seed=5)
            assert res.accepted == 0, f"layer {layer} entry {index} changed but the proof verifies"
