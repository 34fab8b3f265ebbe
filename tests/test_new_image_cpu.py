"""The witness as a program, CPU side: while it emits the circuit, the generator records how layer 0's auxiliary
witnesses (bits, signs, maxima, bits of window sums) follow from the layer values and where an activation range fixes a quantisation
scale (host/neuralNetwork.hpp: witnessProgram). Here the oracle INTERPRETS that record with plain loops for another picture and every
layer value must equal a session built for that picture from scratch -- the same comparison tests/test_new_image_gpu.py makes for the
HIP replay (zk_witness_rerun), without a GPU."""
import ctypes

import pytest

from tests import oracle_ffi

MODELS = [
    ("lenet", (32, 32, 1), 1),                                            # max pooling fused with the ReLU, FFT convolutions
    ("lenet.avg", (32, 32, 1), 2),                                        # average pooling: bits of window sums
    ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2),      # FFT conv + max pool + naive conv + avg pool + fc
    ("custom:C3:3:1:n M F6 F4", (8, 8, 1), 1),                            # two-layer naive convolution
]


def _replay(o, picture_seed):
    fn = o.lib.oracle_session_replay_program
    fn.restype = ctypes.c_int64
    return fn(ctypes.c_void_p(o.h), ctypes.c_uint64(picture_seed))


@pytest.mark.parametrize("model,pic,pp", MODELS)
def test_recorded_program_reproduces_a_fresh_build(oracle, model, pic, pp):
    """a session built for picture 1 fixes the circuit's quantisation scales; every other picture either fits them -- then the replayed
    program must give, value for value, what a CALIBRATED build of that picture gives (same scales, from scratch) -- or does not, and
    both sides must say so"""
    fits = refused = 0
    with oracle_ffi.OracleSession(model, pic, pp, data_seed=4242, picture_seed=1) as o:
        scales = o.statement()
        for p in range(2, 9):
            r = _replay(o, p)
            assert r in (0, -2), f"picture {p}: {r} layer values of the replayed program differ from a calibrated build"
            try:
                with oracle_ffi.OracleSession(model, pic, pp, data_seed=4242, picture_seed=p, calibrated=scales) as c:
                    assert r == 0 and c.statement() == scales and c.prove(seed=3)[0].accepted == 1
                fits += 1
            except RuntimeError:
                assert r == -2, f"picture {p} does not fit the scales but the replay did not notice"
                refused += 1
    assert fits >= 1, "none of seven pictures fits the scales of picture 1: nothing was compared"
    print(f"{model}: {fits} pictures fit, {refused} refused")


def test_picture_seed_changes_the_picture_only(oracle):
    model, pic, pp = "custom:C2:3:1:f M F4", (8, 8, 1), 1
    with oracle_ffi.OracleSession(model, pic, pp, data_seed=7, picture_seed=1) as a, \
            oracle_ffi.OracleSession(model, pic, pp, data_seed=7, picture_seed=2) as b, \
            oracle_ffi.OracleSession(model, pic, pp, data_seed=7, picture_seed=1) as a2:
        ra, ta = a.prove(seed=11)
        rb, tb = b.prove(seed=11)
        assert ra.accepted == 1 and rb.accepted == 1
        assert ta != tb                                   # another picture, another proof
        assert a2.prove(seed=11)[1] == ta                 # deterministic
