"""The next picture on a resident circuit (zkcnn_session_new_image -> zk_witness_program_upload / zk_witness_rerun):
picture quantised on the host, every layer value and auxiliary witness recomputed in HBM by replaying the recorded witness program.
The session must then produce, byte for byte, the transcript of a session created for that picture from scratch -- on the GPU and in
the CPU oracle."""
import pytest

import zkcnn_amd
from tests import oracle_ffi

pytestmark = pytest.mark.gpu

REUSE, FS = zkcnn_amd.MODE_REUSE_GENS, zkcnn_amd.MODE_FIAT_SHAMIR
W = 4242
SEED = 0x5EED0021

MODELS = [
    ("lenet", (32, 32, 1), 1),
    ("lenet.avg", (32, 32, 1), 2),
    ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2),
    ("custom:C3:3:1:n M F6 F4", (8, 8, 1), 1),
    ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),          # vgg11 at quarter width
]


def _fits_and_refused(model, pic, pp, scales, pictures):
    """pictures whose values fit `scales` (with the transcript a calibrated oracle session gives for them) / those that do not"""
    want, refused = {}, []
    for p in pictures:
        try:
            with oracle_ffi.OracleSession(model, pic, pp, data_seed=W, picture_seed=p, calibrated=scales) as o:
                res, tr = o.prove(seed=SEED, mode=REUSE)
                assert res.accepted == 1 and o.statement() == scales
                want[p] = tr
        except RuntimeError:
            refused.append(p)
    return want, refused


@pytest.mark.parametrize("model,pic,pp", MODELS)
def test_new_image_transcripts_identical_to_fresh_sessions(built, model, pic, pp):
    with zkcnn_amd.Session(model, pic, pp, data_seed=W, picture_seed=1) as s:
        scales = s.statement()
        res, first = s.prove(seed=SEED, mode=REUSE)
        assert res.accepted == 1
        with oracle_ffi.OracleSession(model, pic, pp, data_seed=W, picture_seed=1) as o:
            assert o.statement() == scales and o.prove(seed=SEED, mode=REUSE)[1] == first
        want, refused = _fits_and_refused(model, pic, pp, scales, range(2, 8))
        assert want, "none of six pictures fits the scales of picture 1"
        last = first
        for p in range(2, 8):
            rc, ms = s.new_image(p)
            if p in want:
                assert rc == 0, f"picture {p} fits the session's scales but new_image returned {rc}"
                res, tr = s.prove(seed=SEED, mode=REUSE)
                assert res.accepted == 1, res.message.decode()
                assert tr == want[p], f"picture {p}: transcript after new_image differs from a calibrated session built for it"
                last = tr
            else:
                # refused, and the session keeps proving the picture it proved before (code 2: the values in HBM were overwritten on the
                # way and the previous picture was replayed)
                assert rc in (1, 2), f"picture {p} does not fit the session's scales but new_image returned {rc}"
                assert s.prove(seed=SEED, mode=REUSE)[1] == last
            assert s.statement() == scales
        # the same picture handed over as pixel values
        p0 = sorted(want)[0]
        rc, _ = s.new_image(pixels=s.synthetic_picture(p0))
        assert rc == 0 and s.prove(seed=SEED, mode=REUSE)[1] == want[p0]
        # a Fiat-Shamir proof of the new picture verifies off line
        res, proof = s.prove(mode=FS)
        assert res.accepted == 1 and s.verify(proof, mode=FS).accepted == 1
    # a calibrated GPU session built for the picture from scratch gives the same bytes
    with zkcnn_amd.Session(model, pic, pp, data_seed=W, picture_seed=p0, calibrated=scales) as fresh:
        assert fresh.statement() == scales and fresh.prove(seed=SEED, mode=REUSE)[1] == want[p0]
    if refused:
        with pytest.raises(RuntimeError):
            zkcnn_amd.Session(model, pic, pp, data_seed=W, picture_seed=refused[0], calibrated=scales)


def _stmt(model, pic, pp, p):
    with oracle_ffi.OracleSession(model, pic, pp, data_seed=W, picture_seed=p) as o:
        return tuple(o.statement())


def test_new_image_full_vgg11(built):
    """BASELINE configs[1] at full size: the program covers 8 FFT convolutions with 2^12-point transforms, 5 max poolings, 3 fc layers"""
    model, pic, pp = "vgg11", (32, 32, 3), 1
    with zkcnn_amd.Session(model, pic, pp, data_seed=W, picture_seed=1) as s:
        assert s.prove(seed=SEED, mode=REUSE).__getitem__(0).accepted == 1
        done = None
        times = []
        for p in range(2, 10):
            rc, ms = s.new_image(p)
            times.append(ms)
            if rc == 0:
                done = p
                break
        assert done is not None, "none of eight pictures kept the quantisation scales"
        res, tr = s.prove(seed=SEED, mode=REUSE)
        assert res.accepted == 1, res.message.decode()
        res, proof = s.prove(mode=FS)
        assert res.accepted == 1 and s.verify(proof, mode=FS).accepted == 1
        stmt = s.statement()
    with zkcnn_amd.Session(model, pic, pp, data_seed=W, picture_seed=done, calibrated=stmt) as fresh:
        assert fresh.statement() == stmt
        assert fresh.prove(seed=SEED, mode=REUSE)[1] == tr
    print("new_image ms:", times)


@pytest.mark.parametrize("model,pic,pp", [("custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5", (8, 8, 2), 3),
                                          ("custom:C4:3:0:s C4:5:2:s A F3", (18, 18, 1), 2), MODELS[4]])
def test_convolutions_in_integers_equal_field_arithmetic(built, model, pic, pp):
    """new_image evaluates a direct convolution from its two tensors in 64-bit integers when a device-side bound test allows it
    (witness_kernels.cuh: k_conv_eval_i64), in field arithmetic otherwise: same layer values, hence same transcripts, either way -- also for a
    weight that is no small integer, which sends exactly that convolution back to the field."""
    DRIVE = zkcnn_amd.MODE_DRIVE_ONLY
    with zkcnn_amd.Session(model, pic, pp, data_seed=W, picture_seed=1) as s:
        p = next(q for q in range(2, 34) if s.new_image(q)[0] == 0)
        n_conv, n_int, wstart = s.conv_paths()
        assert n_conv > 0 and n_int == n_conv, "quantised tensors did not take the integer path"
        res, tr_int = s.prove(seed=SEED, mode=REUSE)
        assert res.accepted == 1
        s.conv_paths(force_field=1)
        assert s.new_image(p)[0] == 0 and s.conv_paths()[1] == 0
        assert s.prove(seed=SEED, mode=REUSE | DRIVE)[1] == tr_int
        # an invalid witness on purpose: one weight becomes a field element of ~255 bits. Whatever becomes of the picture (its values no longer
        # fit anything), the device-side test must have sent exactly that convolution back to field arithmetic
        s.conv_paths(force_field=0)
        assert s.new_image(p)[0] == 0 and s.conv_paths()[:2] == (n_conv, n_conv)
        s.poke(0, wstart, (1, 2, 3, 4))
        try:
            s.new_image(p)
        except RuntimeError:
            pass
        assert s.conv_paths()[:2] == (n_conv, n_conv - 1), "the convolution with a wide weight must fall back to field arithmetic"
