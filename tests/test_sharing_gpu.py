"""One resident circuit and one set of generator tables per GPU, shared by the sessions that use them (round 3; reference src/prover.hpp:47-48
keeps one layeredCircuit per prover -- here eight provers of one model used to hold eight sorted copies of its 1.2e8 gates and eight 3.2 GB byte
tables of the same public generators).

Sessions of the same model under the same quantisation scales upload the SAME circuit: the first builds the device-side lists, the others attach
(zk_sharing_stats). Each keeps its own values, tables, stream and scratch, so they prove different pictures, concurrently, and every transcript
still equals the CPU oracle's for that picture; closing the session that built the circuit leaves the others working (reference counts)."""
import threading

import pytest

import zkcnn_amd
from tests import oracle_ffi

pytestmark = pytest.mark.gpu
REUSE, DRIVE = zkcnn_amd.MODE_REUSE_GENS, zkcnn_amd.MODE_DRIVE_ONLY

MODELS = [
    ("custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5", (8, 8, 2), 1),
    ("lenet", (32, 32, 1), 1),
    ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),
]


def _fitting_picture(sess, first_seed):
    for seed in range(first_seed, first_seed + 64):
        if sess.new_image(seed)[0] == 0:
            return seed
    pytest.skip("no synthetic picture with the circuit's quantisation scales among 64")


@pytest.mark.parametrize("model,pic,pp", MODELS)
def test_sessions_share_one_resident_circuit(built, model, pic, pp):
    before = zkcnn_amd.sharing_stats()
    a = zkcnn_amd.Session(model, pic, pp)
    stmt = a.statement()
    mid = zkcnn_amd.sharing_stats()
    assert mid["circuit_builds"] == before["circuit_builds"] + 1
    b = zkcnn_amd.Session(model, pic, pp, calibrated=stmt)
    c = zkcnn_amd.Session(model, pic, pp, calibrated=stmt)
    after = zkcnn_amd.sharing_stats()
    assert after["circuit_builds"] == mid["circuit_builds"], "a second session of the same circuit sorted and uploaded it again"
    assert after["circuit_attaches"] == mid["circuit_attaches"] + 2
    try:
        seed_b, seed_c = _fitting_picture(b, 1000), _fitting_picture(c, 5000)
        want = {}
        for name, ps in (("a", 0), ("b", seed_b), ("c", seed_c)):
            with oracle_ffi.OracleSession(model, pic, pp, picture_seed=ps, calibrated=stmt) as o:
                res, tr = o.prove(seed=0x5EED0061, mode=REUSE)
                assert res.accepted == 1
                want[name] = tr
        assert want["a"] != want["b"] != want["c"]
        # the three sessions prove side by side, twice (the second use of the public generators builds their byte table: once for all three)
        got, errs = {}, []

        def run(name, s):
            try:
                for k in range(2):
                    res, tr = s.prove(seed=0x5EED0061, mode=REUSE)
                    assert res.accepted == 1, res.message.decode()
                    got[name] = tr
            except BaseException as e:      # noqa: BLE001
                errs.append(e)
        th = [threading.Thread(target=run, args=(n, s)) for n, s in (("a", a), ("b", b), ("c", c))]
        [t.start() for t in th]
        [t.join() for t in th]
        assert not errs, errs
        for name in "abc":
            assert got[name] == want[name], f"session {name}: transcript on the shared circuit differs from the oracle's for its picture"
        tables = zkcnn_amd.sharing_stats()
        assert tables["byte_table_builds"] - before["byte_table_builds"] <= 1, "the byte table of one generator set was built more than once"
        # the session that built the circuit goes away: the others keep their reference
        a.close()
        a = None
        res, tr = b.prove(seed=0x5EED0061, mode=REUSE | DRIVE)
        assert tr == want["b"]
        res, tr = c.prove(seed=0x5EED0061, mode=REUSE)
        assert res.accepted == 1 and tr == want["c"]
    finally:
        for s in (a, b, c):
            if s is not None:
                s.close()
    # everything released: the next session of the model builds the circuit again
    again = zkcnn_amd.sharing_stats()
    with zkcnn_amd.Session(model, pic, pp) as d:
        assert zkcnn_amd.sharing_stats()["circuit_builds"] == again["circuit_builds"] + 1
        assert d.prove(seed=0x5EED0061, mode=REUSE)[1] == want["a"]
