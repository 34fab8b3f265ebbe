"""Lock-step batches on the GPU (include/zkcnn_hip.h: zk_batch_*; include/zkcnn_api.h: zkcnn_batch_*): K sessions of one model become the lanes
of a batch -- one host thread drives their K verifier loops (reference src/verifier.cpp:149-264, unchanged), the sumcheck rounds of the K proofs
(reference src/prover.cpp:360-426) are ONE kernel launch each -- and every lane's canonical transcript must be byte for byte the CPU oracle's
for that lane's picture and challenge seed: K = 2, 8 and a ragged 3, direct and FFT convolutions, every protocol mode."""
import os

import numpy as np
import pytest

import zkcnn_amd
from tests import oracle_ffi

pytestmark = pytest.mark.gpu
LANE_RESIDENT = os.environ.get("ZKCNN_LANE_RESIDENT", "0") == "1"          # (the library's default: sumcheck.hip policy::LANE_RESIDENT_TAIL = false)
M = zkcnn_amd
REUSE, DRIVE = M.MODE_REUSE_GENS, M.MODE_DRIVE_ONLY

QUARTER_VGG11 = "vgg:16 M 32 M 64 64 M 128 128 M 128 128 M"
CASES = [
    ("custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5", (8, 8, 2), 1, 2),      # direct convolutions (factored gate sums), max pooling
    ("custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5", (8, 8, 2), 1, 8),
    ("lenet", (32, 32, 1), 1, 3),                                      # FFT convolutions of one picture, average pooling; ragged batch
    ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2, 3),  # FFT convolutions over two pictures per circuit: cubic rounds, factored DOT_PROD table
    (QUARTER_VGG11, (32, 32, 3), 1, 3),                                # vgg11 at quarter width: tables up to 2^21 entries (the large-table round kernel)
    (QUARTER_VGG11, (32, 32, 3), 1, 8),
]


def _lanes(model, pic, pp, k, clones=False):
    """k sessions on ONE resident circuit (the first one's quantisation scales), each with its own picture; clones: sessions 1.. are clones of
    the first (zkcnn_session_clone: no host circuit of their own) instead of sessions built from the data under the first one's scales"""
    first = M.Session(model, pic, pp)
    stmt = first.statement()
    ss, seeds = [first], [0]
    try:
        for i in range(1, k):
            s = first.clone() if clones else M.Session(model, pic, pp, calibrated=stmt)
            ss.append(s)
            for ps in range(1000 * i, 1000 * i + 64):
                if s.new_image(ps)[0] == 0:
                    seeds.append(ps)
                    break
            else:
                pytest.skip("no synthetic picture with the circuit's quantisation scales among 64")
    except BaseException:
        for s in ss:
            s.close()
        raise
    return ss, seeds, stmt


def _oracle(model, pic, pp, stmt, picture_seed, seed, mode):
    with oracle_ffi.OracleSession(model, pic, pp, picture_seed=picture_seed, calibrated=stmt) as o:
        res, tr = o.prove(seed=seed, mode=mode)
    return res, tr


@pytest.mark.parametrize("model,pic,pp,k", CASES)
def test_every_lane_of_a_batch_equals_the_oracle(built, model, pic, pp, k):
    ss, pics, stmt = _lanes(model, pic, pp, k, clones=(k == 8))          # (the batches of eight are made the way bench.py makes them: one session + clones)
    try:
        seeds = [0x5EED0B00 + 3 * i for i in range(k)]
        want = [_oracle(model, pic, pp, stmt, pics[i], seeds[i], REUSE) for i in range(k)]
        assert len({t for _, t in want}) == k
        with M.BatchSession(ss) as B:
            for rep in range(3):                 # (the second use of the public generators builds their byte table, the third is the path bench.py times)
                if rep == 2:
                    before = B.stats()
                got = B.prove(seeds=seeds, mode=REUSE)
                for i in range(k):
                    assert got[i][0].accepted == 1, f"lane {i}: {got[i][0].message.decode()}"
                    assert got[i][1] == want[i][1], f"lane {i} (proof {rep}): transcript differs from the CPU oracle's for its picture and seed"
                    assert abs(got[i][0].proof_kb - want[i][0].proof_kb) < 1e-9
            st = B.stats()
            rounds = got[0][0].n_rounds
            fused, lane = st["fused_launches"] - before["fused_launches"], st["lane_launches"] - before["lane_launches"]
            # once the generator tables exist the lanes are in lock step from the first call to the last: EVERY deferred launch of the proof was
            # fused over all k lanes, and a round is one launch
            assert lane == k * fused and fused > 0, st
            assert fused <= 2.5 * rounds, f"{fused} fused launches for {rounds} rounds"
            # drive-only (the verifier's checks skipped) makes the same calls: same bytes
            drv = B.prove(seeds=seeds, mode=REUSE | DRIVE)
            assert all(drv[i][0].accepted == -1 and drv[i][1] == want[i][1] for i in range(k))
        # the sessions are their own again: alone, on their own streams (resident round kernels), the same proofs
        for i in (0, k - 1):
            res, tr = ss[i].prove(seed=seeds[i], mode=REUSE)
            assert res.accepted == 1 and tr == want[i][1]
    finally:
        for s in ss:
            s.close()


@pytest.mark.parametrize("mode", [0, M.MODE_FULL_IPA, M.MODE_FIAT_SHAMIR, M.MODE_ZK, M.MODE_ZK | M.MODE_FIAT_SHAMIR, REUSE | M.MODE_HOST_TAIL, REUSE | M.MODE_GPU_TAIL,
                                  REUSE | M.MODE_CROSS_PRED, M.MODE_FIAT_SHAMIR | M.MODE_FS_DEVICE])
def test_batch_in_every_protocol_mode(built, mode):
    """fresh generators per lane (drawn by each lane's verifier from its own seeded stream), the argument down to length 1, challenges hashed
    from the transcript, masked rounds + blinded commitments (private coins per lane), the hybrid host tail at 64 entries / at the lanes' default 32 (every other mode) / not at
    all (MODE_GPU_TAIL: every round a fused launch), host + GPU predicates side by side,
    and the device-side Fiat-Shamir chain (which a lane does not take: its driver hashes -- same bytes)"""
    model, pic, pp, k = "custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2, 3
    ss, pics, stmt = _lanes(model, pic, pp, k)
    try:
        seeds = [0x5EED0C00 + 5 * i for i in range(k)]
        want = [_oracle(model, pic, pp, stmt, pics[i], seeds[i], mode & ~(M.MODE_HOST_TAIL | M.MODE_GPU_TAIL | M.MODE_CROSS_PRED | M.MODE_FS_DEVICE)) for i in range(k)]
        on_host = ss[0].host_tail_rounds()
        with M.BatchSession(ss) as B:
            got = B.prove(seeds=seeds, mode=mode)
        # a lane's small rounds run on the host unless the mode says otherwise -- or, with the lanes' resident tail (ZKCNN_LANE_RESIDENT=1 / the library's
        # policy), in a resident workgroup of one fused launch per phase: then only the modes that ask for a host tail (explicitly, or the zero-knowledge mode) have one
        want_host_tail = not mode & M.MODE_GPU_TAIL and (not LANE_RESIDENT or bool(mode & (M.MODE_HOST_TAIL | M.MODE_ZK)))
        assert (ss[0].host_tail_rounds() > on_host) == want_host_tail
        for i in range(k):
            assert got[i][0].accepted == 1, f"lane {i}: {got[i][0].message.decode()}"
            assert got[i][1] == want[i][1], f"lane {i}: transcript differs from the CPU oracle's"
    finally:
        for s in ss:
            s.close()


def test_batch_soundness_and_lanes_out_of_step(built):
    """a corrupted message is rejected in every lane (each lane's verifier leaves its loop early; the batch goes on without it); an INVALID
    witness in ONE lane (a poked value) is rejected there, accepted elsewhere, and that lane's transcript still equals the oracle's for the
    same corrupted witness; a new picture in one lane between two batch proofs"""
    model, pic, pp, k = "custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5", (8, 8, 2), 1, 3
    ss, pics, stmt = _lanes(model, pic, pp, k)
    try:
        seeds = [21, 22, 23]
        with M.BatchSession(ss) as B:
            good = B.prove(seeds=seeds, mode=REUSE)
            assert [r.accepted for r, _ in good] == [1] * k
            n_msg = good[0][0].n_messages
            for at in (0, 7, n_msg // 2, n_msg - 1):
                bad = B.prove(seeds=seeds, mode=REUSE | M.MODE_TAMPER | (at << 8))
                assert [r.accepted for r, _ in bad] == [0] * k, f"message {at}"
            assert [t for _, t in B.prove(seeds=seeds, mode=REUSE)] == [t for _, t in good]
            # lane 1 gets an invalid witness: one value of layer 1 changed
            one = [int(v) for v in oracle_ffi.load().from_canonical(np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]]
            ss[1].poke(1, 0, one)
            mixed = B.prove(seeds=seeds, mode=REUSE)
            assert [r.accepted for r, _ in mixed] == [1, 0, 1]
            assert mixed[0][1] == good[0][1] and mixed[2][1] == good[2][1]
            with oracle_ffi.OracleSession(model, pic, pp, picture_seed=pics[1], calibrated=stmt) as o:
                o.poke(1, 0, one)
                ores, otr = o.prove(seed=seeds[1], mode=REUSE)
            assert ores.accepted == 0 and otr == mixed[1][1]
            # another picture in lane 1 (the witness is replayed in HBM on the batch's stream), then the batch again
            for ps in range(7000, 7064):
                if ss[1].new_image(ps)[0] == 0:
                    break
            else:
                pytest.skip("no fitting picture")
            again = B.prove(seeds=seeds, mode=REUSE)
            assert [r.accepted for r, _ in again] == [1] * k
            assert again[1][1] == _oracle(model, pic, pp, stmt, ps, seeds[1], REUSE)[1]
            assert again[0][1] == good[0][1]
    finally:
        for s in ss:
            s.close()


def test_two_batches_side_by_side(built):
    """two batches of one model on two host threads (two streams): the deployment shape of bench.py -- while one batch waits for a round the
    other's kernels run"""
    import threading
    model, pic, pp = QUARTER_VGG11, (32, 32, 3), 1
    ss, pics, stmt = _lanes(model, pic, pp, 4)
    try:
        seeds = [31, 32, 33, 34]
        want = [_oracle(model, pic, pp, stmt, pics[i], seeds[i], REUSE | DRIVE)[1] for i in range(4)]
        with M.BatchSession(ss[:2]) as A, M.BatchSession(ss[2:]) as B:
            got, errs = {}, []

            def run(name, batch, sd):
                try:
                    for _ in range(3):
                        got[name] = batch.prove(seeds=sd, mode=REUSE | DRIVE)
                except BaseException as e:      # noqa: BLE001
                    errs.append(e)
            th = [threading.Thread(target=run, args=("a", A, seeds[:2])), threading.Thread(target=run, args=("b", B, seeds[2:]))]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not errs, errs
            assert [t for _, t in got["a"]] == want[:2] and [t for _, t in got["b"]] == want[2:]
    finally:
        for s in ss:
            s.close()


def test_batches_created_as_a_group_in_the_reference_semantics(built):
    """BatchSession.group (zkcnn_batch_create_group): every batch's stream exists before any lane is attached -- the shape bench.py times since round 6 -- and the
    proofs run in the reference's semantics (a generator set of its own per proof, drawn from the proof's seed; inner-product argument down to length 1): every
    lane's transcript is the oracle's, proof after proof (the second proof of a lane re-uses the pooled table buffers of the first one's dead generator set)."""
    import threading
    model, pic, pp = QUARTER_VGG11, (32, 32, 3), 1
    FULL = M.MODE_FULL_IPA
    ss, pics, stmt = _lanes(model, pic, pp, 4)
    try:
        seeds = [[41, 42, 43, 44], [51, 52, 53, 54]]
        want = [[_oracle(model, pic, pp, stmt, pics[i], sd[i], FULL | DRIVE)[1] for i in range(4)] for sd in seeds]
        A, B = M.BatchSession.group([ss[:2], ss[2:]])
        try:
            got, errs = {}, []

            def run(name, batch, lo):
                try:
                    got[name] = [batch.prove(seeds=sd[lo:lo + 2], mode=FULL | DRIVE) for sd in seeds]
                except BaseException as e:      # noqa: BLE001
                    errs.append(e)
            th = [threading.Thread(target=run, args=("a", A, 0)), threading.Thread(target=run, args=("b", B, 2))]
            [t.start() for t in th]
            [t.join() for t in th]
            assert not errs, errs
            for step in range(2):
                assert [t for _, t in got["a"][step]] == want[step][:2] and [t for _, t in got["b"][step]] == want[step][2:]
            full = A.prove(seeds=[61, 62], mode=FULL)            # ... and with the verifier's checks on
            assert [r.accepted for r, _ in full] == [1, 1]
        finally:
            A.close()
            B.close()
        with pytest.raises(RuntimeError):
            M.BatchSession.group([[ss[0]], [ss[0]]])             # a session cannot be a lane of two batches
        res, _ = ss[0].prove(seed=1, mode=REUSE)                 # (nothing is left attached by the failed attempt)
        assert res.accepted == 1
    finally:
        for s in ss:
            s.close()


@pytest.mark.parametrize("switches", [{"ZKCNN_DIGIT_OPENING": "0"}, {"ZKCNN_DIGIT_PAIRS": "1"}, {"ZKCNN_DIGIT_PAIRS": "32"}, {"ZKCNN_DIGIT_AFFINE_PER": "64"}, {"ZKCNN_LOAD_SHAPES": "0"},
                                      {"ZKCNN_HOST_GENERATORS": "1"}])
def test_opening_through_the_digit_table_and_through_window_tables_agree(built, monkeypatch, switches):
    """Lanes of a batch open a fresh generator set through the digit table of its commitment (k_bytes_acc by window + k_cl_whorner32: no window tables); a proof on its
    own builds window tables and sums digit planes. Both are the oracle's bytes -- with the route switched off, with one column and with 32 columns per lane (ragged and
    single-chunk grids), with 64 digits per inversion in the digit table's conversion, with a lone proof's launch shapes, with the verifier's generators multiplied on
    host threads instead of the GPU (zk_fixed_base_mul; the oracle always multiplies on the host) -- and the verifier's own multiplications (a set that never had a commitment:
    window tables built on demand) accept."""
    model, pic, pp = QUARTER_VGG11, (32, 32, 3), 1
    FULL = M.MODE_FULL_IPA
    ss, pics, stmt = _lanes(model, pic, pp, 2)
    try:
        want = [_oracle(model, pic, pp, stmt, pics[i], 71 + i, FULL | DRIVE)[1] for i in range(2)]
        lone = [ss[i].prove(seed=71 + i, mode=FULL | DRIVE)[1] for i in range(2)]          # window tables beside the proof, digit planes
        assert lone == want
        for k, v in switches.items():
            monkeypatch.setenv(k, v)
        b = M.BatchSession(ss)
        try:
            assert [t for _, t in b.prove(seeds=[71, 72], mode=FULL | DRIVE)] == want
            assert [r.accepted for r, _ in b.prove(seeds=[73, 74], mode=FULL)] == [1, 1]
        finally:
            b.close()
    finally:
        for s in ss:
            s.close()


def test_attach_refuses_other_circuits(built):
    a = M.Session("custom:F8 F4", (4, 4, 1), 1)
    b = M.Session("custom:F8 F5", (4, 4, 1), 1)
    try:
        with pytest.raises(RuntimeError):
            M.BatchSession([a, b])
        with pytest.raises(RuntimeError):
            M.BatchSession([a, a])
        # nothing is left attached by the failed attempts
        res, _ = a.prove(seed=1)
        assert res.accepted == 1
    finally:
        a.close()
        b.close()


def test_clones_share_the_circuit_and_nothing_else(built):
    """zkcnn_session_clone: a context of its own on the parent's resident circuit, a copy of the witness, no host circuit. The clone proves what the
    parent proves, takes pictures of its own, outlives the parent, verifies serialized proofs (GPU predicates) -- and refuses the modes that need the
    gate lists on the host."""
    model, pic, pp = "custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5", (8, 8, 2), 1
    before = M.sharing_stats()
    a = M.Session(model, pic, pp)
    stmt = a.statement()
    b = a.clone()
    c = b.clone()                                  # a clone of a clone
    try:
        mid = M.sharing_stats()
        assert mid["circuit_builds"] == before["circuit_builds"] + 1 and mid["circuit_attaches"] == before["circuit_attaches"] + 2
        ra, ta = a.prove(seed=5, mode=REUSE)
        rb, tb = b.prove(seed=5, mode=REUSE)
        assert ra.accepted == 1 and rb.accepted == 1 and ta == tb
        assert rb.gate_cnt_bin == ra.gate_cnt_bin and rb.gate_cnt_uni == ra.gate_cnt_uni
        fa, fta = a.prove(mode=M.MODE_FIAT_SHAMIR)
        fb, ftb = b.prove(mode=M.MODE_FIAT_SHAMIR)          # the statement hash needs the wiring digest and the gate counts: the structure copy carries them
        assert fa.accepted == 1 and fb.accepted == 1 and fta == ftb
        assert b.verify(fta, mode=M.MODE_FIAT_SHAMIR).accepted == 1
        for ps in range(3000, 3064):
            if c.new_image(ps)[0] == 0:
                break
        else:
            pytest.skip("no fitting picture")
        a.close()                                  # the parent goes away: the clones keep the circuit and the witness program
        a = None
        rc_, tc = c.prove(seed=6, mode=REUSE)
        assert rc_.accepted == 1 and tc == _oracle(model, pic, pp, stmt, ps, 6, REUSE)[1]
        assert b.prove(seed=5, mode=REUSE)[1] == ta
        for bad_mode in (M.MODE_HOST_PRED, M.MODE_CROSS_PRED):
            r, _ = b.prove(seed=7, mode=REUSE | bad_mode)
            assert r.accepted == 0 and b"structure copy" in r.message
    finally:
        for s in (a, b, c):
            if s is not None:
                s.close()
