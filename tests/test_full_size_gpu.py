"""Byte parity with the CPU oracle at FULL size: vgg11 / CIFAR shape, pic_cnt = 1 -- BASELINE.json configs[2], the workload bench.py times.

The reduced-width circuits of the other GPU tests reach every kernel, but not the 2^24-entry tables of the real workload: the large-table
round kernel's live-prefix path with its fill / guard-quad switching, the 24 wide rows of the commitment riding as virtual rows of the 4096-column
MSM, the factored sums of 512-channel convolutions. Here the canonical transcripts of the full circuit are compared byte for byte
(reference loop being restated: src/prover.cpp:155-239, 396-426), in the plain interactive mode and in the modes bench.py times, plus one
INVALID witness at full size (a non-zero constraint row behind the first RELU: the measured live prefix moves to the end of a 2^21-entry table).

The oracle needs ~27 s of one host core per full-size proof (+ ~10 s for its gate-walking verifier): the three oracle runs go side by side in
three processes.
"""
import hashlib
import os
import multiprocessing as mp

import numpy as np
import pytest

import zkcnn_amd

pytestmark = pytest.mark.gpu

MODEL, PIC, PP = "vgg11", (32, 32, 3), 1
REUSE, DRIVE, FULL_IPA = zkcnn_amd.MODE_REUSE_GENS, zkcnn_amd.MODE_DRIVE_ONLY, zkcnn_amd.MODE_FULL_IPA
SEED, SEED_BAD = 0x5EED0001, 0x5EED0041
RELU = 4


def _oracle_job(job):
    """(mode, seed, poke or None) -> (accepted, sha256, length) of the oracle's transcript; runs in its own process"""
    from tests import oracle_ffi
    mode, seed, poke = job
    with oracle_ffi.OracleSession(MODEL, PIC, PP) as o:
        if poke is not None:
            o.poke(*poke)
        res, tr = o.prove(seed=seed, mode=mode)
    return int(res.accepted), hashlib.sha256(tr).hexdigest(), len(tr)


def test_full_size_vgg11_identical_to_oracle(built, oracle):
    one = [int(v) for v in oracle.from_canonical(np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]]
    with zkcnn_amd.Session(MODEL, PIC, PP) as s:
        # the first RELU layer (1.5e6 entries: outputs, then constraint rows that are zero for a valid witness); its last row gets a 1
        i, relu = 0, None
        while relu is None:
            size, ty = s.layer_size(i)
            assert size >= 0, "no RELU layer found"
            if ty == RELU:
                relu = (i, size - 1, one)
            i += 1
        # ... and the reference's own semantics -- fresh generators drawn by the verifier for this proof, the inner-product argument run down to length 1
        # (reference src/verifier.cpp:119-128) -- plus the full argument over the session's public generators: the companions bench.py reports
        jobs = [(0, SEED, None), (REUSE | DRIVE, SEED, None), (REUSE, SEED_BAD, relu), (FULL_IPA | DRIVE, SEED + 1, None), (REUSE | FULL_IPA | DRIVE, SEED + 2, None)]
        with mp.get_context("spawn").Pool(5) as pool:
            pending = pool.map_async(_oracle_job, jobs)

            def digest(tr):
                return hashlib.sha256(tr).hexdigest(), len(tr)
            res, plain = s.prove(seed=SEED)                               # fresh generators drawn by the verifier, full verification
            assert res.accepted == 1, res.message.decode()
            res, first_use = s.prove(seed=SEED, mode=REUSE)               # session generators, first use: window / digit tables
            assert res.accepted == 1, res.message.decode()
            res, timed = s.prove(seed=SEED, mode=REUSE | DRIVE)           # second use: the byte table -- the path bench.py times
            assert res.accepted == -1
            assert first_use == timed, "drive-only transcript differs from the verified one"
            res, full_fresh = s.prove(seed=SEED + 1, mode=FULL_IPA)        # fresh generators, 12 argument rounds, fully verified
            assert res.accepted == 1, res.message.decode()
            res, full_reuse = s.prove(seed=SEED + 2, mode=REUSE | FULL_IPA)
            assert res.accepted == 1, res.message.decode()
            s.poke(*relu)
            bad_res, bad = s.prove(seed=SEED_BAD, mode=REUSE)
            (acc0, sha0, len0), (acc1, sha1, len1), (acc2, sha2, len2), (_, sha3, len3), (_, sha4, len4) = pending.get(timeout=900)
    assert acc0 == 1 and (sha0, len0) == digest(plain), "full-size vgg11, interactive mode: GPU transcript differs from the CPU oracle's"
    assert acc1 == -1 and (sha1, len1) == digest(timed), "full-size vgg11, REUSE_GENS | DRIVE_ONLY (bench mode): GPU transcript differs from the CPU oracle's"
    assert acc2 == 0 and bad_res.accepted == 0, "a non-zero constraint row at full size must be rejected (oracle and GPU)"
    assert (sha2, len2) == digest(bad), "full-size vgg11 with a corrupted witness: GPU transcript differs from the CPU oracle's"
    assert (sha3, len3) == digest(full_fresh), "full-size vgg11, fresh generators + full inner-product argument (reference semantics): GPU transcript differs from the CPU oracle's"
    assert (sha4, len4) == digest(full_reuse), "full-size vgg11, session generators + full inner-product argument: GPU transcript differs from the CPU oracle's"


def _oracle_lane_job(job):
    """(picture seed, statement, challenge seed, mode) -> (sha256, length) of the oracle's transcript for that picture on the calibrated circuit"""
    from tests import oracle_ffi
    picture_seed, stmt, seed, mode = job
    with oracle_ffi.OracleSession(MODEL, PIC, PP, picture_seed=picture_seed, calibrated=stmt) as o:
        res, tr = o.prove(seed=seed, mode=mode)
    return hashlib.sha256(tr).hexdigest(), len(tr)


@pytest.mark.parametrize("k", [2, 8])
def test_full_size_vgg11_lock_step_batch_identical_to_oracle(built, k):
    """BASELINE.json configs[2] as bench.py times it since round 4: k sessions of the full vgg11 circuit as the lanes of ONE lock-step batch (one host
    thread, one stream, one launch per sumcheck round for all lanes: reference src/prover.cpp:360-426 once per round per lane). Every lane's
    transcript -- its own picture, its own challenge seed -- must be the CPU oracle's, byte for byte (k oracle processes side by side), in the
    bench's modes; and a full round count of fused launches."""
    import threading
    first = zkcnn_amd.Session(MODEL, PIC, PP)
    stmt = first.statement()
    ss, pics = [first] + [None] * (k - 1), [0] * k

    def build(i):
        ss[i] = first.clone()                     # (a lane as bench.py makes it: no host circuit of its own)
        for ps in range(1000 * i, 1000 * i + 64):
            if ss[i].new_image(ps)[0] == 0:
                pics[i] = ps
                return
        ss[i].close()
        ss[i] = None
    th = [threading.Thread(target=build, args=(i,)) for i in range(1, k)]
    [t.start() for t in th]
    [t.join() for t in th]
    try:
        assert all(s is not None for s in ss), "no synthetic picture with the circuit's quantisation scales among 64"
        seeds = [SEED + 16 + i for i in range(k)]
        with mp.get_context("spawn").Pool(k) as pool:
            pending = pool.map_async(_oracle_lane_job, [(pics[i], stmt, seeds[i], REUSE | DRIVE) for i in range(k)])
            with zkcnn_amd.BatchSession(ss) as B:
                ver = B.prove(seeds=seeds, mode=REUSE)                     # fully verified (first use of the generators: window / digit tables)
                assert [r.accepted for r, _ in ver] == [1] * k, [r.message for r, _ in ver]
                B.prove(seeds=seeds, mode=REUSE | DRIVE)                   # second use: the byte table is built
                before = B.stats()
                got = B.prove(seeds=seeds, mode=REUSE | DRIVE)             # the timed path of bench.py
                st = B.stats()
            want = pending.get(timeout=900)
        for i in range(k):
            assert ver[i][1] == got[i][1], f"lane {i}: drive-only transcript differs from the verified one"
            assert (hashlib.sha256(got[i][1]).hexdigest(), len(got[i][1])) == want[i], f"lane {i}: full-size batch transcript differs from the CPU oracle's"
        fused, lanes = st["fused_launches"] - before["fused_launches"], st["lane_launches"] - before["lane_launches"]
        assert lanes == k * fused, (fused, lanes)
        assert fused <= 1.35 * got[0][0].n_rounds, f"{fused} fused launches for a proof of {got[0][0].n_rounds} rounds"
    finally:
        for s in ss:
            if s is not None:
                s.close()


def test_vgg16_thirty_two_pictures_as_one_circuit(built):
    """BASELINE.json configs[4] in the reference's OWN meaning (reference src/main_demo_vgg.cpp:36: vgg16 with pic_cnt = 32 is ONE circuit; layer 0 =
    2^28 entries, ~38 GB of layer values, 16 384 x 16 384 commitment matrix): it fits one MI355X (110 GB of HBM for circuit + witness + tables, 140 GB
    with the generator tables). The CPU oracle cannot hold it in the build container, so the evidence is the verifier's -- every round identity, every
    wiring predicate on the GPU, the Hyrax opening -- plus determinism, replay of the serialized proof and rejection of a corrupted message."""
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except ImportError:
        avail = None
    if avail is not None and avail < 300e9:
        # (round-4 review: a green run on a smaller box must not look like coverage -- with ZKCNN_REQUIRE_BIG=1 the test FAILS instead of skipping)
        msg = f"needs ~250 GB of host memory for the circuit and witness of 32 pictures; {avail / 1e9:.0f} GB available"
        if os.environ.get("ZKCNN_REQUIRE_BIG") == "1":
            pytest.fail("ZKCNN_REQUIRE_BIG=1: " + msg)
        pytest.skip(msg)
    TAMPER = zkcnn_amd.MODE_TAMPER
    with zkcnn_amd.Session("vgg16", PIC, 32) as s:
        res, tr = s.prove(seed=SEED, mode=REUSE)
        assert res.accepted == 1, res.message.decode()
        assert res.n_layers == 99 and res.input_bits == 28 and res.n_rounds == 2605
        r2, tr2 = s.prove(seed=SEED, mode=REUSE | DRIVE)
        assert tr2 == tr
        # BYTE parity at 2^28 (round 6): the CPU oracle's transcript of this circuit was made once on a GPU box's host (tests/golden/make_golden_full.py
        # vgg16_pp32: ~40 minutes and > 100 GB on one core) and its SHA-256 + length are a committed fixture
        import json
        gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "full_size.json")))["full_size"].get("vgg16_pp32")
        if gold is not None:
            g = gold["reuse_gens"]
            _, tr3 = s.prove(seed=g["challenge_seed"], mode=REUSE | DRIVE)
            assert len(tr3) == g["transcript_len"] and hashlib.sha256(tr3).hexdigest() == g["sha256"], "vgg16 pic_cnt = 32: transcript differs from the CPU oracle's"
        assert s.verify(tr, seed=SEED, mode=REUSE).accepted == 1
        bad, _ = s.prove(seed=SEED, mode=REUSE | TAMPER | ((res.n_messages // 2) << 8))
        assert bad.accepted == 0
        print(f"vgg16 pic_cnt=32 as one circuit: prover {1e3 * (r2.prove_s + r2.poly_prove_s):.1f} ms = {32 / (r2.prove_s + r2.poly_prove_s):.0f} pictures/s, proof {res.proof_kb + res.poly_proof_kb:.0f} KB")
