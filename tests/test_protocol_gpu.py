"""Whole-proof parity on the GPU: the HIP prover driven by the verifier must (1) be accepted and (2) produce
the byte-identical canonical transcript as the CPU oracle prover under the same seeded challenge stream."""
import hashlib

import pytest

import zkcnn_amd
from tests import oracle_ffi

pytestmark = pytest.mark.gpu

CASES = [
    ("custom:F4", (4, 4, 1), 1),
    ("custom:F8 F4", (4, 4, 1), 1),
    ("custom:C2:3:1:s F4", (4, 4, 1), 1),
    ("custom:C2:3:1:s A F4", (4, 4, 1), 1),
    ("custom:C2:3:1:s M F4", (4, 4, 1), 1),
    ("custom:C2:3:1:n A F4", (4, 4, 2), 1),
    ("custom:C2:3:1:f F4", (4, 4, 1), 1),
    ("custom:C2:3:1:f M F4", (8, 8, 1), 2),
    ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2),
    ("lenet", (32, 32, 1), 1),
    ("custom:C3:3:1:f A C2:3:1:f M F6 F3", (8, 8, 1), 3),          # ragged batch (3 pictures), FFT conv + avg + max pool
    ("lenetCifar", (32, 32, 3), 1),                                 # three FFT conv blocks, 28 layers
    ("lenet.avg", (28, 28, 1), 1),                                  # MNIST-size input, padded first conv, average pooling
    ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),  # vgg11 at quarter width: 8e6 mul gates, 2^21 inputs
]


@pytest.mark.parametrize("model,pic,pp", CASES)
def test_gpu_transcript_identical_to_oracle(built, model, pic, pp):
    with zkcnn_amd.Session(model, pic, pp) as s:
        res, gpu = s.prove(seed=0x5EED0001)
        assert res.accepted == 1, res.message.decode()
        res2, gpu2 = s.prove(seed=0x5EED0002)          # same session, second proof, other challenges
        assert res2.accepted == 1 and gpu2 != gpu
        res3, gpu3 = s.prove(seed=0x5EED0001, mode=zkcnn_amd.MODE_DRIVE_ONLY)
        assert res3.accepted == -1 and gpu3 == gpu       # drive-only makes the same calls with the same challenges
        # verifier's wiring predicates: GPU (default) == host loops of the reference verifier, value for value, in every layer
        res4, gpu4 = s.prove(seed=0x5EED0003, mode=zkcnn_amd.MODE_CROSS_PRED)
        assert res4.accepted == 1, res4.message.decode()
        res5, gpu5 = s.prove(seed=0x5EED0003, mode=zkcnn_amd.MODE_HOST_PRED)
        assert res5.accepted == 1 and gpu5 == gpu4
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        ores, cpu = o.prove(seed=0x5EED0001)
        assert ores.accepted == 1
    assert len(gpu) == len(cpu) and hashlib.sha256(gpu).hexdigest() == hashlib.sha256(cpu).hexdigest()
    assert abs(res.proof_kb - ores.proof_kb) < 1e-9 and abs(res.poly_proof_kb - ores.poly_proof_kb) < 1e-9


@pytest.mark.parametrize("case,want", [(7, 1), (8, 1), (10, 1), (9, 0), (6, 0)])
def test_dot_prod_tables_of_several_pictures_in_factored_form(built, case, want):
    """FFT convolutions over pic_cnt >= 2 pictures with a power-of-two channel_out: the DOT_PROD layer's phase-1 table is beta_hi[p] * S[ci, t]
    (upload.hip checks the gate pattern; kernels.cuh k_dot_s) -- the transcripts of these circuits are compared with the oracle's above, here:
    that the factored builder is what ran (3 pictures: pp need not be a power of two; channel_out = 3 or one picture: generic builder)."""
    model, pic, pp = CASES[case]
    with zkcnn_amd.Session(model, pic, pp) as s:
        assert s.factored_dot_layers() == want


def test_full_size_vgg11_accepted_and_deterministic(built):
    """BASELINE.json configs[2] at full size (1.16e8 multiplication gates, 2^24 inputs): the verifier -- which checks the sumcheck
    round identity p(0) + p(1) == claim in every one of the 1129 rounds, every layer's wiring predicate and the Hyrax opening --
    accepts the GPU proof, and proving twice under one seed gives the same bytes (size-independent properties; the byte-for-byte
    comparison with the CPU oracle at this size takes ~1 min of CPU and is run by scripts/, see profiles/r01_summary.md)."""
    with zkcnn_amd.Session("vgg11", (32, 32, 3), 1) as s:
        res, t1 = s.prove(seed=0x5EED0001)
        assert res.accepted == 1, res.message.decode()
        assert res.n_layers == 37 and res.input_bits == 24 and res.gate_cnt_bin == 115904256
        res2, t2 = s.prove(seed=0x5EED0001, mode=zkcnn_amd.MODE_DRIVE_ONLY)
        assert t1 == t2
        bad, _ = s.prove(seed=0x5EED0001, mode=zkcnn_amd.MODE_TAMPER | (700 << 8))
        assert bad.accepted == 0
        cross, _ = s.prove(seed=0x5EED0004, mode=zkcnn_amd.MODE_CROSS_PRED)     # GPU predicates == host predicates on 1.3e8 gates
        assert cross.accepted == 1, cross.message.decode()
        print(f"verifier: GPU predicates {res.verify_s:.3f} s, host predicates {cross.verify_s - res.verify_s:.3f} s more")


def test_gpu_verifier_rejects_any_corrupted_message(built):
    """soundness smoke test with the verifier's predicates on the GPU: perturbing any single prover message must be rejected, and
    the abandoned proofs must leave the session usable (the next proof is accepted)"""
    with zkcnn_amd.Session("custom:C2:3:1:f M F4", (8, 8, 1), 2) as s:
        ok, _ = s.prove(seed=11)
        assert ok.accepted == 1
        n_sum = ok.n_messages
        for k in sorted(set(list(range(0, n_sum, 5)) + [n_sum - 1, n_sum])):
            bad, _ = s.prove(seed=11, mode=zkcnn_amd.MODE_TAMPER | (k << 8))
            assert bad.accepted == 0, f"message {k} of {n_sum} corrupted but accepted"
        again, _ = s.prove(seed=12)
        assert again.accepted == 1


def test_concurrent_gpu_sessions_are_deterministic(built):
    """three proofs in flight on one GPU (own session, host thread and HIP stream each): same bytes as one at a time"""
    import threading
    model, pic, pp = "custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2
    sessions = [zkcnn_amd.Session(model, pic, pp, data_seed=100 + i) for i in range(3)]
    try:
        alone = [s.prove(seed=5)[1] for s in sessions]
        got = [None] * 3

        def work(i):
            for _ in range(4):
                res, tr = sessions[i].prove(seed=5)
                assert res.accepted == 1
            got[i] = tr
        th = [threading.Thread(target=work, args=(i,)) for i in range(3)]
        [t.start() for t in th]
        [t.join() for t in th]
        assert got == alone and len(set(alone)) == 3
    finally:
        for s in sessions:
            s.close()


GOLDEN = __import__("json").load(open(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "transcripts.json")))


@pytest.mark.parametrize("key", sorted(GOLDEN))
def test_gpu_matches_golden_transcripts(built, key):
    """committed fixtures (tests/golden): the GPU prover reproduces the interactive and the Fiat-Shamir transcript of every case,
    and the serialized proofs verify without the prover (replay) with the predicates on the GPU"""
    g = GOLDEN[key]
    with zkcnn_amd.Session(g["model"], tuple(g["pic"]), g["pic_cnt"], data_seed=g["data_seed"]) as s:
        res, tr = s.prove(seed=g["challenge_seed"])
        assert res.accepted == 1 and hashlib.sha256(tr).hexdigest() == g["sha256"]
        fres, ftr = s.prove(seed=9, mode=zkcnn_amd.MODE_FIAT_SHAMIR)
        assert fres.accepted == 1 and hashlib.sha256(ftr).hexdigest() == g["fiat_shamir_sha256"]
        assert s.verify(tr, seed=g["challenge_seed"]).accepted == 1
        assert s.verify(ftr, mode=zkcnn_amd.MODE_FIAT_SHAMIR).accepted == 1
        bad = bytearray(ftr)
        bad[len(bad) // 3] ^= 4
        assert s.verify(bytes(bad), mode=zkcnn_amd.MODE_FIAT_SHAMIR).accepted == 0


def test_standalone_gpu_verifier_from_proof_file(built):
    """a proof file made by one session verifies in a verifier-only session built from the file alone (circuit from model descriptor +
    statement, wiring predicates on the GPU, no witness anywhere); LeNet5 and a proof made by the CPU oracle as well"""
    from zkcnn_amd import proof_io
    for model, pic, pp in (("lenet", (32, 32, 1), 1), ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2)):
        with zkcnn_amd.Session(model, pic, pp) as s:
            _, tr = s.prove(seed=0, mode=zkcnn_amd.MODE_FIAT_SHAMIR)
            blob = proof_io.dumps_from(s, tr, 0, zkcnn_amd.MODE_FIAT_SHAMIR)
            stmt = s.statement()
        res = proof_io.verify_standalone(blob, zkcnn_amd.Session)
        assert res.accepted == 1, res.message.decode()
        with oracle_ffi.OracleSession(model, pic, pp) as o:
            assert o.statement() == stmt
            _, otr = o.prove(seed=3)
            oblob = proof_io.dumps_from(o, otr, 3, 0)
        with pytest.raises(proof_io.NotAProofError):
            proof_io.verify_standalone(oblob, zkcnn_amd.Session)       # an interactive transcript: a replay must be asked for
        assert proof_io.verify_standalone(oblob, zkcnn_amd.Session, allow_seeded_replay=True).accepted == 1
        damaged = proof_io.dumps(tr[:200] + bytes([tr[200] ^ 2]) + tr[201:], model, pic, pp, 20260928, 0, zkcnn_amd.MODE_FIAT_SHAMIR, stmt)
        assert proof_io.verify_standalone(damaged, zkcnn_amd.Session).accepted == 0
