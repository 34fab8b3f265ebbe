"""Serialized proofs and the optional Fiat-Shamir mode on the CPU: SHA-256 and point decompression against
known answers, replay verification of transcripts (interactive seed and Fiat-Shamir), rejection of every kind of damage."""
import hashlib

import numpy as np
import pytest

import zkcnn_amd
from zkcnn_amd import proof_io
from tests import oracle_ffi

MODEL = ("custom:C2:3:1:f M F4", (8, 8, 1), 2)      # FFT conv + max pool + FC: every message type occurs
FS = zkcnn_amd.MODE_FIAT_SHAMIR


def test_sha256_matches_hashlib(oracle):
    rng = np.random.default_rng(1)
    for n in [0, 1, 3, 55, 56, 57, 63, 64, 65, 119, 120, 128, 1000, 4097]:
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        for split in (0, 1, 64, n // 2):
            assert oracle.sha256(data, split) == hashlib.sha256(data).digest(), (n, split)
    assert oracle.sha256(b"abc").hex() == "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"


def test_blake2s_matches_hashlib_and_the_word_chain(oracle):
    """the Fiat-Shamir hash (ff/blake2s.hpp, RFC 7693) against hashlib.blake2s, and the word-level chain step the GPU uses against the same"""
    import struct
    rng = np.random.default_rng(2)
    for n in [0, 1, 3, 32, 63, 64, 65, 127, 128, 129, 200, 1000, 4097]:
        data = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        for split in (0, 1, 64, n // 2):
            assert oracle.blake2s(data, split) == hashlib.blake2s(data).digest(), (n, split)
    for n_words in (0, 8, 24):              # empty retry step, one claim, a quadratic round polynomial (exactly two compressions)
        state = [int(x) for x in rng.integers(0, 1 << 32, size=8)]
        msg = [int(x) for x in rng.integers(0, 1 << 32, size=n_words)]
        want = hashlib.blake2s(struct.pack("<8I", *state) + struct.pack(f"<{n_words}I", *msg)).digest()
        assert struct.pack("<8I", *oracle.blake2s_chain(state, msg)) == want


def test_g1_compression_round_trip_and_rejections(oracle):
    base = oracle.g1_base()
    ks = oracle.random(6, 99)
    for i in range(6):
        p = oracle.g1_mul(base, ks[i:i + 1])
        enc = oracle.g1_serialize(p)
        back = oracle.g1_deserialize(enc)
        assert back is not None and np.array_equal(back, p)
        flipped = bytes([enc[0] ^ 0x20]) + enc[1:]                  # the other root: still a valid point, the negated one
        neg = oracle.g1_deserialize(flipped)
        assert neg is not None and not np.array_equal(neg, p) and oracle.g1_serialize(neg) == flipped
        assert oracle.g1_deserialize(bytes([enc[0] & 0x7f]) + enc[1:]) is None          # compression flag missing
    # generator of BLS12-381 G1 in the standard compressed form (the value every implementation publishes)
    g = bytes.fromhex("97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb")
    assert oracle.g1_serialize(base) == g and np.array_equal(oracle.g1_deserialize(g), base)
    inf = bytes([0xc0]) + bytes(47)
    assert oracle.g1_deserialize(inf) is not None and oracle.g1_deserialize(bytes([0xc0]) + bytes(46) + b"\x01") is None
    p_mod = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    too_big = (p_mod + 1).to_bytes(48, "big")
    assert oracle.g1_deserialize(bytes([too_big[0] | 0x80]) + too_big[1:]) is None     # x >= p
    # points on the curve but outside the order-r subgroup (almost every curve point: the cofactor is ~2^126), and x with no y
    on_curve = off_curve = 0
    for x in range(1, 40):
        enc = x.to_bytes(48, "big")
        enc = bytes([enc[0] | 0x80]) + enc[1:]
        if oracle.g1_deserialize(enc, check_subgroup=False) is None:
            off_curve += 1
        else:
            on_curve += 1
            assert oracle.g1_deserialize(enc, check_subgroup=True) is None
    assert on_curve > 5 and off_curve > 5


@pytest.mark.parametrize("mode", [0, zkcnn_amd.MODE_REUSE_GENS, FS, FS | zkcnn_amd.MODE_REUSE_GENS])
def test_replay_accepts_and_rejects(oracle, mode):
    with oracle_ffi.OracleSession(*MODEL) as o:
        res, tr = o.prove(seed=41, mode=mode)
        assert res.accepted == 1
        ok = o.verify(tr, seed=41, mode=mode)
        assert ok.accepted == 1, ok.message
        assert ok.n_messages == res.n_messages
        # challenges of another run do not fit the proof
        other = o.verify(tr, seed=42, mode=mode)
        assert other.accepted == (1 if mode & FS else 0)                   # Fiat-Shamir ignores the seed
        assert o.verify(tr, seed=41, mode=mode ^ FS).accepted == 0          # wrong challenge derivation
        # damage: any flipped bit, truncation, extension
        rng = np.random.default_rng(5)
        for pos in sorted(set([0, 47, 48, len(tr) - 1] + list(rng.integers(0, len(tr), size=24)))):
            bad = bytearray(tr)
            bad[pos] ^= 1 << int(rng.integers(0, 8))
            r = o.verify(bytes(bad), seed=41, mode=mode)
            assert r.accepted == 0, f"byte {pos} of {len(tr)} damaged but accepted"
        assert o.verify(tr[:-32], seed=41, mode=mode).accepted == 0
        assert o.verify(tr + bytes(32), seed=41, mode=mode).accepted == 0
        assert o.verify(b"", seed=41, mode=mode).accepted == 0


def test_fiat_shamir_is_deterministic_and_binds_the_statement(oracle):
    with oracle_ffi.OracleSession(*MODEL) as o:
        _, a = o.prove(seed=1, mode=FS)
        _, b = o.prove(seed=2, mode=FS)
        _, c = o.prove(seed=1)
        assert a == b and a != c                    # no dependence on the seed; differs from the interactive transcript
    with oracle_ffi.OracleSession(MODEL[0], MODEL[1], MODEL[2], data_seed=7) as o2:
        _, d = o2.prove(seed=1, mode=FS)
        assert d != a                               # other picture and weights: other commitment, other challenges
    with oracle_ffi.OracleSession("custom:C2:3:1:f M F5", MODEL[1], MODEL[2]) as o3:
        assert o3.verify(a, mode=FS).accepted == 0  # the statement (circuit) is hashed into every challenge


def test_proof_file_round_trip(oracle, tmp_path):
    with oracle_ffi.OracleSession(*MODEL) as o:
        for mode, seed in ((FS, 0), (zkcnn_amd.MODE_REUSE_GENS, 77)):
            _, tr = o.prove(seed=seed, mode=mode)
            path = tmp_path / f"proof_{mode}.zkp"
            proof_io.save(path, tr, MODEL[0], MODEL[1], MODEL[2], 20260928, seed, mode)
            header, back = proof_io.load(path)
            assert back == tr and header["model"] == MODEL[0]
            if mode & FS:
                assert proof_io.verify_with(o, path.read_bytes()).accepted == 1
            else:           # an interactive transcript is not a proof off line: refused unless a replay is asked for explicitly
                with pytest.raises(proof_io.NotAProofError):
                    proof_io.verify_with(o, path.read_bytes())
                assert proof_io.verify_with(o, path.read_bytes(), allow_seeded_replay=True).accepted == 1
            blob = bytearray(path.read_bytes())
            blob[len(blob) // 2] ^= 0x40
            with pytest.raises(ValueError):
                proof_io.loads(bytes(blob))


@pytest.mark.parametrize("extra", [zkcnn_amd.MODE_ZK, zkcnn_amd.MODE_FULL_IPA, zkcnn_amd.MODE_ZK | zkcnn_amd.MODE_REUSE_GENS])
def test_proof_file_keeps_the_protocol_shaping_mode_bits(oracle, extra):
    """a zero-knowledge / full-argument proof holds other messages than a plain one: the file header must carry those bits, or the replay
    parses it as a plain proof and rejects it (round-2 advisor finding); run-only bits (TAMPER, DRIVE_ONLY, SEEDED) never reach the file"""
    mode = FS | extra
    with oracle_ffi.OracleSession(*MODEL) as o:
        res, tr = o.prove(seed=9, mode=mode | zkcnn_amd.MODE_SEEDED)
        assert res.accepted == 1
        blob = proof_io.dumps_from(o, tr, 9, mode | zkcnn_amd.MODE_SEEDED | zkcnn_amd.MODE_DRIVE_ONLY)
        header, _ = proof_io.loads(blob)
        assert header["mode"] == mode & (FS | zkcnn_amd.MODE_REUSE_GENS | zkcnn_amd.MODE_ZK | zkcnn_amd.MODE_FULL_IPA)
        assert proof_io.verify_with(o, blob).accepted == 1
    assert proof_io.verify_standalone(blob, oracle_ffi.OracleSession).accepted == 1
    plain = dict(header, mode=FS)               # the same bytes under a plain header: the messages do not parse as a plain proof
    import json, struct, hashlib
    h = json.dumps(plain, sort_keys=True).encode()
    body = proof_io.MAGIC + struct.pack("<I", len(h)) + h + struct.pack("<Q", len(tr)) + bytes(tr)
    assert proof_io.verify_standalone(body + hashlib.sha256(body).digest(), oracle_ffi.OracleSession).accepted == 0


@pytest.mark.parametrize("model", [MODEL, ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2), ("custom:C2:3:1:n A F4", (4, 4, 2), 1)])
def test_standalone_verifier_needs_only_the_proof_file(oracle, model):
    """the circuit is rebuilt from the model descriptor + the recorded quantisation scales (no picture, weights or witness) and the
    proof file verifies against it; a verifier-only session cannot prove; a wrong statement does not verify"""
    with oracle_ffi.OracleSession(*model) as o:
        stmt = o.statement()
        assert len(stmt) >= 2
        _, tr = o.prove(seed=5, mode=FS)
        blob = proof_io.dumps_from(o, tr, 5, FS)
        _, tr2 = o.prove(seed=6)
        blob2 = proof_io.dumps_from(o, tr2, 6, 0)
    assert proof_io.verify_standalone(blob, oracle_ffi.OracleSession).accepted == 1
    with pytest.raises(proof_io.NotAProofError):            # an interactive transcript names its own challenge seed: not a proof
        proof_io.verify_standalone(blob2, oracle_ffi.OracleSession)
    assert proof_io.verify_standalone(blob2, oracle_ffi.OracleSession, allow_seeded_replay=True).accepted == 1
    with oracle_ffi.OracleSession(*model, statement=stmt) as v:
        assert v.statement() == stmt
        with pytest.raises(RuntimeError):
            v.prove(seed=1)
        bad = bytearray(tr)
        bad[100] ^= 1
        assert v.verify(bytes(bad), mode=FS).accepted == 0
    # a statement with another scale: the circuit (shift constants of the re-quantisation gates) differs, the proof does not fit;
    # a statement of the wrong length is refused outright
    wrong = list(stmt)
    wrong[0] += 1                       # the picture's scale: one more bit is cut by the first re-quantisation, its layers grow
    with oracle_ffi.OracleSession(*model, statement=wrong) as v2:
        assert v2.verify(tr, mode=FS).accepted == 0
    with pytest.raises(RuntimeError):
        oracle_ffi.OracleSession(*model, statement=stmt[:-1])
    with pytest.raises(RuntimeError):
        oracle_ffi.OracleSession(*model, statement=stmt + [3])


def test_interactive_transcripts_are_not_proofs_off_line(oracle):
    """zkcnn_session_verify (here: the oracle's twin of it) refuses a transcript whose challenges it cannot re-derive; with OS randomness
    (the default: no seed) two runs differ, both are accepted in process, and neither verifies off line"""
    with oracle_ffi.OracleSession(*MODEL) as o:
        r1, t1 = o.prove()                       # challenges from getrandom(2)
        r2, t2 = o.prove()
        assert r1.accepted == 1 and r2.accepted == 1 and t1 != t2
        res = o.verify(t1)                       # no Fiat-Shamir, no seeded replay
        assert res.accepted == 0 and b"Fiat-Shamir" in res.message
        assert o.verify(t1, seed=1).accepted == 0          # replaying against some seeded stream does not fit either
        _, t3 = o.prove(mode=zkcnn_amd.MODE_REUSE_GENS)
        _, t4 = o.prove(mode=zkcnn_amd.MODE_REUSE_GENS)
        assert t3 != t4 and t3[:48] == t4[:48]   # same witness, same public generators: same commitment, other challenges


def test_public_generators_are_hash_to_curve_points(oracle):
    """session / Fiat-Shamir generators: on the curve, in the order-r subgroup, pairwise distinct, prefix-stable, and not multiples of G
    by a scalar anybody chose (they are a function of the index alone: recomputed here from the definition for the first point)"""
    g64, dig64 = oracle.public_generators(64)
    g16, dig16 = oracle.public_generators(16)
    assert np.array_equal(g64[:16], g16) and dig16 != dig64
    assert len({g.tobytes() for g in g64}) == 64
    for g in g64[:8]:
        assert oracle.g1_on_curve(g) and oracle.g1_in_subgroup(g)
    enc = b"".join(oracle.g1_serialize(g) for g in g16)
    assert hashlib.sha256(enc).digest() == dig16
    # definition: x = (H(d || i || ctr || 1) mod 2^255) * 2^256 + H(d || i || ctr || 0) mod p, first ctr with a square x^3 + 4, times h_eff
    p_mod = zkcnn_amd.P_MOD
    dom, i = b"zkcnn-amd/generators/v1", 0
    ctr = 0
    while True:
        tail = i.to_bytes(8, "little") + ctr.to_bytes(4, "little")
        lo = int.from_bytes(hashlib.sha256(dom + tail + b"\x00").digest(), "little")
        hi = int.from_bytes(hashlib.sha256(dom + tail + b"\x01").digest(), "little")
        pick_larger, hi = hi >> 255, hi & ((1 << 255) - 1)
        x = (hi * (1 << 256) + lo) % p_mod
        rhs = (x * x * x + 4) % p_mod
        y = pow(rhs, (p_mod + 1) // 4, p_mod)
        if y * y % p_mod == rhs:
            break
        ctr += 1
    if (y > p_mod - y) != bool(pick_larger):
        y = p_mod - y

    def add(P, Q):                                # affine short Weierstrass, a = 0
        if P is None:
            return Q
        if Q is None:
            return P
        if P[0] == Q[0]:
            if (P[1] + Q[1]) % p_mod == 0:
                return None
            lam = 3 * P[0] * P[0] * pow(2 * P[1], -1, p_mod) % p_mod
        else:
            lam = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p_mod) % p_mod
        x3 = (lam * lam - P[0] - Q[0]) % p_mod
        return x3, (lam * (P[0] - x3) - P[1]) % p_mod
    acc = None
    for bit in bin(0xd201000000010001)[2:]:
        acc = add(acc, acc)
        if bit == "1":
            acc = add(acc, (x, y))
    got = oracle.fp_to_canonical(g64[0])
    assert zkcnn_amd.from_limbs(got) == [acc[0], acc[1]]


def test_statement_with_out_of_range_scales_is_refused(oracle):
    """quantisation scales size layers and index two_mul; a statement (proof file) is untrusted input: out-of-range values must fail
    cleanly instead of building giant or negative-sized layers"""
    with oracle_ffi.OracleSession(*MODEL) as o:
        stmt = o.statement()
    for k, v in ((0, -3), (0, 200), (1, 70), (len(stmt) - 1, -1), (len(stmt) - 1, 10 ** 6)):
        bad = list(stmt)
        bad[k] = v
        with pytest.raises(RuntimeError):
            oracle_ffi.OracleSession(*MODEL, statement=bad)
    with pytest.raises(RuntimeError):
        oracle_ffi.OracleSession(MODEL[0], (1 << 20, 4, 1), 1, statement=stmt)      # absurd picture shape
