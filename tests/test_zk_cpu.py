"""Zero-knowledge mode (ZKCNN_MODE_ZK) on the CPU oracle: blinded commitments over (g, H), masked round polynomials,
proofs of dot product instead of the inner-product argument. The reference has no such mode (reference README.md:5), so these are
protocol properties: completeness, soundness against every corrupted prover message, hiding of the commitments, and the algebra of the
masks against an independent Python restatement."""
import hashlib

import pytest

import zkcnn_amd
from tests import oracle_ffi

ZK, FS, REUSE, TAMPER = zkcnn_amd.MODE_ZK, zkcnn_amd.MODE_FIAT_SHAMIR, zkcnn_amd.MODE_REUSE_GENS, zkcnn_amd.MODE_TAMPER
R = zkcnn_amd.R_MOD
CASES = [("custom:F8 F4", (4, 4, 1), 1), ("custom:C2:3:1:f M F4", (8, 8, 1), 2), ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2)]


@pytest.mark.parametrize("model,pic,pp", CASES)
def test_zero_knowledge_mode_is_complete_and_reproducible(oracle, model, pic, pp):
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        plain, t_plain = o.prove(seed=5)
        zk, t1 = o.prove(seed=5, mode=ZK)
        assert plain.accepted == 1 and zk.accepted == 1, zk.message
        assert len(t1) > len(t_plain) and zk.n_messages > plain.n_messages          # one g(r) per sumcheck instance, longer openings
        assert o.prove(seed=5, mode=ZK)[1] == t1                                      # seeded: challenges AND prover coins reproducible
        a, ta = o.prove(mode=ZK)
        b, tb = o.prove(mode=ZK)                                                      # default: OS randomness on both sides
        assert a.accepted == 1 and b.accepted == 1 and ta != tb
        for mode in (ZK | REUSE, ZK | FS, ZK | FS | REUSE, ZK | zkcnn_amd.MODE_DRIVE_ONLY):
            r, tr = o.prove(seed=7, mode=mode)
            assert r.accepted == (-1 if mode & zkcnn_amd.MODE_DRIVE_ONLY else 1), (mode, r.message)
            if not mode & zkcnn_amd.MODE_DRIVE_ONLY:
                assert o.verify(tr, seed=7, mode=mode).accepted == 1                  # serialized zero-knowledge proofs replay
                bad = bytearray(tr)
                bad[len(bad) // 2] ^= 1
                assert o.verify(bytes(bad), seed=7, mode=mode).accepted == 0
        assert o.verify(t1, seed=5, mode=0).accepted == 0                             # not a plain transcript


def test_blinded_commitments_hide_the_witness(oracle):
    """same witness, same public generators: the plain commitment is a function of the witness, the blinded one is fresh every time"""
    with oracle_ffi.OracleSession(*CASES[1]) as o:
        _, p1 = o.prove(seed=1, mode=REUSE)
        _, p2 = o.prove(seed=2, mode=REUSE)
        rows = 1 << (o.prove(seed=1)[0].input_bits // 2)
        assert p1[:48 * rows] == p2[:48 * rows]
        _, z1 = o.prove(mode=ZK | REUSE)
        _, z2 = o.prove(mode=ZK | REUSE)
        c1 = {z1[48 * i:48 * i + 48] for i in range(rows)}
        c2 = {z2[48 * i:48 * i + 48] for i in range(rows)}
        assert not (c1 & c2) and not (c1 & {p1[48 * i:48 * i + 48] for i in range(rows)})


def test_every_zero_knowledge_message_is_checked(oracle):
    """a cheating prover: corrupting any message that the plain protocol checks is still rejected, and so is every message the mode adds
    (each revealed g(r), both messages of the masks' proof of dot product, the three parts of the input's proof of dot product)"""
    model, pic, pp = CASES[1]
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        plain, _ = o.prove(seed=11)
        zk, _ = o.prove(seed=11, mode=ZK)
        n0, n1 = plain.n_messages, zk.n_messages
        slack0 = [k for k in range(n0) if o.prove(seed=11, mode=TAMPER | (k << 8))[0].accepted != 0]
        slack1 = [k for k in range(n1 + 3) if o.prove(seed=11, mode=ZK | TAMPER | (k << 8))[0].accepted != 0]
        # the plain protocol has a few claims nobody reads (an operand table that does not exist in that layer); masking adds none
        assert len(slack1) == len(slack0), (slack0, slack1)
        assert all(k < n1 - 2 for k in slack1)          # the last two sumcheck-phase messages are the masks' proof of dot product
        assert o.prove(seed=11, mode=ZK | TAMPER | ((n1 + 3) << 8))[0].accepted == 1      # past the last message: nothing to corrupt


def test_mask_algebra_against_python(oracle):
    """q_j of zk_mask.hpp restated with Python integers for a random g: q_0(0) + q_0(1) = G, q_j(r_j) = q_{j+1}(0) + q_{j+1}(1), and
    q_{l-1}(r_{l-1}) = g(r) -- the identities that make the masked transcript a valid sumcheck of f + rho g"""
    import random
    rnd = random.Random(7)
    for ell, deg in ((1, 2), (5, 2), (7, 3)):
        a0 = rnd.randrange(R)
        a = [[rnd.randrange(R) for _ in range(deg)] for _ in range(ell)]
        r = [rnd.randrange(R) for _ in range(ell)]
        gi = lambda i, t: sum(a[i][e] * pow(t, e + 1, R) for e in range(deg)) % R        # noqa: E731
        G = (pow(2, ell, R) * a0 + pow(2, ell - 1, R) * sum(gi(i, 1) for i in range(ell))) % R
        brute = 0
        for x in range(1 << ell):
            brute += a0 + sum(gi(i, (x >> i) & 1) for i in range(ell))
        assert brute % R == G

        def q(j, t):
            s = a0 + sum(gi(i, r[i]) for i in range(j)) + gi(j, t)
            rest = ell - j - 1
            out = pow(2, rest, R) * s
            if rest > 0:
                out += pow(2, rest - 1, R) * sum(gi(i, 1) for i in range(j + 1, ell))
            return out % R
        assert (q(0, 0) + q(0, 1)) % R == G
        for j in range(ell - 1):
            assert q(j, r[j]) == (q(j + 1, 0) + q(j + 1, 1)) % R
        assert q(ell - 1, r[ell - 1]) == (a0 + sum(gi(i, r[i]) for i in range(ell))) % R


def test_zero_knowledge_golden_hash(oracle):
    """regression vector of the mode (made by this repo's oracle: parity unpinned like every other vector here)"""
    with oracle_ffi.OracleSession(*CASES[1]) as o:
        _, tr = o.prove(seed=0x5EED0001, mode=ZK | REUSE)
    assert len(tr) == 36432 and hashlib.sha256(tr).hexdigest() == ZK_GOLDEN


ZK_GOLDEN = "411d4703d3f9cab3c776419ace96fa7daa0b357166fd57c47055a57bf71d100a"
