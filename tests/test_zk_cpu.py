"""Zero-knowledge mode (ZKCNN_MODE_ZK) on the CPU oracle: blinded commitments over (g, H), masked round polynomials,
masked evaluation claims (V + Z M, zkcnn_amd/csrc/host/zk_mask.hpp (2)), proofs of dot product instead of the inner-product argument. The
reference has no such mode (reference README.md:5), so these are protocol properties: completeness, soundness against every corrupted prover
message, hiding of the commitments, dependence of EVERY message on the prover's coins, the algebra of the masks against an independent Python
restatement, and the simulator identity of the whole masked sumcheck by exhaustion over a small field."""
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
        assert len(t1) > len(t_plain) and zk.n_messages > plain.n_messages          # one revealed value per sumcheck instance, longer openings
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


def _fold(T, r):
    return [(T[2 * i] + r * (T[2 * i + 1] - T[2 * i])) % R for i in range(len(T) // 2)]


def test_last_round_mask_against_first_principles():
    """zk_mask.hpp (2) restated from the definition: V'_b(x) = V_b(x_low) + Z(x) M_b with Z over ALL l variables of the phase, the multiplier tables
    zero padded to l variables (the reference's `total == 1` case: a short table is absorbed into add_term, times (1 - r) per round). Brute-force
    round polynomials of F = sum_b A_b V'_b against what the provers compute: the unmasked table-folding polynomial, plus -- in the last round only
    -- Z' t (1 - t) sum_b M_b A_b(t), with A_b(t) = m0 + (m1 - m0) t for a live pair and abs_m (1 - t) for an absorbed one."""
    import random
    rnd = random.Random(11)
    ell, bl = 3, (1, 3)                                        # pair 0 is two entries long: absorbed after its only variable
    V = [[rnd.randrange(R) for _ in range(1 << b)] for b in bl]
    A = [[rnd.randrange(R) for _ in range(1 << b)] for b in bl]
    M = [rnd.randrange(R) for _ in bl]
    r = [rnd.randrange(R) for _ in range(ell)]

    def mle(T, pt):                                            # little-endian variables, as the round kernels fold them
        for x in pt:
            T = _fold(T, x)
        return T[0]

    def F(pt):
        z = 1
        for x in pt:
            z = z * x * (1 - x) % R
        tot = 0
        for b in range(2):
            pad = 1
            for x in pt[bl[b]:]:
                pad = pad * (1 - x) % R
            tot += mle(A[b], pt[:bl[b]]) * pad * (mle(V[b], pt[:bl[b]]) + z * M[b])
        return tot % R

    def brute(j, t):
        tot = 0
        for rest in range(1 << (ell - j - 1)):
            tot += F(r[:j] + [t] + [(rest >> k) & 1 for k in range(ell - j - 1)])
        return tot % R

    # the provers' bookkeeping: fold live tables, absorb a pair when one entry is left, add_term x (1 - r) per round
    Vt, At = [list(v) for v in V], [list(a) for a in A]
    add_term, abs_m, absorbed = 0, [0, 0], [False, False]
    for j in range(ell):
        if j:
            add_term = add_term * (1 - r[j - 1]) % R
            abs_m = [m * (1 - r[j - 1]) % R if ab else m for m, ab in zip(abs_m, absorbed)]
        for b in range(2):
            if absorbed[b]:
                continue
            if j:
                Vt[b], At[b] = _fold(Vt[b], r[j - 1]), _fold(At[b], r[j - 1])
            if len(Vt[b]) == 1:
                add_term = (add_term + Vt[b][0] * At[b][0]) % R
                abs_m[b], absorbed[b] = At[b][0], True

        def plain(t):
            tot = add_term * (1 - t)
            for b in range(2):
                if not absorbed[b]:
                    for i in range(len(Vt[b]) // 2):
                        tot += (At[b][2 * i] + t * (At[b][2 * i + 1] - At[b][2 * i])) * (Vt[b][2 * i] + t * (Vt[b][2 * i + 1] - Vt[b][2 * i]))
            return tot % R
        for t in range(5):
            want = brute(j, t)
            if j < ell - 1:
                assert plain(t) == want, (j, t)                  # Z vanishes while one variable is boolean: nothing changes
            else:
                zp = 1
                for x in r[:ell - 1]:
                    zp = zp * x * (1 - x) % R
                corr = 0
                for b in range(2):
                    a_t = abs_m[b] * (1 - t) if absorbed[b] else At[b][0] + t * (At[b][1] - At[b][0])
                    corr += M[b] * a_t
                assert (plain(t) + zp * t * (1 - t) * corr) % R == want, t
    # the masked claims are what the next phase / the final check is about
    z = 1
    for x in r:
        z = z * x * (1 - x) % R
    claims = [(mle(V[b], r[:bl[b]]) + z * M[b]) % R for b in range(2)]
    pads = [1, 1]
    for b in range(2):
        for x in r[bl[b]:]:
            pads[b] = pads[b] * (1 - x) % R
    assert F(r) == sum(mle(A[b], r[:bl[b]]) * pads[b] * claims[b] for b in range(2)) % R


def test_masked_sumcheck_simulator_identity_by_exhaustion():
    """The whole masked sumcheck of zk_mask.hpp over F_5, every choice of the prover's coins enumerated (5^9): two variables, two operand pairs, an
    incoming mask term K, the masking polynomial g with degrees (2, 3), one mask per claim. For ANY witness the map coins -> transcript
    (claimed sum, G, both round messages, both masked claims, v) is a bijection onto the set of transcripts that pass the verifier's checks --
    so the verifier's view is the uniform distribution on a set that does not depend on the witness: a simulator samples it without one."""
    np = pytest.importorskip("numpy")
    P, r0, r1, rho = 5, 2, 3, 2
    n = P ** 9
    idx = np.arange(n, dtype=np.int64)
    coins = []
    for _ in range(9):
        coins.append(idx % P)
        idx = idx // P
    K, M0, M1, a0, a01, a02, a11, a12, a13 = coins
    A = [[1, 3, 2, 4], [2, 2, 1, 3]]                             # the multipliers are public in this instance (a phase 2 / the Liu sumcheck)

    def lerp(T, r):
        return [(T[2 * i] + r * (T[2 * i + 1] - T[2 * i])) % P for i in range(len(T) // 2)]
    g0 = lambda t: (a01 * t + a02 * t * t) % P                   # noqa: E731
    g1 = lambda t: (a11 * t + a12 * t * t + a13 * t ** 3) % P    # noqa: E731
    zp, z = r0 * (1 - r0) % P, r0 * (1 - r0) * r1 * (1 - r1) % P
    assert z != 0

    def view(V):
        H = sum(A[b][x] * V[b][x] for b in range(2) for x in range(4)) % P
        claim = (H + K) % P                                      # what the verifier holds: a combination of MASKED claims of the layer above
        G = (4 * a0 + 2 * (g0(1) + g1(1))) % P
        msgs = [claim, G]
        # round 0: variable 0 (the low bit), nothing of Z yet
        for t in range(3):
            p = 0
            for b in range(2):
                for x1 in range(2):
                    p += (A[b][2 * x1] + t * (A[b][2 * x1 + 1] - A[b][2 * x1])) * (V[b][2 * x1] + t * (V[b][2 * x1 + 1] - V[b][2 * x1]))
            q = 2 * (a0 + g0(t)) + g1(1)
            msgs.append((p + rho * q + K * (1 - t)) % P)
        Vf, Af = [lerp(V[b], r0) for b in range(2)], [lerp(A[b], r0) for b in range(2)]
        # round 1: the last one carries Z' t (1 - t) sum_b M_b A_b(t)
        for t in range(4):
            p = 0
            for b, Mb in ((0, M0), (1, M1)):
                a_t = Af[b][0] + t * (Af[b][1] - Af[b][0])
                p = p + a_t * (Vf[b][0] + t * (Vf[b][1] - Vf[b][0])) + zp * t * (1 - t) * Mb * a_t
            q = a0 + g0(r0) + g1(t)
            msgs.append((p + rho * q + (1 - r0) * K * (1 - t)) % P)
        c = [(lerp(Vf[b], r1)[0] + z * Mb) % P for b, Mb in ((0, M0), (1, M1))]
        v = (rho * (a0 + g0(r0) + g1(r1)) + (1 - r0) * (1 - r1) * K) % P
        msgs += c + [v]
        # the verifier's checks hold for every choice of coins
        m0 = msgs[2:5]
        m1 = msgs[5:9]
        assert np.all((m0[0] + m0[1]) % P == (claim + rho * G) % P)
        at_r0 = (m0[0] * ((r0 - 1) * (r0 - 2) * pow(2, -1, P)) + m0[1] * (r0 * (r0 - 2) * pow(-1, -1, P)) + m0[2] * (r0 * (r0 - 1) * pow(2, -1, P))) % P
        assert np.all((m1[0] + m1[1]) % P == at_r0)
        at_r1 = 0
        for i in range(4):                                       # Lagrange through t = 0 .. 3
            num, den = 1, 1
            for k in range(4):
                if k != i:
                    num, den = num * (r1 - k), den * (i - k)
            at_r1 = at_r1 + m1[i] * (num * pow(den % P, -1, P) % P)
        Ar = [lerp(Af[b], r1)[0] for b in range(2)]
        assert np.all((at_r1 - v) % P == (Ar[0] * c[0] + Ar[1] * c[1]) % P)
        code = np.zeros(n, dtype=np.int64)
        for m in msgs:
            code = code * P + (m % P)
        return np.unique(code)
    w1 = view([[1, 2, 3, 4], [0, 1, 4, 2]])
    w2 = view([[3, 3, 0, 1], [2, 0, 0, 4]])
    assert len(w1) == n and len(w2) == n                         # injective: the view is uniform on its support
    assert np.array_equal(w1, w2)                                # ... and the support is the same for both witnesses


def test_every_message_depends_on_the_coins(oracle, monkeypatch):
    """the same challenges with other prover coins (test hook ZKCNN_TEST_COIN_SALT): every commitment, round polynomial, evaluation claim and
    revealed value changes -- all that stays is the public output's value Vres and claims about operands a layer does not have (zero). With
    unmasked claims (the first version of the mode) every claim_u / claim_v would be among the unchanged messages."""
    for model, pic, pp in CASES[1:]:
        with oracle_ffi.OracleSession(model, pic, pp) as o:
            monkeypatch.setenv("ZKCNN_TEST_COIN_SALT", "0")
            r1, t1 = o.prove(seed=21, mode=ZK | REUSE)
            _, t1b = o.prove(seed=21, mode=ZK | REUSE)
            monkeypatch.setenv("ZKCNN_TEST_COIN_SALT", "12345")
            r2, t2 = o.prove(seed=21, mode=ZK | REUSE)
            _, p2 = o.prove(seed=21, mode=REUSE)
            monkeypatch.setenv("ZKCNN_TEST_COIN_SALT", "0")
            _, p1 = o.prove(seed=21, mode=REUSE)
            assert r1.accepted == 1 and r2.accepted == 1 and t1 == t1b and len(t1) == len(t2)
            assert p1 == p2                                      # the plain protocol has no coins
            same = [k for k in range(0, len(t1), 16) if t1[k:k + 16] == t2[k:k + 16] and any(t1[k:k + 16])]      # (32- and 48-byte messages: 16-byte grid)
            assert len(same) <= 2, (model, len(same), len(t1) // 16)


def test_zero_knowledge_golden_hash(oracle):
    """regression vector of the mode (made by this repo's oracle: parity unpinned like every other vector here)"""
    with oracle_ffi.OracleSession(*CASES[1]) as o:
        _, tr = o.prove(seed=0x5EED0001, mode=ZK | REUSE)
    assert len(tr) == 36672 and hashlib.sha256(tr).hexdigest() == ZK_GOLDEN


ZK_GOLDEN = "eb87bddffa61a22837c7b9f12f9c205374048660b9c477c1b7476014c06cef2c"      # round 5: masked evaluation claims (was 411d4703.. / 36 432 bytes)
