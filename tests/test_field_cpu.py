"""Pins the CPU field / curve library and the oracle's table algorithms against Python big-int arithmetic
and against the mathematical definitions (the reference ships no vectors for this path: SURVEY.md 8(c))."""
import random

import numpy as np
import pytest

from zkcnn_amd import P_MOD, R_MOD, from_limbs, from_mont, to_limbs, to_mont


def test_constants():
    assert R_MOD.bit_length() == 255 and (R_MOD - 1) % (1 << 32) == 0 and (R_MOD - 1) % (1 << 33) != 0
    assert pow(5, (R_MOD - 1) // 2, R_MOD) == R_MOD - 1          # 5 is a quadratic non-residue
    assert P_MOD.bit_length() == 381 and P_MOD % 4 == 3


def test_field_ops_vs_bigint(oracle):
    random.seed(1)
    xs = [random.randrange(R_MOD) for _ in range(200)] + [0, 1, R_MOD - 1, 2, (R_MOD - 1) // 2]
    ys = [random.randrange(R_MOD) for _ in range(200)] + [R_MOD - 1, R_MOD - 1, R_MOD - 1, 0, (R_MOD + 1) // 2]
    X, Y = oracle.from_canonical(to_limbs(xs)), oracle.from_canonical(to_limbs(ys))
    assert from_limbs(X) == [x * (1 << 256) % R_MOD for x in xs]       # Montgomery form, R = 2^256
    assert np.array_equal(X, to_mont(xs))
    for op, f in (("mul", lambda a, b: a * b % R_MOD), ("add", lambda a, b: (a + b) % R_MOD), ("sub", lambda a, b: (a - b) % R_MOD)):
        assert from_mont(oracle.binop(op, X, Y)) == [f(a, b) for a, b in zip(xs, ys)], op
    assert from_mont(oracle.inv(X)) == [pow(a, -1, R_MOD) if a else 0 for a in xs]


def test_root_of_unity_is_primitive(oracle):
    for n in (1, 2, 5, 9, 12, 20, 32):
        w = from_mont(oracle.root_of_unity(n))[0]
        assert pow(w, 1 << (n - 1), R_MOD) == R_MOD - 1
    assert from_mont(oracle.root_of_unity(0))[0] == 1


def _eq(r, i):
    acc = 1
    for j, rj in enumerate(r):
        acc = acc * (rj if (i >> j) & 1 else (1 - rj)) % R_MOD
    return acc


@pytest.mark.parametrize("n", [0, 1, 4, 7])
def test_eq_tables_match_definition(oracle, n):
    random.seed(n)
    r0 = [random.randrange(R_MOD) for _ in range(max(n, 1))]
    r1 = [random.randrange(R_MOD) for _ in range(max(n, 1))]
    a, b = random.randrange(R_MOD), random.randrange(R_MOD)
    t = from_mont(oracle.eq_table2(n, to_mont(r0), to_mont(r1), to_mont([a]), to_mont([b])))
    assert t == [(a * _eq(r0[:n], i) + b * _eq(r1[:n], i)) % R_MOD for i in range(1 << n)]
    assert sum(t) % R_MOD == (a + b) % R_MOD                       # sum_i eq(r, i) = 1
    t1 = from_mont(oracle.eq_table1(n, to_mont(r0), to_mont([a])))
    assert t1 == [a * _eq(r0[:n], i) % R_MOD for i in range(1 << n)]


@pytest.mark.parametrize("n", [2, 3, 6])
@pytest.mark.parametrize("inverse", [False, True])
def test_phi_table_is_mle_of_dft_matrix(oracle, n, inverse):
    """phi[u] = scale * sum_g eq(rx, g) w^{+-g u}, g over n (forward) or n-1 (inverse) bits"""
    random.seed(10 * n + inverse)
    w = from_mont(oracle.root_of_unity(n))[0]
    if inverse:
        w = pow(w, -1, R_MOD)
    rx = [random.randrange(R_MOD) for _ in range(n)]
    scale = random.randrange(R_MOD)
    got = from_mont(oracle.phi_table(to_mont(rx), to_mont([scale]), n, inverse))
    gbits = n - 1 if inverse else n
    cnt = (1 << n) if inverse else (1 << (n - 1))
    want = [scale * sum(_eq(rx[:gbits], g) * pow(w, g * u, R_MOD) for g in range(1 << gbits)) % R_MOD for u in range(cnt)]
    assert got == want


@pytest.mark.parametrize("logn", [1, 3, 6])
def test_ntt_is_dft_and_inverts(oracle, logn):
    random.seed(logn)
    n = 1 << logn
    xs = [random.randrange(R_MOD) for _ in range(n)]
    w = from_mont(oracle.root_of_unity(logn))[0]
    fwd = oracle.ntt(to_mont(xs), logn, False)
    assert from_mont(fwd) == [sum(x * pow(w, j * k, R_MOD) for j, x in enumerate(xs)) % R_MOD for k in range(n)]
    assert from_mont(oracle.ntt(fwd, logn, True)) == xs


def test_ntt_convolution_equals_direct_convolution(oracle):
    """the idea behind the reference's FFT conv layers (reference src/models.cpp:305-375 compares timings only)"""
    random.seed(5)
    logn, n = 5, 32
    a = [random.randrange(-255, 256) % R_MOD for _ in range(n // 2)] + [0] * (n // 2)
    b = [random.randrange(-255, 256) % R_MOD for _ in range(n // 2)] + [0] * (n // 2)
    fa, fb = from_mont(oracle.ntt(to_mont(a), logn, False)), from_mont(oracle.ntt(to_mont(b), logn, False))
    prod = from_mont(oracle.ntt(to_mont([x * y % R_MOD for x, y in zip(fa, fb)]), logn, True))
    direct = [0] * n
    for i in range(n // 2):
        for j in range(n // 2):
            direct[i + j] = (direct[i + j] + a[i] * b[j]) % R_MOD
    assert prod == direct


def _fp(a):
    return [sum(int(a[i, j]) << (64 * j) for j in range(6)) for i in range(a.shape[0])]


def _padd(P1, P2):
    if P1 is None:
        return P2
    if P2 is None:
        return P1
    (x1, y1), (x2, y2) = P1, P2
    if x1 == x2 and (y1 + y2) % P_MOD == 0:
        return None
    lam = 3 * x1 * x1 * pow(2 * y1, -1, P_MOD) % P_MOD if P1 == P2 else (y2 - y1) * pow(x2 - x1, -1, P_MOD) % P_MOD
    x3 = (lam * lam - x1 - x2) % P_MOD
    return x3, (lam * (x1 - x3) - y1) % P_MOD


def _pmul(k, Pt):
    acc = None
    while k:
        if k & 1:
            acc = _padd(acc, Pt)
        Pt = _padd(Pt, Pt)
        k >>= 1
    return acc


def test_curve_vs_python_affine_arithmetic(oracle):
    random.seed(3)
    g = oracle.g1_base()
    gx, gy = _fp(oracle.fp_to_canonical(g))
    assert gx == 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb
    assert (gy * gy - gx ** 3 - 4) % P_MOD == 0
    k = random.randrange(R_MOD)
    assert tuple(_fp(oracle.fp_to_canonical(oracle.g1_mul(g, to_mont([k]))))) == _pmul(k, (gx, gy))
    neg = oracle.g1_mul(g, to_mont([R_MOD - 1]))                        # (r-1) G = -G  =>  r G = O
    assert _fp(oracle.fp_to_canonical(neg)) == [gx, P_MOD - gy]
    assert not oracle.g1_add(neg, g).any()
    # compressed serialisation: 48 bytes, big-endian x, flag bits
    ser = oracle.g1_serialize(g)
    assert len(ser) == 48 and ser[0] & 0x80 and int.from_bytes(bytes([ser[0] & 0x1f]) + ser[1:], "big") == gx
    assert bool(ser[0] & 0x20) == (gy > P_MOD - gy)
    assert oracle.g1_serialize(np.zeros(12, dtype=np.uint64))[0] == 0xc0


def test_msm_vs_python(oracle):
    random.seed(4)
    n = 24
    gens = oracle.generators(n, 7)
    pts = [tuple(_fp(oracle.fp_to_canonical(gens[i]))) for i in range(n)]
    sc = [random.randrange(R_MOD) for _ in range(n - 6)] + [0, 1, R_MOD - 1, R_MOD - 255, 255, 1 << 200]
    got = tuple(_fp(oracle.fp_to_canonical(oracle.msm(to_mont(sc), gens))))
    acc = None
    for k, pt in zip(sc, pts):
        acc = _padd(acc, _pmul(k, pt))
    assert got == acc


def test_dot_prod_layer_identities_of_design_4i():
    """The three identities the GPU's DOT_PROD phase 1 relies on, against the definition of the reference's loop
    (src/prover.cpp:86-91 builds X, :103-144 the cubic rounds) in Python integers:
      (1) X[(p CI + ci, t)] = sum_co beta_g[p CO + co] F[((pp + co) CI + ci, t)] = beta_hi[p] S[ci, t] when beta_g = alpha eq(r, .) and CO = 2^k;
      (2) behind X's live prefix a round's pairs add nothing, and folding X there gives zeros;
      (3) once the periodic table is a scalar m, the round polynomial is m times the quadratic of (X, Y) with cubic coefficient zero."""
    rnd = random.Random(41)
    pp, CO, CI, fb = 3, 4, 2, 2
    ln, k = 1 << fb, 2
    gbits = 4                                             # pp CO = 12 outputs, padded to 16
    r = [rnd.randrange(R_MOD) for _ in range(gbits)]
    alpha = rnd.randrange(R_MOD)
    beta_g = [alpha * _eq(r, g) % R_MOD for g in range(1 << gbits)]
    rows = (pp + CO) * CI
    F = [[rnd.randrange(R_MOD) for _ in range(ln)] for _ in range(rows)]
    X = [[0] * ln for _ in range(rows)]
    for p in range(pp):
        for co in range(CO):
            for ci in range(CI):
                g, u, v = p * CO + co, p * CI + ci, (pp + co) * CI + ci
                for t in range(ln):
                    X[u][t] = (X[u][t] + beta_g[g] * F[v][t]) % R_MOD
    beta_lo = [_eq(r[:k], co) for co in range(CO)]
    beta_hi = [alpha * _eq(r[k:], p) % R_MOD for p in range(1 << (gbits - k))]
    S = [[sum(beta_lo[co] * F[(pp + co) * CI + ci][t] for co in range(CO)) % R_MOD for t in range(ln)] for ci in range(CI)]
    for p in range(pp):
        for ci in range(CI):
            assert X[p * CI + ci] == [beta_hi[p] * S[ci][t] % R_MOD for t in range(ln)]
    assert all(x == 0 for row in X[pp * CI:] for x in row)            # the weight vectors' rows have no gate: X's live prefix ends at pp CI rows
    # flat tables, padded to a power of two; Y = the FFT layer itself
    N = 1
    while N < rows * ln:
        N <<= 1
    xs = [x for row in X for x in row] + [0] * (N - rows * ln)
    ys = [y for row in F for y in row] + [0] * (N - rows * ln)
    ms = [rnd.randrange(R_MOD) for _ in range(ln)]                    # periodic in t

    def lerp(a, b, c):
        return (a + c * (b - a)) % R_MOD

    def cubic(xs, ys, ms):
        """coefficients (c3, c2, c1, c0) of sum_i X_i(x) Y_i(x) M_{i mod |M|/2}(x) over the pairs (2i, 2i + 1)"""
        c = [0, 0, 0, 0]
        mp = max(len(ms) // 2, 1)
        for i in range(len(xs) // 2):
            x0, dx, y0, dy = xs[2 * i], xs[2 * i + 1] - xs[2 * i], ys[2 * i], ys[2 * i + 1] - ys[2 * i]
            m0, dm = (ms[2 * (i % mp)], ms[2 * (i % mp) + 1] - ms[2 * (i % mp)]) if len(ms) > 1 else (ms[0], 0)
            q2, q1, q0 = dx * dy, x0 * dy + dx * y0, x0 * y0
            c[0] += q2 * dm; c[1] += q2 * m0 + q1 * dm; c[2] += q1 * m0 + q0 * dm; c[3] += q0 * m0
        return [v % R_MOD for v in c]

    live = pp * CI * ln
    while len(xs) > 2:
        full = cubic(xs, ys, ms)
        pl = (live + 1) // 2
        assert full == cubic(xs[:2 * pl], ys[:2 * pl], ms)             # (2) pairs behind the live prefix add nothing
        if len(ms) == 1:                                               # (3) a scalar m: m times the quadratic, no cubic term
            a = sum((xs[2 * i + 1] - xs[2 * i]) * (ys[2 * i + 1] - ys[2 * i]) for i in range(len(xs) // 2)) % R_MOD
            c0 = sum(xs[2 * i] * ys[2 * i] for i in range(len(xs) // 2)) % R_MOD
            p1 = sum(xs[2 * i + 1] * ys[2 * i + 1] for i in range(len(xs) // 2)) % R_MOD
            assert full == [0, ms[0] * a % R_MOD, ms[0] * (p1 - a - c0) % R_MOD, ms[0] * c0 % R_MOD]
        c = rnd.randrange(R_MOD)
        xs = [lerp(xs[2 * i], xs[2 * i + 1], c) for i in range(len(xs) // 2)]
        ys = [lerp(ys[2 * i], ys[2 * i + 1], c) for i in range(len(ys) // 2)]
        if len(ms) > 1:
            ms = [lerp(ms[2 * i], ms[2 * i + 1], c) for i in range(len(ms) // 2)]
        live = (live + 1) // 2
        assert all(x == 0 for x in xs[live:])                          # (2) the fold of zeros
