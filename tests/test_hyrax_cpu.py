"""The polynomial commitment on its own, re-verified with Python integers (review r4, weak #1: "not independent: the Hyrax verifier").

`zkcnn_amd/csrc/hyrax-bls12-381/polyCommit.hpp` is a from-scratch implementation of the surface the reference's call sites need (the upstream
submodule is absent: SURVEY.md Appendix B), and prover, verifier and oracle all share it -- so this file restates the protocol from its
definition: row commitments, the evaluation as a vector-matrix-vector product, every round of the inner-product argument (the prover's
cross terms AND the verifier's folding), the final check, and the zero-knowledge variant's proof of dot product; curve arithmetic, point
decompression and eq tables are Python's own. Nothing here calls the C++ verifier for the verdict it asserts."""
import ctypes

import pytest

from zkcnn_amd import P_MOD, R_MOD, from_mont, to_mont, u64p


def _padd(A, B):
    if A is None:
        return B
    if B is None:
        return A
    (x1, y1), (x2, y2) = A, B
    if x1 == x2 and (y1 + y2) % P_MOD == 0:
        return None
    lam = 3 * x1 * x1 * pow(2 * y1, -1, P_MOD) % P_MOD if A == B else (y2 - y1) * pow(x2 - x1, -1, P_MOD) % P_MOD
    x3 = (lam * lam - x1 - x2) % P_MOD
    return x3, (lam * (x1 - x3) - y1) % P_MOD


def _jdbl(P):
    """Jacobian doubling on y^2 = x^3 + 4 (a = 0); None = infinity"""
    if P is None:
        return None
    X, Y, Z = P
    if Y == 0:
        return None
    A, B = X * X % P_MOD, Y * Y % P_MOD
    C = B * B % P_MOD
    D = 2 * ((X + B) * (X + B) - A - C) % P_MOD
    E = 3 * A % P_MOD
    X3 = (E * E - 2 * D) % P_MOD
    return X3, (E * (D - X3) - 8 * C) % P_MOD, 2 * Y * Z % P_MOD


def _jadd_affine(P, Q):
    """Jacobian + affine (mixed addition), every special case through the general routines"""
    if P is None:
        return Q[0], Q[1], 1
    X1, Y1, Z1 = P
    x2, y2 = Q
    Z1Z1 = Z1 * Z1 % P_MOD
    U2, S2 = x2 * Z1Z1 % P_MOD, y2 * Z1 % P_MOD * Z1Z1 % P_MOD
    if U2 == X1:
        return _jdbl(P) if S2 == Y1 else None
    H, Rr = (U2 - X1) % P_MOD, (S2 - Y1) % P_MOD
    HH = H * H % P_MOD
    HHH, V = H * HH % P_MOD, X1 * HH % P_MOD
    X3 = (Rr * Rr - HHH - 2 * V) % P_MOD
    return X3, (Rr * (V - X3) - Y1 * HHH) % P_MOD, Z1 * H % P_MOD


def _pmul(k, A):
    """k A by double-and-add on Jacobian coordinates, one inversion at the end (the affine _padd costs one per step)"""
    k %= R_MOD
    if A is None or k == 0:
        return None
    acc = None
    for bit in bin(k)[2:]:
        acc = _jdbl(acc)
        if bit == "1":
            acc = _jadd_affine(acc, A)
    if acc is None:
        return None
    X, Y, Z = acc
    zi = pow(Z, -1, P_MOD)
    zi2 = zi * zi % P_MOD
    return X * zi2 % P_MOD, Y * zi2 % P_MOD * zi % P_MOD


def _msm(ks, pts):
    acc = None
    for k, pt in zip(ks, pts):
        acc = _padd(acc, _pmul(k, pt))
    return acc


def _decompress(b):
    """48 bytes, big-endian x; bit 7 = compressed, bit 6 = infinity, bit 5 = the larger of the two roots (the published BLS12-381 encoding)"""
    assert b[0] & 0x80
    if b[0] & 0x40:
        assert b == bytes([0xc0]) + bytes(47)              # the one encoding of the point at infinity
        return None
    x = int.from_bytes(bytes([b[0] & 0x1f]) + b[1:], "big")
    assert x < P_MOD
    y = pow((x ** 3 + 4) % P_MOD, (P_MOD + 1) // 4, P_MOD)
    assert (y * y - x ** 3 - 4) % P_MOD == 0
    if bool(b[0] & 0x20) != (y > P_MOD - y):
        y = P_MOD - y
    return x, y


def _eq(r):
    """eq table, variable j = bit j of the index"""
    t = [1]
    for x in r:
        t = [v * (1 - x) % R_MOD for v in t] + [v * x % R_MOD for v in t]
    return t


def _fr(b):
    v = int.from_bytes(b, "little")
    assert v < R_MOD
    return v


class _Reader:
    def __init__(self, data):
        self.d, self.o = data, 0

    def g1(self):
        self.o += 48
        return _decompress(self.d[self.o - 48:self.o])

    def fr(self):
        self.o += 32
        return _fr(self.d[self.o - 32:self.o])

    def done(self):
        return self.o == len(self.d)


def _run(oracle, Z, n, gens_mont, blinds, x, ev, chal_seed, stop_len, tamper_at=-1, coin_seed=77):
    fn = oracle.lib.oracle_hyrax_run
    fn.restype = ctypes.c_int32
    out = (ctypes.c_uint8 * (1 << 16))()
    ln = ctypes.c_uint64(0)
    zm, xm, em = to_mont(Z), to_mont(x), to_mont([ev])
    bm = to_mont(blinds) if blinds is not None else None
    ok = fn(u64p(zm), ctypes.c_int32(n), u64p(gens_mont), ctypes.c_uint64(gens_mont.shape[0]), u64p(bm) if bm is not None else None, u64p(xm), u64p(em),
            ctypes.c_uint64(chal_seed), ctypes.c_uint64(coin_seed), ctypes.c_int32(stop_len), ctypes.c_int32(tamper_at), out, ctypes.c_uint64(len(out)), ctypes.byref(ln))
    return ok, bytes(out[:ln.value])


def _points(oracle, gens_mont):
    c = oracle.fp_to_canonical(gens_mont)
    return [(sum(int(c[2 * i, j]) << (64 * j) for j in range(6)), sum(int(c[2 * i + 1, j]) << (64 * j) for j in range(6))) for i in range(gens_mont.shape[0])]


@pytest.mark.parametrize("n,stop_len", [(6, 1), (5, 2), (4, 256)])
def test_commitment_and_inner_product_argument_from_the_definition(oracle, n, stop_len):
    import random
    rnd = random.Random(100 + n)
    rb, cb = n >> 1, n - (n >> 1)
    rows, cols = 1 << rb, 1 << cb
    Z = [rnd.randrange(R_MOD) if rnd.random() < 0.7 else rnd.randrange(-255, 256) % R_MOD for _ in range(1 << n)]
    x = [rnd.randrange(R_MOD) for _ in range(n)]
    gens_mont = oracle.generators(cols, 5 + n)
    g = _points(oracle, gens_mont)
    L, b = _eq(x[cb:]), _eq(x[:cb])
    w = [sum(L[i] * Z[i * cols + j] for i in range(rows)) % R_MOD for j in range(cols)]
    ev = sum(wj * bj for wj, bj in zip(w, b)) % R_MOD
    # the multilinear extension of Z at x IS L^T Z R (column bits are the low bits)
    full = _eq(x)
    assert ev == sum(f * z for f, z in zip(full, Z)) % R_MOD
    ok, tr = _run(oracle, Z, n, gens_mont, None, x, ev, chal_seed=900 + n, stop_len=stop_len)
    assert ok == 1
    rd = _Reader(tr)
    C = [rd.g1() for _ in range(rows)]
    for i in range(rows):
        assert C[i] == _msm(Z[i * cols:(i + 1) * cols], g), f"row {i}: commitment is not <Z_i, g>"
    P, y = _msm(L, C), ev
    rounds = 0
    while (cols >> rounds) > stop_len and (cols >> rounds) > 1:
        rounds += 1
    cs = from_mont(oracle.random(max(rounds, 1), 900 + n))[:rounds]
    a, gg, bb = list(w), list(g), list(b)
    for t in range(rounds):
        h = len(a) // 2
        Lk, Rk, yL, yR = rd.g1(), rd.g1(), rd.fr(), rd.fr()
        # the prover's cross terms, from the definition
        assert Lk == _msm(a[:h], gg[h:]) and Rk == _msm(a[h:], gg[:h])
        assert yL == sum(p * q for p, q in zip(a[:h], bb[h:])) % R_MOD and yR == sum(p * q for p, q in zip(a[h:], bb[:h])) % R_MOD
        c = cs[t]
        # the verifier's fold ...
        P = _padd(_padd(Lk, _pmul(c, P)), _pmul(c * c % R_MOD, Rk))
        y = (yL + c * y + c * c * yR) % R_MOD
        # ... matches the folded instance
        a = [(a[j] + c * a[j + h]) % R_MOD for j in range(h)]
        bb = [(c * bb[j] + bb[j + h]) % R_MOD for j in range(h)]
        gg = [_padd(_pmul(c, gg[j]), gg[j + h]) for j in range(h)]
        assert P == _msm(a, gg) and y == sum(p * q for p, q in zip(a, bb)) % R_MOD
    fin = [rd.fr() for _ in range(len(a))]
    assert rd.done() and fin == a
    assert P == _msm(fin, gg) and y == sum(p * q for p, q in zip(fin, bb)) % R_MOD
    # a wrong evaluation and a corrupted message are refused by the C++ verifier the product uses
    assert _run(oracle, Z, n, gens_mont, None, x, (ev + 1) % R_MOD, 900 + n, stop_len)[0] == 0
    for k in range(rounds + 1):
        assert _run(oracle, Z, n, gens_mont, None, x, ev, 900 + n, stop_len, tamper_at=k)[0] == 0


@pytest.mark.parametrize("n", [4, 6])
def test_blinded_commitment_and_proof_of_dot_product_from_the_definition(oracle, n):
    import random
    rnd = random.Random(200 + n)
    rb, cb = n >> 1, n - (n >> 1)
    rows, cols = 1 << rb, 1 << cb
    Z = [rnd.randrange(R_MOD) for _ in range(1 << n)]
    x = [rnd.randrange(R_MOD) for _ in range(n)]
    blinds = [rnd.randrange(R_MOD) for _ in range(rows)]
    gens_mont = oracle.generators(cols + 1, 50 + n)              # g_0 .. g_{m-1}, H
    pts = _points(oracle, gens_mont)
    g, H = pts[:cols], pts[cols]
    L, b = _eq(x[cb:]), _eq(x[:cb])
    w = [sum(L[i] * Z[i * cols + j] for i in range(rows)) % R_MOD for j in range(cols)]
    ev = sum(wj * bj for wj, bj in zip(w, b)) % R_MOD
    ok, tr = _run(oracle, Z, n, gens_mont, blinds, x, ev, chal_seed=333 + n, stop_len=1)
    assert ok == 1
    rd = _Reader(tr)
    C = [rd.g1() for _ in range(rows)]
    for i in range(rows):
        assert C[i] == _padd(_msm(Z[i * cols:(i + 1) * cols], g), _pmul(blinds[i], H)), f"row {i}: not <Z_i, g> + s_i H"
    P = _msm(L, C)
    s_w = sum(l * s for l, s in zip(L, blinds)) % R_MOD
    assert P == _padd(_msm(w, g), _pmul(s_w, H))                 # the combined row is a commitment to w under the combined blind
    delta, t = rd.g1(), rd.fr()
    c = from_mont(oracle.random(1, 333 + n))[0]
    z = [rd.fr() for _ in range(cols)]
    z_s = rd.fr()
    assert rd.done()
    # the verifier's two checks
    assert _padd(_msm(z, g), _pmul(z_s, H)) == _padd(_pmul(c, P), delta)
    assert sum(p * q for p, q in zip(z, b)) % R_MOD == (c * ev + t) % R_MOD
    # ... and the honest prover's messages: z = c w + d with delta = Com(d; s_d), t = <d, b>
    d = [(zj - c * wj) % R_MOD for zj, wj in zip(z, w)]
    s_d = (z_s - c * s_w) % R_MOD
    assert delta == _padd(_msm(d, g), _pmul(s_d, H)) and t == sum(p * q for p, q in zip(d, b)) % R_MOD
    # other coins: another d, another delta -- the response is a fresh uniform vector every time
    _, tr2 = _run(oracle, Z, n, gens_mont, blinds, x, ev, chal_seed=333 + n, stop_len=1, coin_seed=78)
    assert tr2[:48 * rows] == tr[:48 * rows] and tr2[48 * rows:] != tr[48 * rows:]
    assert _run(oracle, Z, n, gens_mont, blinds, x, (ev + 1) % R_MOD, 333 + n, 1)[0] == 0
    for k in range(3):
        assert _run(oracle, Z, n, gens_mont, blinds, x, ev, 333 + n, 1, tamper_at=k)[0] == 0
