"""Kernel-level parity on the GPU: every zk_k_* entry point of include/zkcnn_hip.h against the CPU oracle
(bit-exact: integer field arithmetic) and against Python big-int arithmetic."""
import numpy as np
import pytest

import zkcnn_amd
from zkcnn_amd import P_MOD, R_MOD, from_mont, to_mont

pytestmark = pytest.mark.gpu


def _rand(oracle, n, seed):
    return oracle.random(n, seed)


def test_fr_ops_match_oracle_and_bigint(hip, oracle):
    n = 5000
    a, b = _rand(oracle, n, 1), _rand(oracle, n, 2)
    # edge values: 0, 1, r-1 in every pairing
    edge = to_mont([0, 1, R_MOD - 1, 2, R_MOD - 2, (R_MOD - 1) // 2, (R_MOD + 1) // 2])
    a[:7], b[:7] = edge, edge[::-1]
    ai, bi = from_mont(a), from_mont(b)
    for op, f in (("mul", lambda x, y: x * y % R_MOD), ("add", lambda x, y: (x + y) % R_MOD), ("sub", lambda x, y: (x - y) % R_MOD)):
        g = hip.fr_binop(op, a, b)
        assert np.array_equal(g, oracle.binop(op, a, b)), op
        assert from_mont(g[:200]) == [f(x, y) for x, y in zip(ai[:200], bi[:200])], op


@pytest.mark.parametrize("n", [0, 1, 2, 5, 10, 13, 17])
def test_eq_table(hip, oracle, n):
    r0, r1 = _rand(oracle, max(n, 1), 10 + n), _rand(oracle, max(n, 1), 50 + n)
    al, be = _rand(oracle, 1, 3), _rand(oracle, 1, 4)
    zero = np.zeros((1, 4), dtype=np.uint64)
    for alpha, beta in ((al, be), (al, zero), (zero, be), (zero, zero)):
        g = hip.eq_table(n, r0, r1, alpha, beta)
        assert np.array_equal(g, oracle.eq_table2(n, r0, r1, alpha, beta))


@pytest.mark.parametrize("n", [2, 5, 9, 12])
@pytest.mark.parametrize("inverse", [False, True])
def test_phi_table_closed_form_equals_reference_recursion(hip, oracle, n, inverse):
    rx, scale = _rand(oracle, n, 77 + n), _rand(oracle, 1, 5)
    assert np.array_equal(hip.phi_table(rx, scale, n, inverse), oracle.phi_table(rx, scale, n, inverse))


@pytest.mark.parametrize("logn", [1, 2, 3, 8, 14, 18])
def test_round_quadratic_full_sumcheck(hip, oracle, logn):
    """run a whole sumcheck (first round + folds) on GPU and oracle, compare every round polynomial,
    check the round identity p(0) + p(1) == claim, and the final folded values"""
    n = 1 << logn
    V, M = _rand(oracle, n, 100 + logn), _rand(oracle, n, 200 + logn)
    if logn >= 3:       # zero tail, like a padded table
        V[n - n // 3:] = 0
    Vg, Mg, Vo, Mo = V.copy(), M.copy(), V.copy(), M.copy()
    rs = _rand(oracle, logn, 300 + logn)
    claim = sum(x * y for x, y in zip(from_mont(V), from_mont(M))) % R_MOD if logn <= 8 else None
    first, ng = True, n
    for k in range(logn):
        r = rs[max(k - 1, 0):max(k - 1, 0) + 1]
        cg, ng2 = hip.round_quadratic(Vg[:ng], Mg[:ng], r, first)
        co, no2 = oracle.round_quadratic(Vo[:ng], Mo[:ng], r, first)
        assert ng2 == no2
        assert np.array_equal(cg, co), f"round {k}"
        assert np.array_equal(Vg[:ng2], Vo[:ng2]) and np.array_equal(Mg[:ng2], Mo[:ng2])
        if claim is not None:
            a, b, c = from_mont(cg)
            assert (c + a + b + c) % R_MOD == claim
            x = from_mont(rs[k:k + 1])[0]
            claim = (a * x * x + b * x + c) % R_MOD
        first, ng = False, ng2


def test_microbench_report(hip):
    """not a pass/fail test: prints the ALU and HBM ceilings the roofline numbers are read against"""
    sec = hip.bench_fr_mul(1 << 20, 256, iters=3)
    rate = (1 << 20) * 256 / sec
    cp = hip.bench_copy(1 << 30, iters=10)
    rq, nbytes = hip.bench_round_quadratic(24, iters=10)
    print(f"\nfr_mul: {rate / 1e9:.1f} G mul/s; copy: {2 * (1 << 30) / cp / 1e9:.0f} GB/s; "
          f"round_quad 2^24: {rq * 1e3:.3f} ms = {nbytes / rq / 1e9:.0f} GB/s algorithmic")
    assert rate > 1e9


@pytest.mark.parametrize("n,kind", [(1, "rand"), (7, "rand"), (300, "rand"), (1024, "small"), (2048, "bits"), (513, "mixed")])
def test_msm_matches_cpu_pippenger(hip, oracle, n, kind):
    bases = oracle.generators(n, 1000 + n)
    if kind == "rand":
        sc = oracle.random(n, 5 + n)
    elif kind == "small":        # signed 9-bit quantised weights
        rng = np.random.default_rng(n)
        sc = to_mont([int(x) % R_MOD for x in rng.integers(-255, 256, n)])
    elif kind == "bits":
        rng = np.random.default_rng(n)
        sc = to_mont([int(x) for x in rng.integers(0, 2, n)])
    else:
        sc = oracle.random(n, 9)
        sc[::3] = 0
        sc[1::7] = to_mont([R_MOD - 1])[0]
        bases[5] = 0             # point at infinity among the bases
    want = oracle.msm(sc, bases)
    g = hip.msm(sc, bases)
    assert np.array_equal(g, want)
    assert oracle.g1_on_curve(g)
    # the same generators again: the context now builds the full byte table (one mixed addition per non-zero scalar byte,
    # 64-wide reduction trees, last partial sums on the host) -- same point
    assert np.array_equal(hip.msm(sc, bases), want)
    sc2 = oracle.random(n, 77 + n)
    assert np.array_equal(hip.msm(sc2, bases), oracle.msm(sc2, bases))


def test_fp_inverse_binary_gcd_against_python(hip):
    """the table kernels' field inversion (binary extended Euclid on the Montgomery representation, g1_dev.cuh: fp_inv) against pow(a, -1, p):
    edge values (0 -> 0, 1, p - 1, 2, powers of two, values just below p and around 2^380) and random elements"""
    P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
    RR = 1 << 384
    rng = np.random.default_rng(381)
    vals = [0, 1, 2, 3, P - 1, P - 2, (P - 1) // 2, (P + 1) // 2, 1 << 380, (1 << 380) - 1, (1 << 381) % P, P - (1 << 200)]
    vals += [1 << k for k in range(0, 380, 19)]
    vals += [int.from_bytes(rng.bytes(48), "little") % P for _ in range(700)]

    def pack(xs):
        a = np.zeros((len(xs), 6), dtype=np.uint64)
        for i, x in enumerate(xs):
            m = x * RR % P
            for k in range(6):
                a[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
        return a

    got = hip.fp_inv(pack(vals))
    want = pack([pow(v, -1, P) if v else 0 for v in vals])
    assert np.array_equal(got, want)
    # ... and on the Montgomery representation's own edge: elements whose REPRESENTATION aR mod p is 1, 2, p - 1 (the algorithm runs on it)
    reps = [1, 2, P - 1, P - 2, 1 << 380]
    a = np.zeros((len(reps), 6), dtype=np.uint64)
    for i, m in enumerate(reps):
        for k in range(6):
            a[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    inv_r = pow(RR, -1, P)
    want2 = pack([pow(m * inv_r % P, -1, P) for m in reps])
    assert np.array_equal(hip.fp_inv(a), want2)


def test_commit_rows_digit_table_and_high_byte_rows(hip, oracle):
    """the commitInput data path: signed-byte witnesses through the digit table, rows with larger scalars through the
    bit-plane path for their remaining windows; every row against the CPU Pippenger"""
    rows, cols = 12, 512
    rng = np.random.default_rng(7)
    vals = rng.integers(-255, 256, size=(rows, cols)).astype(object)
    vals[3, :] = rng.integers(0, 2, cols)                          # bit row
    vals[5, 10:40] = [int(x) for x in rng.integers(-(1 << 20), 1 << 20, 30)]   # a few wide scalars (biases / maxima)
    vals[7, :] = 0                                                 # empty row -> point at infinity
    sc = to_mont([int(v) % R_MOD for v in vals.reshape(-1)])
    big = oracle.random(cols, 99)
    sc[8 * cols:9 * cols] = big                                    # a full-width random row
    bases = oracle.generators(cols, 4242)
    got = hip.commit_rows(sc, bases, rows, cols)
    for r in range(rows):
        assert np.array_equal(got[r], oracle.msm(sc[r * cols:(r + 1) * cols], bases)), f"row {r}"
    assert not got[7].any()


@pytest.mark.parametrize("wide_max", [1 << 20, 1 << 63, None])
def test_commit_rows_more_wide_rows_than_the_virtual_row_list_holds(hip, oracle, wide_max):
    """> 128 rows with scalars beyond one byte: the first 128 ride along as virtual rows of the hot kernel (window sums through the digit table, shifted by
    k_cl_whorner), the rest take the separate pass over the window tables -- digit planes in non-adjacent form of the magnitude WITHOUT its low byte (that byte
    went through the digit table; 0xff = 0x100 - 1 must not be recoded across the two paths), with room for the top digit one bit above the magnitude
    (values below 2^64 reach bit 64). Low bytes of 0xff / 0x80 / 0x00 and values around 2^8, 2^16, 2^63, 2^64 in every row; first use of the generators
    (no byte table) and second use (byte table)."""
    rows, cols = 150, 128
    rng = np.random.default_rng(17)
    vals = np.zeros((rows, cols), dtype=object)
    edge = [0xff, 0x100, 0x1ff, 0xff00, 0xffff, 0x10000, 0x80, 0x8000, (1 << 63) - 1, (1 << 64) - 1, 0x7fff_ffff_ffff_ff00]
    for r in range(rows):
        if wide_max is None:
            row = [int.from_bytes(rng.bytes(31), "little") for _ in range(cols)]
        else:
            row = [int(x) for x in rng.integers(-wide_max, wide_max, cols, dtype=np.int64)]
            row[:len(edge)] = [e if e < 2 * wide_max else e % wide_max for e in edge]
            row[len(edge)] = -0xff
            row[len(edge) + 1] = -0x100
        vals[r, :] = row
    sc = to_mont([int(v) % R_MOD for v in vals.reshape(-1)])
    bases = oracle.generators(cols, 5151 + (0 if wide_max is None else wide_max % 97))
    for use in range(2):
        got = hip.commit_rows(sc, bases, rows, cols)
        for r in (0, 1, 63, 127, 128, 129, 149):
            assert np.array_equal(got[r], oracle.msm(sc[r * cols:(r + 1) * cols], bases)), f"use {use} row {r}"


def test_commit_rows_byte_table_path(hip, oracle):
    """many rows over generators that are used again: every row through the full byte table (no bit planes, no high-byte pass)"""
    rows, cols = 80, 128
    rng = np.random.default_rng(11)
    vals = rng.integers(-255, 256, size=(rows, cols)).astype(object)
    vals[2, :] = rng.integers(0, 2, cols)
    vals[4, 3:9] = [int(x) for x in rng.integers(-(1 << 40), 1 << 40, 6)]
    vals[6, :] = 0
    sc = to_mont([int(v) % R_MOD for v in vals.reshape(-1)])
    sc[9 * cols:10 * cols] = oracle.random(cols, 5)
    bases = oracle.generators(cols, 777)
    first = hip.commit_rows(sc, bases, rows, cols)          # first use of the generators: digit table + bit planes
    again = hip.commit_rows(sc, bases, rows, cols)          # second use: byte table
    assert np.array_equal(first, again)
    for r in range(rows):
        assert np.array_equal(again[r], oracle.msm(sc[r * cols:(r + 1) * cols], bases)), f"row {r}"


@pytest.mark.parametrize("logn", [1, 2, 5, 9, 12])
def test_batched_ntt_matches_reference_fft(hip, oracle, logn):
    """K9 against the oracle's restatement of reference src/utils.cpp:105-145, used as calcFFTLayer uses it:
    forward = zero-padded half-length inputs, inverse = scaled, first half kept; and inverse(forward(x)) == x"""
    length, count = 1 << logn, 7 if logn < 12 else 3
    half = length // 2
    x = oracle.random(count * half, 400 + logn)
    padded = np.zeros((count * length, 4), dtype=np.uint64)
    for c in range(count):
        padded[c * length:c * length + half] = x[c * half:(c + 1) * half]
    fwd = hip.witness_ntt(x, logn, False, count)
    assert np.array_equal(fwd, oracle.ntt(padded, logn, False))
    back = hip.witness_ntt(fwd, logn, True, count)
    full_back = oracle.ntt(fwd, logn, True)
    for c in range(count):
        assert np.array_equal(back[c * half:(c + 1) * half], full_back[c * length:c * length + half])
    assert np.array_equal(back, x)


def _gates_expected(n_out, uni, bin_, v0, vp, tm, scale):
    """calcNormalLayer (reference src/neuralNetwork.cpp:918-935) on Python integers"""
    out = [0] * n_out
    for gt in uni:
        x = (vp if gt["lu"] else v0)[gt["u"]]
        out[gt["g"]] = (out[gt["g"]] + x * tm[gt["sc"]]) % R_MOD
    for gt in bin_:
        x = (v0 if gt["l"] == 0 else vp)[gt["u"]]
        y = (vp if (gt["l"] & 1) else v0)[gt["v"]]
        out[gt["g"]] = (out[gt["g"]] + x * y * tm[gt["sc"]]) % R_MOD
    return [o * scale % R_MOD for o in out]


@pytest.mark.parametrize("case", ["grouped", "shuffled", "layer1", "uni_only", "long_runs"])
def test_witness_gates(hip, oracle, case):
    """zk_witness_gates: every gate of a generic layer, against big-int arithmetic; layer 0 uploaded in two overlapping spans"""
    rng = np.random.default_rng(7)
    n0, n_prev, n_out = 3000, 2000, 700
    v0m, vpm = _rand(oracle, n0, 71), _rand(oracle, n_prev, 72)
    q = 20
    tm = [pow(2, k, R_MOD) for k in range(q + 1)] + [(-pow(2, k, R_MOD)) % R_MOD for k in range(q + 1)]
    tmm = to_mont(tm)
    scale = 1 if case in ("grouped", "long_runs") else pow(49, -1, R_MOD)
    if case == "long_runs":                 # segments longer than a block (conv: thousands of products per output), outputs without gates
        gs = np.sort(rng.choice(np.array([3, 4, 5, 250, 699]), size=6000))
    else:
        gs = np.sort(rng.integers(0, n_out, size=6000))
    n_uni = 0 if case == "layer1" else 2500
    uni = np.zeros(n_uni, dtype=zkcnn_amd.HipContext.UNI_GATE)
    uni["g"] = np.sort(rng.integers(0, n_out, size=n_uni)) if case != "long_runs" else np.sort(rng.choice(np.array([3, 5, 699]), size=n_uni))
    uni["lu"] = rng.integers(0, 2, size=n_uni) * 9
    uni["u"] = [rng.integers(0, n_prev if lu else n0) for lu in uni["lu"]]
    uni["sc"] = rng.integers(0, len(tm), size=n_uni) * (rng.integers(0, 2, size=n_uni))
    n_bin = 0 if case == "uni_only" else gs.shape[0]
    bin_ = np.zeros(n_bin, dtype=zkcnn_amd.HipContext.BIN_GATE)
    bin_["g"] = gs[:n_bin]
    bin_["l"] = rng.integers(0, 3, size=n_bin)
    bin_["u"] = [rng.integers(0, n0 if l == 0 else n_prev) for l in bin_["l"]]
    bin_["v"] = [rng.integers(0, n_prev if (l & 1) else n0) for l in bin_["l"]]
    bin_["sc"] = rng.integers(0, len(tm), size=n_bin) * (rng.integers(0, 2, size=n_bin))
    if case == "shuffled":                  # equal outputs no longer adjacent: the entry point regroups
        bin_ = bin_[rng.permutation(n_bin)]
        uni = uni[rng.permutation(n_uni)]
    prev = vpm
    if case == "layer1":                    # previous layer == layer 0 (prev = NULL)
        prev, n_prev_eff = None, n0
        bin_["u"] %= n0
        bin_["v"] %= n0
        vp_int = from_mont(v0m)
    else:
        vp_int = from_mont(vpm)
    hip.witness_input(0, v0m[:2000])
    hip.witness_input(1500, v0m[1500:])     # overlapping second span, as after a resize of layer 0
    got = hip.witness_gates(n_out, uni, bin_, prev, tmm, to_mont([scale]))
    want = _gates_expected(n_out, uni, bin_, from_mont(v0m), vp_int, tm, scale)
    assert from_mont(got) == want


def test_witness_gates_rejects_out_of_range(hip, oracle):
    v0m = _rand(oracle, 10, 5)
    hip.witness_input(0, v0m)
    uni = np.zeros(1, dtype=zkcnn_amd.HipContext.UNI_GATE)
    uni["u"] = 10                            # one past the end of layer 0
    with pytest.raises(RuntimeError):
        hip.witness_gates(4, uni, np.zeros(0, dtype=zkcnn_amd.HipContext.BIN_GATE), None, to_mont([1]), to_mont([1]))


# ---- row-cooperative base-field / curve arithmetic (hip/fpc_dev.cuh: one element per 16-lane DPP row; the arithmetic of the commitment's reduction trees) ----
_RR = 1 << 384


def _fp_pack(xs):
    a = np.zeros((len(xs), 6), dtype=np.uint64)
    for i, x in enumerate(xs):
        m = x * _RR % P_MOD
        for k in range(6):
            a[i, k] = (m >> (64 * k)) & 0xFFFFFFFFFFFFFFFF
    return a


def _fp_unpack(a):
    inv = pow(_RR, -1, P_MOD)
    return [sum(int(row[k]) << (64 * k) for k in range(6)) * inv % P_MOD for row in a.reshape(-1, 6)]


def _jac(points, zs):
    """Jacobian (X z^2, Y z^3, z) of affine integer points (None = infinity), 18 words each"""
    flat = []
    for p, z in zip(points, zs):
        flat += [0, 1, 0] if p is None else [p[0] * z * z % P_MOD, p[1] * z * z * z % P_MOD, z]
    return _fp_pack(flat).reshape(-1, 18)


def _affine(j):
    out = []
    for X, Y, Z in zip(*[iter(_fp_unpack(j))] * 3):
        if Z == 0:
            out.append(None)
        else:
            zi = pow(Z, -1, P_MOD)
            out.append((X * zi * zi % P_MOD, Y * zi * zi * zi % P_MOD))
    return out


def test_row_cooperative_field_arithmetic_against_python(hip):
    """products, sums and differences of the 16-lanes-per-element form against Python integers: 0, 1, p - 1, equal operands, operands whose sum is
    exactly p, limbs of all ones (carries that ripple through a whole row), and random elements"""
    rng = np.random.default_rng(950)
    ones = (1 << 352) - 1
    a = [0, 1, P_MOD - 1, P_MOD - 1, 5, P_MOD - 5, ones, ones, (1 << 380), P_MOD - 1 - ones, 0xffffffff, 1 << 32]
    b = [7, P_MOD - 1, P_MOD - 1, 1, 5, 5, 1, ones, (1 << 380), ones + 1, 0xffffffff, (1 << 352) - (1 << 32)]
    a += [int.from_bytes(rng.bytes(48), "little") % P_MOD for _ in range(2000)]
    b += [int.from_bytes(rng.bytes(48), "little") % P_MOD for _ in range(2000)]
    prod, tot, dif = hip.fpc_ops(_fp_pack(a), _fp_pack(b))
    assert _fp_unpack(prod) == [x * y % P_MOD for x, y in zip(a, b)]
    assert _fp_unpack(tot) == [(x + y) % P_MOD for x, y in zip(a, b)]
    assert _fp_unpack(dif) == [(x - y) % P_MOD for x, y in zip(a, b)]
    # canonical outputs (< p): the row form's results feed equality tests and the transcript
    for arr in (prod, tot, dif):
        assert all(sum(int(r[k]) << (64 * k) for k in range(6)) < P_MOD for r in arr)


def test_row_cooperative_point_arithmetic_every_case(hip, oracle):
    """general addition and doubling in row form against affine arithmetic with Python integers: generic pairs in different Jacobian scalings, either
    or both operands at infinity, P = Q in the same and in different coordinates (the doubling branch), P = -Q"""
    from tests.test_hyrax_cpu import _padd
    rng = np.random.default_rng(951)
    n = 96
    g = oracle.generators(2 * n, 4242)
    aff = [tuple(_fp_unpack(row.reshape(2, 6))) for row in g]
    P, Q = aff[:n], aff[n:]
    P[0] = None
    Q[1] = None
    P[2] = Q[2] = None
    Q[3] = P[3]
    Q[4] = P[4]
    Q[5] = (P[5][0], (P_MOD - P[5][1]) % P_MOD)
    Q[6] = (P[6][0], (P_MOD - P[6][1]) % P_MOD)
    zp = [1] * n
    zq = [1] * n
    for i in range(n):
        if i != 3 and i != 5 and i % 2 == 0:
            zp[i] = int.from_bytes(rng.bytes(47), "little") % P_MOD or 1
            zq[i] = int.from_bytes(rng.bytes(47), "little") % P_MOD or 1
    got_sum, got_dbl = hip.cl_add(_jac(P, zp), _jac(Q, zq))
    assert _affine(got_sum) == [_padd(p, q) for p, q in zip(P, Q)]
    assert _affine(got_dbl) == [_padd(p, p) for p in P]


@pytest.mark.parametrize("n_seg,segs,n_in", [(1, 7, 64), (5, 9, 64), (64, 4, 64), (100, 3, 64), (700, 2, 64), (129, 2, 32)])
def test_row_cooperative_reduction_tree(hip, oracle, n_seg, segs, n_in):
    """k_cl_tree: sums of runs of points (ragged last run, a point at infinity and a repeated point among them) against a Python chain of additions"""
    from tests.test_hyrax_cpu import _padd
    g = oracle.generators(n_seg * segs, 77 + n_seg)
    aff = [tuple(_fp_unpack(row.reshape(2, 6))) for row in g]
    if len(aff) > 4:
        aff[2] = None
        aff[4] = aff[3]
    got = _affine(hip.cl_tree(_jac(aff, [1 + 3 * i for i in range(len(aff))]), segs, n_in))
    want = []
    for s in range(segs):
        acc = None
        for p in aff[s * n_seg:(s + 1) * n_seg]:
            acc = _padd(acc, p)
        want.append(acc)
    assert got == want
