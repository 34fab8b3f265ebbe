"""An independent verifier, in Python, for a real zero-knowledge transcript (review r4, weak #1: prover, oracle and verifier share `host/verifier.hpp`,
`host/zk_mask.hpp` and `hyrax-bls12-381/polyCommit.hpp`; nothing outside that C++ had ever checked a whole proof).

The protocol is restated here from its definition -- the reference's verifier loop (src/verifier.cpp:118-373: challenge order, round checks, the final check
of a layer from the wiring predicates, the layer-0 combine) and this repo's zero-knowledge extension (masked round polynomials, masked evaluation claims,
one proof of dot product for the revealed values, the input's masked claim opened against P + Z D_0) -- with Python integers, Python's own curve arithmetic
and the gate lists as data (`oracle_session_layer_dump`). It parses the bytes of a transcript the CPU oracle produced (the GPU prover's are byte-identical:
tests/test_zk_gpu.py), redraws the verifier's challenges from the seeded stream in the order the protocol fixes, and accepts; it rejects the same
transcript with one bit flipped -- with the same verdict as the product's verifier on every corrupted copy. Every layer type of the reference: fully
connected, direct convolution, pooling, ReLU / truncation (with and without a second phase), and the FFT convolution's PADDING / FFT / DOT_PROD / IFFT layers
with their closed-form predicates (the multilinear extension of the DFT matrix, the (vector, position) split of the padding layer, the cubic phase of the
dot-product layer) over two pictures per circuit."""
import ctypes

import numpy as np
import pytest

import zkcnn_amd
from tests import oracle_ffi
from tests.test_circuit_cpu import dump
from tests.test_hyrax_cpu import _decompress, _eq, _msm, _padd, _pmul, _points
from zkcnn_amd import P_MOD, R_MOD, from_mont, u64p

ZK, REUSE = zkcnn_amd.MODE_ZK, zkcnn_amd.MODE_REUSE_GENS
U0, U1, V0, V1 = 0, 1, 2, 3


class Reject(Exception):
    pass


class _Stream:
    """the verifier's challenge stream: the draws of the seeded generator, in order"""

    def __init__(self, oracle, seed, n=4096):
        self.v, self.k = from_mont(oracle.random(n, seed)), 0

    def draw(self, n=None):
        if n is None:
            self.k += 1
            return self.v[self.k - 1]
        self.k += n
        return list(self.v[self.k - n:self.k])


class _FsStream:
    """the non-interactive mode's challenges (host/replay.hpp: fiatShamir): state <- BLAKE2s-256(state || everything absorbed since the last challenge), one step
    per draw; the state with bit 255 cleared IS the challenge's Montgomery form if it is below r, else one more step on nothing. A field element is absorbed as
    its Montgomery form (32 bytes), a group element as its 48-byte encoding; the chain starts from the statement's encoding and the generators' digest."""

    def __init__(self, statement):
        import hashlib
        self.h = hashlib.blake2s
        self.st = self.h(b"zkcnn-amd/fiat-shamir/v2").digest()
        self.pending = bytearray(statement)
        self.rinv = pow(1 << 256, -1, R_MOD)

    def absorb_fr(self, v):
        self.pending += ((v << 256) % R_MOD).to_bytes(32, "little")

    def absorb(self, b):
        self.pending += b

    def draw(self, n=None):
        if n is not None:
            return [self.draw() for _ in range(n)]
        while True:
            self.st = self.h(self.st + bytes(self.pending)).digest()
            self.pending = bytearray()
            t = int.from_bytes(self.st, "little") & ((1 << 255) - 1)
            if t < R_MOD:
                return t * self.rinv % R_MOD


class _Msgs:
    def __init__(self, data, tap=None):
        self.d, self.o, self.tap = data, 0, tap

    def fr(self):
        self.o += 32
        if self.o > len(self.d):
            raise Reject("truncated")
        v = int.from_bytes(self.d[self.o - 32:self.o], "little")
        if v >= R_MOD:
            raise Reject("non-canonical field element")
        if self.tap:
            self.tap.absorb_fr(v)
        return v

    def g1(self):
        self.o += 48
        if self.o > len(self.d):
            raise Reject("truncated")
        try:
            pt = _decompress(self.d[self.o - 48:self.o])
        except AssertionError:
            raise Reject("bad point")
        if self.tap:
            self.tap.absorb(self.d[self.o - 48:self.o])
        return pt


def _ev(coef_high_first, t):
    acc = 0
    for c in coef_high_first:
        acc = (acc * t + c) % R_MOD
    return acc


def _chunks(gates, n=1 << 18):
    """a gate list as Python tuples, a slice at a time (a large model has 10^8 gates)"""
    for k in range(0, len(gates), n):
        yield gates[k:k + n].tolist()


def _z(rs):
    z = 1
    for r in rs:
        z = z * r * (1 - r) % R_MOD
    return z


def _circuit(o, size):
    n = ctypes.c_uint64(0)
    o.lib.oracle_session_consts(ctypes.c_void_p(o.h), None, ctypes.c_uint64(0), ctypes.byref(n), None)
    tm, sc = np.zeros((n.value, 4), dtype=np.uint64), np.zeros((size, 4), dtype=np.uint64)
    o.lib.oracle_session_consts(ctypes.c_void_p(o.h), u64p(tm), ctypes.c_uint64(n.value), ctypes.byref(n), u64p(sc))
    keys = ("ty", "size", "n_uni", "n_bin", "size_u0", "size_u1", "size_v0", "size_v1", "bl", "bl_u0", "bl_u1", "bl_v0", "bl_v1", "max_u", "max_v", "fft_bl", "phase2", "zero_start")
    layers = []
    for i in range(size):
        m, uni, bn, ou, ov = dump(o.lib, o.h, i)
        L = dict(zip(keys, m))
        L.update(uni=uni, bin=bn, ori_u=ou, ori_v=ov)
        layers.append(L)
    return layers, from_mont(tm), from_mont(sc)


FFT, IFFT, DOT_PROD, PADDING = 1, 2, 9, 10          # layerType (reference src/circuit.h:35-37)


def _active(L, s):
    """zk_mask.hpp slotActive: the operand table exists and its phase has a variable (a DOT_PROD layer's phase 1 ends in ONE claim)"""
    if s >= 2 and not L["phase2"]:
        return False
    if L["ty"] == DOT_PROD and s == 0:
        return False
    if (L["max_v"] if s >= 2 else L["max_u"]) < 1:
        return False
    return (L["bl_u0"], L["bl_u1"], L["bl_v0"], L["bl_v1"])[s] >= 0


G1_X = 0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb      # the generator every BLS12-381 implementation publishes
G1_Y = 0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1


def python_verify(oracle, o, transcript, seed, n_layers, zk=True, fs_statement=None, fresh_gens=False, full_ipa=False):
    """raises Reject; returns the number of messages checked. zk=False: the plain protocol (the reference's own: no masks, claims in the clear, the input opened
    by the inner-product argument -- which for these small inputs stops before its first round: the prover sends the combined row)"""
    layers, two_mul, scales = _circuit(o, n_layers)
    size, logn = len(layers), layers[0]["bl"]
    rb, cb = logn >> 1, logn - (logn >> 1)
    m = 1 << cb
    # fs_statement: the non-interactive mode -- every challenge is a hash of the statement and of the messages so far, drawn AFTER the message it answers
    lazy = fs_statement is not None
    rnd = _FsStream(fs_statement) if lazy else _Stream(oracle, seed, 8192 + m)          # (the generators of the reference's set-up are drawn from it as well)
    tr = _Msgs(transcript, rnd if lazy else None)
    if fresh_gens:
        # the reference's own set-up (src/verifier.cpp:119-128): the verifier draws a random multiple of the base point per column, before anything else
        assert (G1_Y * G1_Y - G1_X ** 3 - 4) % P_MOD == 0
        pts = [_pmul(k, (G1_X, G1_Y)) for k in rnd.draw(m + (1 if zk else 0))]
    else:
        gens_mont, _ = oracle.public_generators(m + (1 if zk else 0))
        pts = _points(oracle, gens_mont)
    g, H = pts[:m], pts[m] if zk else None
    comm = [tr.g1() for _ in range(1 << rb)]

    # ---- the mode's plan (zk_mask.hpp: plan): row 0 | g of every instance | M of every claim ----
    inst, total = [], m
    for i in list(range(size - 1, 0, -1)) + [0]:
        l1 = layers[i]["max_u"] if i else layers[0]["bl"]
        l2 = layers[i]["max_v"] if i and layers[i]["phase2"] else 0
        base1 = 3 if i and layers[i]["ty"] == DOT_PROD else 2          # a DOT_PROD layer's phase 1 is a cubic sumcheck (reference src/prover.cpp:103-144)
        deg = [(base1 if j < l1 else 2) + (1 if zk and (j == l1 - 1 or j == l1 + l2 - 1) else 0) for j in range(l1 + l2)]
        inst.append(dict(layer=i, off=total, deg=deg))
        total += 1 + sum(deg)
    slot = {}
    for i in range(1, size):
        for s in range(4):
            slot[(i, s)] = total
            total += 1
    rows = (total + m - 1) // m if zk else 0
    mask_commit = [tr.g1() for _ in range(rows)]
    sums = [tr.fr() if zk else 0 for _ in inst]
    rho = rnd.draw() if zk else 0
    u_vec, zmask = [0] * total, {}
    closed = []                                       # what every instance leaves for the proof of dot product (gamma is drawn after the last of them)

    # ---- stage 1: the layers (reference src/verifier.cpp:132-266) ----
    alpha, beta, relu_rou = 1, 0, 1
    r_u, r_v = {size: rnd.draw(layers[size - 1]["bl"]), size + 1: []}, {size: [], size + 1: []}
    claim_u0, claim_v0 = {}, {}
    prev_sum = tr.fr()                                # Vres: the public output's value
    claim_u1 = claim_v1 = 0
    n_checked = 1
    for k, i in enumerate(range(size - 1, 0, -1)):
        L = layers[i]
        r_u[i] = rnd.draw(L["max_u"])
        relu_rou = rnd.draw() if L["zero_start"] < L["size"] else 1
        incoming = []                                  # (slot key, weight) of the masked claims this layer starts from
        if i < size - 1:
            if layers[i + 1]["ty"] in (FFT, IFFT):           # the transform's claim is carried as it is
                if _active(layers[i + 1], U1):
                    incoming.append(((i + 1, U1), 1))
            else:
                if _active(layers[i + 1], U1):
                    incoming.append(((i + 1, U1), alpha))
                if _active(layers[i + 1], V1):
                    incoming.append(((i + 1, V1), beta))
        claim = (prev_sum + rho * sums[k]) % R_MOD
        point, E = [], 1
        # phase 1, phase 2
        finals = {}
        finals[V0], finals[V1] = 0, claim_v1              # (a layer without a second phase: its v claims are not sent; beta is 0 then)
        r_v[i] = []
        for ph, (ell, rs, s0) in enumerate(((L["max_u"], r_u[i], U0), (L["max_v"], None, V0))):
            if ph == 1:
                if not L["phase2"]:
                    break
                r_v[i] = rnd.draw(L["max_v"])
                rs = r_v[i]
            for j in range(ell):
                deg = inst[k]["deg"][len(point)]
                coef = [tr.fr() for _ in range(deg + 1)]      # highest first
                n_checked += 1
                if lazy:
                    rs[j] = rnd.draw()                       # (the draws made up front are overwritten: reference order kept, values replaced)
                if (_ev(coef, 0) + _ev(coef, 1)) % R_MOD != claim:
                    raise Reject(f"layer {i} phase {ph + 1} round {j}")
                claim = _ev(coef, rs[j])
                point.append(rs[j])
                E = E * (1 - rs[j]) % R_MOD
            if L["ty"] == DOT_PROD and ph == 0:
                c0, c1 = 0, tr.fr()                          # sumcheckDotProdFinalize1 (reference src/prover.cpp:146-153): the one operand table's value
            else:
                c0, c1 = tr.fr(), tr.fr()
            finals[s0], finals[s0 + 1] = c0, c1
            for b in range(2):
                if _active(L, s0 + b):
                    zmask[(i, s0 + b)] = _z(rs[:ell])
        v = tr.fr() if zk else 0
        claim = (claim - v) % R_MOD
        closed.append(dict(k=k, point=point, E=E, v=v, incoming=incoming))
        # the layer's final check from the wiring predicates (src/verifier.cpp:36-116, :256-262)
        bl, r0, r1 = L["bl"], r_u[i + 1], r_v[i + 1]
        sc, ty, fft_bl = scales[i], L["ty"], L["fft_bl"]
        bu = _eq(r_u[i])
        uni, binv = [0, 0], [0, 0, 0]
        if ty in (FFT, IFFT):
            # the transform as a layer: out[g] = scale sum_u w^{+-g u} in[u]; its predicate at (r0, r_u) is the multilinear extension of the DFT matrix
            # (reference src/verifier.cpp:89-93 with the table of src/utils.cpp:61-103, restated from its definition as in tests/test_field_cpu.py)
            w = from_mont(oracle.root_of_unity(fft_bl))[0]
            gbits = fft_bl - 1 if ty == IFFT else fft_bl
            if ty == IFFT:
                w = pow(w, -1, R_MOD)
            eg = _eq(r0[:gbits])
            wp = [pow(w, e, R_MOD) for e in range(1 << fft_bl)]
            for uu in range(1 << L["max_u"]):
                phi = sc * sum(eg[gi] * wp[gi * uu % (1 << fft_bl)] for gi in range(1 << gbits)) % R_MOD
                uni[1] += phi * bu[uu]
        else:
            if ty == PADDING:
                # outputs are (vector, position): the vector part carries the two-point claim of the DOT_PROD layer two levels up, the position part the
                # transform's sumcheck point (reference src/verifier.cpp:46-58)
                fft_blh = fft_bl - 1
                hi = [alpha * x % R_MOD for x in _eq(r_u[i + 2][fft_bl:fft_bl + bl - fft_blh])]
                if beta:
                    hi = [(p + beta * q) % R_MOD for p, q in zip(hi, _eq(r_v[i + 2][:bl - fft_blh]))]
                lo = _eq(r0[:fft_blh])
                bg = [hi[gi >> fft_blh] * lo[gi & ((1 << fft_blh) - 1)] % R_MOD for gi in range(1 << bl)]
            elif ty == DOT_PROD:
                # (reference src/verifier.cpp:59-69) output (vector, frequency): the vector part at the point two levels up, the frequency part tied to this
                # layer's own point by eq
                bg = [alpha * x % R_MOD for x in _eq(r_u[i + 2][fft_bl - 1:fft_bl - 1 + bl - fft_bl])]
                same = 1
                for j in range(fft_bl):
                    same = same * (r0[j] * r_u[i][j] + (1 - r0[j]) * (1 - r_u[i][j])) % R_MOD
                bu = [same * x % R_MOD for x in _eq(r_u[i][fft_bl:L["max_u"]])]
            else:
                bg = [alpha * sc * x % R_MOD for x in _eq(r0[:bl])]
                if beta:
                    bg = [(p + beta * sc * q) % R_MOD for p, q in zip(bg, _eq(r1[:bl]))]
                if L["zero_start"] < L["size"]:
                    bg = [x * relu_rou % R_MOD if gi >= L["zero_start"] else x for gi, x in enumerate(bg)]
            for chunk in _chunks(L["uni"]):
                for gg, uu, lu, s in chunk:
                    uni[1 if lu else 0] += bg[gg] * bu[uu] * two_mul[s]
                uni = [x % R_MOD for x in uni]
            if L["phase2"]:
                bv = _eq(r_v[i])
                uni = [x * bv[0] % R_MOD for x in uni]
                for chunk in _chunks(L["bin"]):
                    for gg, uu, vv, s, ll in chunk:
                        binv[ll] += bg[gg] * bu[uu] % R_MOD * bv[vv] % R_MOD * (1 if ty == DOT_PROD else two_mul[s])
                binv = [x % R_MOD for x in binv]
        cu0, cu1, cv0, cv1 = finals[U0], finals[U1], finals[V0], finals[V1]
        expect = (binv[0] * cu0 * cv0 + binv[1] * cu1 * cv1 + binv[2] * cu1 * cv0 + uni[0] * cu0 + uni[1] * cu1) % R_MOD
        if claim != expect:
            raise Reject(f"final check of layer {i}")
        claim_u0[i], claim_v0[i], claim_u1, claim_v1 = cu0, cv0, cu1, cv1
        if ty in (FFT, IFFT):
            prev_sum = claim_u1                              # carry the claim, alpha / beta stay as they are (reference src/verifier.cpp:228-230)
        else:
            alpha = rnd.draw() if L["bl_u1"] >= 0 else 0
            beta = rnd.draw() if L["bl_v1"] >= 0 else 0
            prev_sum = (alpha * claim_u1 + beta * claim_v1) % R_MOD

    # ---- stage 2: all claims about layer 0 in one sumcheck (src/verifier.cpp:268-357) ----
    sig_u, sig_v = rnd.draw(size - 1), rnd.draw(size - 1)
    r_u[0] = rnd.draw(logn)
    prev_sum, incoming = 0, []
    for i in range(1, size):
        if layers[i]["bl_u0"] >= 0:
            prev_sum += sig_u[i - 1] * claim_u0[i]
        if layers[i]["bl_v0"] >= 0:
            prev_sum += sig_v[i - 1] * claim_v0[i]
        if _active(layers[i], U0):
            incoming.append(((i, U0), sig_u[i - 1]))
        if _active(layers[i], V0):
            incoming.append(((i, V0), sig_v[i - 1]))
    k = len(inst) - 1
    claim = (prev_sum + rho * sums[k]) % R_MOD
    point, E = [], 1
    for j in range(logn):
        coef = [tr.fr() for _ in range(inst[k]["deg"][j] + 1)]
        n_checked += 1
        if lazy:
            r_u[0][j] = rnd.draw()
        if (_ev(coef, 0) + _ev(coef, 1)) % R_MOD != claim:
            raise Reject(f"layer-0 combine, round {j}")
        claim = _ev(coef, r_u[0][j])
        point.append(r_u[0][j])
        E = E * (1 - r_u[0][j]) % R_MOD
    eval_in = tr.fr()
    z_in = _z(r_u[0])
    v = tr.fr() if zk else 0
    claim = (claim - v) % R_MOD
    closed.append(dict(k=k, point=point, E=E, v=v, incoming=incoming))
    bg0 = _eq(r_u[0])
    gr = 0
    for i in range(1, size):
        L = layers[i]
        for blk, ori, sig, pt in ((L["bl_u0"], L["ori_u"], sig_u[i - 1], r_u[i]), (L["bl_v0"], L["ori_v"], sig_v[i - 1], r_v[i])):
            if blk >= 0:
                e = _eq(pt[:blk])
                gr += sig * sum(bg0[int(ori[j])] * e[j] for j in range(len(ori)))
    if eval_in * gr % R_MOD != claim:
        raise Reject("layer-0 combine, final check")

    Lrow, b = _eq(r_u[0][cb:]), _eq(r_u[0][:cb])
    if not zk:
        # the inner-product argument (polyCommit.hpp). The recursion stops at length 256 (no round at all for the small inputs: the combined row w = L^T Z comes in
        # the clear; one round for LeNet5's 512 columns) or, full_ipa, at length 1: a round sends the cross terms, the verifier folds P, y and -- here, explicitly -- the generators and the eq vector
        P, y, gg, bb = _msm(Lrow, comm), eval_in, list(g), list(b)
        while len(gg) > (1 if full_ipa else 256):
            h = len(gg) // 2
            Lk, Rk, yL, yR = tr.g1(), tr.g1(), tr.fr(), tr.fr()
            c = rnd.draw()
            P = _padd(_padd(Lk, _pmul(c, P)), _pmul(c * c % R_MOD, Rk))
            y = (yL + c * y + c * c * yR) % R_MOD
            gg = [_padd(_pmul(c, gg[j]), gg[j + h]) for j in range(h)]
            bb = [(c * bb[j] + bb[j + h]) % R_MOD for j in range(h)]
        wrow = [tr.fr() for _ in range(len(gg))]
        if tr.o != len(transcript):
            raise Reject("trailing bytes")
        if P != _msm(wrow, gg):
            raise Reject("input opening: commitment")
        if sum(p * q for p, q in zip(wrow, bb)) % R_MOD != y:
            raise Reject("input opening: value")
        return n_checked

    # ---- the revealed values against the commitment: one proof of dot product over rows 1 .. (zk_mask.hpp) ----
    gamma = rnd.draw()
    w, y = 1, 0
    for rec in closed:
        it = inst[rec["k"]]
        u_vec[it["off"]] = (u_vec[it["off"]] + w * rho) % R_MOD
        o_ = it["off"] + 1
        for j, d in enumerate(it["deg"]):
            pw = w * rho % R_MOD
            for e in range(d):
                pw = pw * rec["point"][j] % R_MOD
                u_vec[o_ + e] = (u_vec[o_ + e] + pw) % R_MOD
            o_ += d
        for key, weight in rec["incoming"]:
            u_vec[slot[key]] = (u_vec[slot[key]] + w * rec["E"] % R_MOD * weight % R_MOD * zmask[key]) % R_MOD
        y = (y + w * rec["v"]) % R_MOD
        w = w * gamma % R_MOD
    delta, t = [tr.g1() for _ in range(rows - 1)], tr.fr()
    c = rnd.draw()
    z = [tr.fr() for _ in range((rows - 1) * m)]
    z_blind = [tr.fr() for _ in range(rows - 1)]
    for rr in range(rows - 1):
        if _padd(_msm(z[rr * m:(rr + 1) * m], g), _pmul(z_blind[rr], H)) != _padd(_pmul(c, mask_commit[rr + 1]), delta[rr]):
            raise Reject(f"masks: row {rr + 1} of the proof of dot product")
    if sum(p * q for p, q in zip(z, u_vec[m:])) % R_MOD != (c * y + t) % R_MOD:
        raise Reject("masks: value of the proof of dot product")

    # ---- the input's masked claim: proof of dot product against P + Z D_0 (polyCommit.hpp: verifyZk) ----
    P = _padd(_msm(Lrow, comm), _pmul(z_in, mask_commit[0]))
    d1, t1 = tr.g1(), tr.fr()
    c1 = rnd.draw()
    z1, zs1 = [tr.fr() for _ in range(m)], tr.fr()
    if tr.o != len(transcript):
        raise Reject("trailing bytes")
    if _padd(_msm(z1, g), _pmul(zs1, H)) != _padd(_pmul(c1, P), d1):
        raise Reject("input opening: commitment")
    if sum(p * q for p, q in zip(z1, b)) % R_MOD != (c1 * eval_in + t1) % R_MOD:
        raise Reject("input opening: value")
    return n_checked


MODELS = [("custom:F8 F4", (4, 4, 1), 1), ("custom:F6 F5 F3", (4, 4, 1), 1), ("custom:C2:3:1:s M F4", (4, 4, 1), 1), ("custom:C2:3:1:n A F4", (4, 4, 2), 1),
          ("custom:C2:3:1:f M F4", (4, 4, 1), 2)]          # the last one: an FFT convolution over two pictures (PADDING, FFT, DOT_PROD, IFFT layers)


@pytest.mark.parametrize("model,pic,pp", [MODELS[0], MODELS[2], MODELS[4]])
def test_python_verifier_accepts_the_zero_knowledge_transcript_and_rejects_corruptions(oracle, model, pic, pp):
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        res, tr = o.prove(seed=0x5EED0042, mode=ZK | REUSE)
        assert res.accepted == 1
        n = python_verify(oracle, o, tr, 0x5EED0042, res.n_layers)
        assert n == res.n_rounds + 1                                   # Vres and every round polynomial went through a check
        # another proof of the same statement (other coins through the seed): accepted too, and not the same bytes
        res2, tr2 = o.prove(seed=0x5EED0043, mode=ZK | REUSE)
        assert tr2 != tr and python_verify(oracle, o, tr2, 0x5EED0043, res.n_layers) == n
        # one flipped byte in the sumcheck part (behind the commitments, before the proofs of dot product): rejected wherever it lands
        logn = res.input_bits
        start = 48 * (1 << (logn >> 1))
        import random
        rnd = random.Random(9)
        rejected = 0
        for _ in range(6):
            pos = rnd.randrange(start, len(tr) - 32 * (1 << (logn - (logn >> 1))))
            bad = bytearray(tr)
            bad[pos] ^= 1 << rnd.randrange(8)
            try:
                python_verify(oracle, o, bytes(bad), 0x5EED0042, res.n_layers)
                mine = True
            except Reject:
                mine = False
            assert mine == (o.verify(bytes(bad), seed=0x5EED0042, mode=ZK | REUSE).accepted == 1), pos      # (same verdict as the product's verifier)
            rejected += not mine
        assert rejected >= 5
        # the verifier of the product agrees on both counts
        assert o.verify(tr, seed=0x5EED0042, mode=ZK | REUSE).accepted == 1


@pytest.mark.parametrize("model,pic,pp", MODELS)
def test_python_verifier_accepts_the_plain_transcript_and_rejects_corruptions(oracle, model, pic, pp):
    """the same restatement without the zero-knowledge extension: the reference's protocol as it is (src/verifier.cpp:118-373), claims in the clear"""
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        res, tr = o.prove(seed=0x5EED0044, mode=REUSE)
        assert res.accepted == 1
        assert python_verify(oracle, o, tr, 0x5EED0044, res.n_layers, zk=False) == res.n_rounds + 1
        # one flipped bit anywhere: this verifier and the product's agree on the verdict (a few claims of the plain protocol are read by nobody -- the value of an
        # operand table a layer does not have: both accept those), and almost every position is rejected
        import random
        rnd = random.Random(10)
        rejected = 0
        for _ in range(10):
            pos = rnd.randrange(len(tr))
            bad = bytearray(tr)
            bad[pos] ^= 1 << rnd.randrange(8)
            try:
                python_verify(oracle, o, bytes(bad), 0x5EED0044, res.n_layers, zk=False)
                mine = True
            except Reject:
                mine = False
            assert mine == (o.verify(bytes(bad), seed=0x5EED0044, mode=REUSE).accepted == 1), pos
            rejected += not mine
        assert rejected >= 8



def _fs_statement(o, n_gens):
    ln = ctypes.c_uint64(0)
    buf = (ctypes.c_uint8 * (1 << 16))()
    assert o.lib.oracle_session_fs_statement(ctypes.c_void_p(o.h), ctypes.c_uint64(n_gens), buf, ctypes.c_uint64(len(buf)), ctypes.byref(ln)) == 0
    return bytes(buf[:ln.value])


@pytest.mark.parametrize("model,pic,pp,zk", [MODELS[0] + (False,), MODELS[0] + (True,), MODELS[4] + (True,)])
def test_python_verifier_accepts_non_interactive_proofs(oracle, model, pic, pp, zk):
    """Fiat-Shamir: the proof is the transcript, every challenge a BLAKE2s chain step over the statement and the messages so far (hashlib here).
    The statement's encoding is taken as bytes from the session; the chain, the challenges' derivation and every check are Python's own. One flipped bit
    changes every later challenge: rejected, as by the product's off-line verifier."""
    FS = zkcnn_amd.MODE_FIAT_SHAMIR
    mode = FS | (ZK if zk else 0)
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        res, tr = o.prove(mode=mode)
        assert res.accepted == 1 and o.verify(tr, mode=mode).accepted == 1
        cb = res.input_bits - (res.input_bits >> 1)
        stmt = _fs_statement(o, (1 << cb) + (1 if zk else 0))
        assert python_verify(oracle, o, tr, None, res.n_layers, zk=zk, fs_statement=stmt) == res.n_rounds + 1
        import random
        rnd = random.Random(12)
        for _ in range(4):
            bad = bytearray(tr)
            bad[rnd.randrange(len(tr))] ^= 1 << rnd.randrange(8)
            try:
                python_verify(oracle, o, bytes(bad), None, res.n_layers, zk=zk, fs_statement=stmt)
                mine = True
            except Reject:
                mine = False
            assert mine == (o.verify(bytes(bad), mode=mode).accepted == 1)
        # another statement (one byte of its encoding changed): unrelated challenges, the same proof does not verify
        with pytest.raises(Reject):
            python_verify(oracle, o, tr, None, res.n_layers, zk=zk, fs_statement=stmt[:-1] + bytes([stmt[-1] ^ 1]))


def test_python_verifier_accepts_proofs_over_fresh_generators(oracle):
    """the reference's semantics (src/verifier.cpp:119-128): the verifier draws the commitment generators itself, k_i G from its challenge stream, for every
    proof -- the generators here are Python's own multiples of the published base point"""
    model, pic, pp = MODELS[2]
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        for zk, full in ((False, False), (True, False), (False, True)):      # the last one: the reference's semantics in full -- fresh generators, argument down to length 1
            mode = (ZK if zk else 0) | (zkcnn_amd.MODE_FULL_IPA if full else 0)
            res, tr = o.prove(seed=0x5EED0060, mode=mode)
            assert res.accepted == 1
            assert python_verify(oracle, o, tr, 0x5EED0060, res.n_layers, zk=zk, fresh_gens=True, full_ipa=full) == res.n_rounds + 1
            for pos in (len(tr) // 3, len(tr) - 40):
                bad = bytearray(tr)
                bad[pos] ^= 4
                with pytest.raises(Reject):
                    python_verify(oracle, o, bytes(bad), 0x5EED0060, res.n_layers, zk=zk, fresh_gens=True, full_ipa=full)


def test_python_verifier_accepts_the_committed_lenet5_transcript(oracle):
    """BASELINE.json configs[0] / [1]: LeNet5, pic_cnt = 1 (24 layers, 459 rounds, FFT convolutions, 2^18 inputs) in the reference's own set-up -- interactive,
    generators drawn by the verifier. The transcript whose SHA-256 is committed in tests/golden/transcripts.json (and which the GPU prover reproduces:
    tests/test_live_fallback_gpu.py, test_protocol_gpu.py) passes the independent verifier; a flipped bit does not."""
    import hashlib
    import json
    import os
    golden = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transcripts.json")))["lenet5_pic1"]["sha256"]
    with oracle_ffi.OracleSession("lenet", (32, 32, 1), 1) as o:
        res, tr = o.prove(seed=0x5EED0001)
        assert res.accepted == 1 and hashlib.sha256(tr).hexdigest() == golden
        assert python_verify(oracle, o, tr, 0x5EED0001, res.n_layers, zk=False, fresh_gens=True) == res.n_rounds + 1
        bad = bytearray(tr)
        bad[len(tr) // 2] ^= 16
        with pytest.raises(Reject):
            python_verify(oracle, o, bytes(bad), 0x5EED0001, res.n_layers, zk=False, fresh_gens=True)


GOLDEN = ["fc_only", "naive_conv_maxpool", "naive_conv_mul_add_avgpool", "fft_conv_block_8x8_batch2", "mixed_22_layers"]


@pytest.mark.parametrize("name", GOLDEN)
def test_python_verifier_accepts_the_committed_golden_transcripts(oracle, name):
    """tests/golden/transcripts.json holds the SHA-256 of every model's transcript in the reference's set-up (interactive, generators drawn by the verifier) and as
    a non-interactive proof -- the vectors the GPU prover is held to (tests/test_protocol_gpu.py). The transcripts with exactly those digests pass the
    independent verifier: every committed golden vector is a proof that an implementation sharing no code with the prover accepts."""
    import hashlib
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "transcripts.json")))[name]
    with oracle_ffi.OracleSession(g["model"], tuple(g["pic"]), g["pic_cnt"]) as o:
        res, tr = o.prove(seed=g["challenge_seed"])
        assert hashlib.sha256(tr).hexdigest() == g["sha256"] and len(tr) == g["transcript_len"]
        assert python_verify(oracle, o, tr, g["challenge_seed"], res.n_layers, zk=False, fresh_gens=True) == g["n_rounds"] + 1
        res, tr = o.prove(mode=zkcnn_amd.MODE_FIAT_SHAMIR)
        assert hashlib.sha256(tr).hexdigest() == g["fiat_shamir_sha256"]
        cb = res.input_bits - (res.input_bits >> 1)
        assert python_verify(oracle, o, tr, None, res.n_layers, zk=False, fs_statement=_fs_statement(o, 1 << cb)) == g["n_rounds"] + 1
