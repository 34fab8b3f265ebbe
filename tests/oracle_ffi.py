"""ctypes bindings of oracle/liboracle.so -- the CPU checker. Test infrastructure only."""
import ctypes
import os

import numpy as np

import zkcnn_amd
from zkcnn_amd import u64p

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Oracle:
    def __init__(self, lib):
        self.lib = lib

    # ---- field ----
    def from_canonical(self, a):
        out = np.zeros_like(a)
        self.lib.oracle_fr_from_canonical(u64p(out), u64p(a), ctypes.c_uint64(a.shape[0]))
        return out

    def to_canonical(self, a):
        out = np.zeros_like(a)
        self.lib.oracle_fr_to_canonical(u64p(out), u64p(a), ctypes.c_uint64(a.shape[0]))
        return out

    def binop(self, op, a, b):
        out = np.zeros_like(a)
        getattr(self.lib, "oracle_fr_" + op)(u64p(out), u64p(a), u64p(b), ctypes.c_uint64(a.shape[0]))
        return out

    def inv(self, a):
        out = np.zeros_like(a)
        self.lib.oracle_fr_inv(u64p(out), u64p(a), ctypes.c_uint64(a.shape[0]))
        return out

    def random(self, n, seed):
        out = np.zeros((n, 4), dtype=np.uint64)
        self.lib.oracle_fr_random(u64p(out), ctypes.c_uint64(n), ctypes.c_uint64(seed))
        return out

    def root_of_unity(self, n):
        out = np.zeros((1, 4), dtype=np.uint64)
        self.lib.oracle_root_of_unity(u64p(out), ctypes.c_int32(n))
        return out

    # ---- tables ----
    def eq_table2(self, n, r0, r1, alpha, beta):
        out = np.zeros((1 << n, 4), dtype=np.uint64)
        self.lib.oracle_eq_table2(u64p(out), ctypes.c_int32(n), u64p(r0), u64p(r1), u64p(alpha), u64p(beta))
        return out

    def eq_table1(self, n, r, init):
        out = np.zeros((1 << n, 4), dtype=np.uint64)
        self.lib.oracle_eq_table1(u64p(out), ctypes.c_int32(n), u64p(r), u64p(init))
        return out

    def phi_table(self, rx, scale, n, inverse):
        cnt = (1 << n) if inverse else (1 << (n - 1))
        out = np.zeros((cnt, 4), dtype=np.uint64)
        self.lib.oracle_phi_table(u64p(out), u64p(rx), u64p(scale), ctypes.c_int32(n), ctypes.c_int32(int(inverse)))
        return out

    def ntt(self, data, logn, inverse):
        data = np.ascontiguousarray(data.copy())
        batch = data.shape[0] >> logn
        self.lib.oracle_ntt(u64p(data), ctypes.c_int32(logn), ctypes.c_int32(int(inverse)), ctypes.c_uint64(batch))
        return data

    def round_quadratic(self, V, M, r, first):
        out = np.zeros((3, 4), dtype=np.uint64)
        self.lib.oracle_round_quadratic.restype = ctypes.c_uint64
        n = self.lib.oracle_round_quadratic(u64p(V), u64p(M), ctypes.c_uint64(V.shape[0]), u64p(r), ctypes.c_int32(int(first)), u64p(out))
        return out, n

    # ---- curve ----
    def generators(self, n, seed):
        out = np.zeros((n, 12), dtype=np.uint64)
        self.lib.oracle_g1_generators(u64p(out), ctypes.c_uint64(n), ctypes.c_uint64(seed))
        return out

    def public_generators(self, n):
        out = np.zeros((n, 12), dtype=np.uint64)
        dig = (ctypes.c_uint8 * 32)()
        self.lib.oracle_public_generators(u64p(out), dig, ctypes.c_uint64(n))
        return out, bytes(dig)

    def g1_in_subgroup(self, p):
        """r * P == O, through the compressed round trip with the subgroup check on"""
        return self.g1_deserialize(self.g1_serialize(p), check_subgroup=True) is not None

    def msm(self, scalars, bases):
        out = np.zeros(12, dtype=np.uint64)
        self.lib.oracle_msm(u64p(out), u64p(scalars), u64p(bases), ctypes.c_uint64(scalars.shape[0]))
        return out

    def g1_base(self):
        out = np.zeros(12, dtype=np.uint64)
        self.lib.oracle_g1_base(u64p(out))
        return out

    def g1_mul(self, p, k):
        out = np.zeros(12, dtype=np.uint64)
        self.lib.oracle_g1_mul(u64p(out), u64p(p), u64p(k))
        return out

    def g1_add(self, p, q):
        out = np.zeros(12, dtype=np.uint64)
        self.lib.oracle_g1_add(u64p(out), u64p(p), u64p(q))
        return out

    def g1_on_curve(self, p):
        return bool(self.lib.oracle_g1_on_curve(u64p(p)))

    def g1_serialize(self, p):
        buf = (ctypes.c_uint8 * 48)()
        self.lib.oracle_g1_serialize(buf, u64p(p))
        return bytes(buf)

    def g1_deserialize(self, data, check_subgroup=True):
        out = np.zeros(12, dtype=np.uint64)
        buf = (ctypes.c_uint8 * 48).from_buffer_copy(data)
        ok = self.lib.oracle_g1_deserialize(u64p(out), buf, ctypes.c_int32(int(check_subgroup)))
        return out if ok else None

    def sha256(self, data, split=0):
        out = (ctypes.c_uint8 * 32)()
        buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
        self.lib.oracle_sha256(out, buf, ctypes.c_uint64(len(data)), ctypes.c_uint64(split))
        return bytes(out)

    def blake2s(self, data, split=0):
        out = (ctypes.c_uint8 * 32)()
        buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
        self.lib.oracle_blake2s(out, buf, ctypes.c_uint64(len(data)), ctypes.c_uint64(split))
        return bytes(out)

    def blake2s_chain(self, state, msg):
        """state' = BLAKE2s(state || msg) on 32-bit words (the GPU kernel's form of the Fiat-Shamir chain step)"""
        st = (ctypes.c_uint32 * 8)(*state)
        m = (ctypes.c_uint32 * max(len(msg), 1))(*msg)
        self.lib.oracle_blake2s_chain(st, m, ctypes.c_int32(len(msg)))
        return list(st)

    def fp_to_canonical(self, a):
        a = np.ascontiguousarray(a.reshape(-1, 6))
        out = np.zeros_like(a)
        self.lib.oracle_fp_to_canonical(u64p(out), u64p(a), ctypes.c_uint64(a.shape[0]))
        return out


class OracleSession(zkcnn_amd._SessionBase):
    """whole-proof driver of include/zkcnn_api.h backed by the CPU restatement"""
    _prefix = "oracle_"

    def __init__(self, model, pic=(32, 32, 1), pic_cnt=1, data_seed=20260928, statement=None, picture_seed=0, calibrated=None):
        super().__init__(load().lib, model, pic, pic_cnt, data_seed, 0, statement, picture_seed, calibrated)


_oracle = None


def load():
    global _oracle
    if _oracle is None:
        path = os.path.join(ROOT, "oracle", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle`")
        _oracle = Oracle(ctypes.CDLL(path))
    return _oracle


class YieldingOracleSession:
    """CPU session whose prover gives up the thread at every round call when it runs on a fiber (oracle_api.cpp: yieldingProver): the
    stand-in for a GPU lane in the host-logic tests of batchSessionT (zkcnn_amd/csrc/host/session.hpp)."""

    def __init__(self, model, pic, pic_cnt=1, data_seed=20260928, picture_seed=0):
        self.lib = load().lib
        self.desc = zkcnn_amd.ModelDesc(model.encode(), pic[0], pic[1], pic[2], pic_cnt, data_seed, picture_seed)
        self.lib.oracle_session_create_yielding.restype = ctypes.c_void_p
        self.h = self.lib.oracle_session_create_yielding(ctypes.byref(self.desc))
        if not self.h:
            raise RuntimeError("oracle_session_create_yielding failed")

    def prove(self, seed, mode):
        buf = (ctypes.c_uint8 * (1 << 20))()
        res = zkcnn_amd.Result()
        rc = self.lib.oracle_session_prove_yielding(ctypes.c_void_p(self.h), ctypes.c_uint64(seed), ctypes.c_uint32(mode), buf, ctypes.c_uint64(len(buf)), ctypes.byref(res))
        assert rc == 0, res.message
        return res, ctypes.string_at(buf, res.transcript_len)

    def close(self):
        if self.h:
            self.lib.oracle_session_destroy_yielding(ctypes.c_void_p(self.h))
            self.h = None


def oracle_batch_prove(sessions, seeds, mode):
    """K yielding sessions as the lanes of one batchSessionT, driven by this thread; returns ([(Result, bytes)], driver passes)"""
    lib = load().lib
    k = len(sessions)
    bufs = [(ctypes.c_uint8 * (1 << 20))() for _ in range(k)]
    ptrs = (ctypes.c_void_p * k)(*[ctypes.addressof(b) for b in bufs])
    caps = (ctypes.c_uint64 * k)(*[len(b) for b in bufs])
    hs = (ctypes.c_void_p * k)(*[s.h for s in sessions])
    res = (zkcnn_amd.Result * k)()
    passes = ctypes.c_uint64(0)
    rc = lib.oracle_batch_prove(hs, ctypes.c_int32(k), (ctypes.c_uint64 * k)(*seeds), ctypes.c_uint32(mode), ptrs, caps, res, ctypes.byref(passes))
    assert rc == 0, [r.message for r in res]
    return [(zkcnn_amd.Result.from_buffer_copy(res[i]), ctypes.string_at(bufs[i], res[i].transcript_len)) for i in range(k)], passes.value
