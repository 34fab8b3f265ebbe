"""zkcnn_amd -- MI355X-native GKR prover for zkCNN (ctypes bindings over the C-ABI libraries).

Product libraries (built in-tree by zkcnn_amd/build.py):
  lib/libzkcnn_hip.so   include/zkcnn_hip.h   HIP kernels + prover state machine
  lib/libzkcnn_host.so  include/zkcnn_api.h   C++14 host: circuit generator, verifier, prover adapter

Nothing here falls back to a CPU implementation: without the HIP library and a GPU the calls raise.
The CPU checker lives in oracle/ and is only ever loaded by tests, smoke() and bench.py's cpu_baseline.
"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_DIR = os.path.join(ROOT, "zkcnn_amd", "lib")

R_MOD = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
P_MOD = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab

MODE_VERIFY, MODE_DRIVE_ONLY, MODE_REUSE_GENS, MODE_TAMPER, MODE_HOST_PRED, MODE_CROSS_PRED, MODE_FIAT_SHAMIR, MODE_SEEDED, MODE_FULL_IPA = 0, 1, 2, 4, 8, 16, 32, 64, 128
MODE_ZK = 1 << 24
MODE_HOST_ROUNDS = 1 << 25
MODE_HOST_TAIL = 1 << 26
MODE_GPU_TAIL = 1 << 28       # lanes of a batch: every round on the GPU (no hybrid tail)
MODE_FS_DEVICE = 1 << 27      # with MODE_FIAT_SHAMIR: BLAKE2s chain on the GPU (device-side rounds) instead of host-derived challenges over the resident kernels


class ModelDesc(ctypes.Structure):
    _fields_ = [("model", ctypes.c_char_p), ("pic_x", ctypes.c_int32), ("pic_y", ctypes.c_int32),
                ("pic_channel", ctypes.c_int32), ("pic_cnt", ctypes.c_int32), ("data_seed", ctypes.c_uint64),
                ("picture_seed", ctypes.c_uint64)]


class Result(ctypes.Structure):
    _fields_ = [("accepted", ctypes.c_int32), ("n_layers", ctypes.c_int32), ("input_size", ctypes.c_uint64),
                ("input_bits", ctypes.c_int32), ("n_rounds", ctypes.c_int32), ("n_messages", ctypes.c_int32), ("reserved_", ctypes.c_int32),
                ("prove_s", ctypes.c_double), ("poly_prove_s", ctypes.c_double), ("verify_s", ctypes.c_double),
                ("poly_verify_s", ctypes.c_double), ("proof_kb", ctypes.c_double), ("poly_proof_kb", ctypes.c_double),
                ("witness_s", ctypes.c_double), ("upload_s", ctypes.c_double), ("wall_s", ctypes.c_double),
                ("transcript_len", ctypes.c_uint64), ("gate_cnt_uni", ctypes.c_uint64), ("gate_cnt_bin", ctypes.c_uint64),
                ("table_entries", ctypes.c_uint64), ("message", ctypes.c_char * 128)]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["message"] = self.message.decode(errors="replace")
        return d


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run `python zkcnn_amd/build.py` (or __graft_entry__.build())")
    return ctypes.CDLL(path)


_hip = None
_host = None


def hip_lib():
    """libzkcnn_hip.so (raises if it has not been built)."""
    global _hip
    if _hip is None:
        _hip = _load(os.path.join(LIB_DIR, "libzkcnn_hip.so"))
        _hip.zk_last_error.restype = ctypes.c_char_p
        _hip.zk_proof_bytes.restype = ctypes.c_uint64
    return _hip


def host_lib():
    """libzkcnn_host.so (needs libzkcnn_hip.so next to it)."""
    global _host
    if _host is None:
        hip_lib()
        _host = _load(os.path.join(LIB_DIR, "libzkcnn_host.so"))
        _host.zkcnn_session_create.restype = ctypes.c_void_p
    return _host


# ---- field helpers: python ints <-> (n, 4) uint64 limb arrays -----------------------------------------
def to_limbs(xs, n=4):
    a = np.zeros((len(xs), n), dtype=np.uint64)
    for i, x in enumerate(xs):
        for j in range(n):
            a[i, j] = (x >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return a


def from_limbs(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, a.shape[-1])
    return [sum(int(a[i, j]) << (64 * j) for j in range(a.shape[1])) for i in range(a.shape[0])]


def to_mont(xs):
    """canonical python ints -> Montgomery-form limb array (the C-ABI form)"""
    return to_limbs([(x % R_MOD) * (1 << 256) % R_MOD for x in xs])


def from_mont(a):
    rinv = pow(1 << 256, -1, R_MOD)
    return [x * rinv % R_MOD for x in from_limbs(a)]


def u64p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))


class HipContext:
    """One GPU prover context (zk_ctx). Raises RuntimeError when no GPU / library is available."""

    def __init__(self, device=0):
        self.lib = hip_lib()
        self.ctx = ctypes.c_void_p()
        rc = self.lib.zk_ctx_create(ctypes.c_int32(device), ctypes.byref(self.ctx))
        if rc != 0:
            raise RuntimeError("zk_ctx_create failed: " + self.lib.zk_last_error(None).decode())

    def close(self):
        if self.ctx:
            self.lib.zk_ctx_destroy(self.ctx)
            self.ctx = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, what):
        if rc != 0:
            raise RuntimeError(f"{what} failed ({rc}): " + self.lib.zk_last_error(self.ctx).decode())

    def fr_binop(self, op, a, b):
        out = np.zeros_like(a)
        fn = getattr(self.lib, "zk_k_fr_" + op)
        self._check(fn(self.ctx, u64p(out), u64p(a), u64p(b), ctypes.c_uint64(a.shape[0])), "zk_k_fr_" + op)
        return out

    def eq_table(self, n, r0, r1, alpha, beta):
        out = np.zeros((1 << n, 4), dtype=np.uint64)
        self._check(self.lib.zk_k_eq_table(self.ctx, u64p(out), ctypes.c_int32(n), u64p(r0), u64p(r1), u64p(alpha), u64p(beta)),
                    "zk_k_eq_table")
        return out

    def phi_table(self, rx, scale, n, inverse):
        cnt = (1 << n) if inverse else (1 << (n - 1))
        out = np.zeros((cnt, 4), dtype=np.uint64)
        self._check(self.lib.zk_k_phi_table(self.ctx, u64p(out), u64p(rx), u64p(scale), ctypes.c_int32(n), ctypes.c_int32(int(inverse))),
                    "zk_k_phi_table")
        return out

    def round_quadratic(self, V, M, r, first):
        """in-place on V / M; returns (coefficients (3,4), new length)"""
        out = np.zeros((3, 4), dtype=np.uint64)
        n_out = ctypes.c_uint64()
        self._check(self.lib.zk_k_round_quadratic(self.ctx, u64p(V), u64p(M), ctypes.c_uint64(V.shape[0]), u64p(r),
                                                  ctypes.c_int32(int(first)), u64p(out), ctypes.byref(n_out)), "zk_k_round_quadratic")
        return out, n_out.value

    def msm(self, scalars, bases):
        out = np.zeros(12, dtype=np.uint64)
        self._check(self.lib.zk_k_msm(self.ctx, u64p(out), u64p(scalars), u64p(bases), ctypes.c_uint64(scalars.shape[0])), "zk_k_msm")
        return out

    def fp_inv(self, a):
        """a: (n, 6) uint64 base-field elements in Montgomery form; element-wise inverse (0 -> 0)"""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        out = np.zeros_like(a)
        self._check(self.lib.zk_k_fp_inv(self.ctx, u64p(out), u64p(a), ctypes.c_uint64(a.shape[0])), "zk_k_fp_inv")
        return out

    def fpc_ops(self, a, b):
        """row-cooperative Fp arithmetic on (n, 6) Montgomery-form elements: (a * b, a + b, a - b)"""
        a = np.ascontiguousarray(a, dtype=np.uint64)
        b = np.ascontiguousarray(b, dtype=np.uint64)
        out = np.zeros((3,) + a.shape, dtype=np.uint64)
        self._check(self.lib.zk_k_fpc_ops(self.ctx, u64p(out), u64p(a), u64p(b), ctypes.c_uint64(a.shape[0])), "zk_k_fpc_ops")
        return out[0], out[1], out[2]

    def cl_add(self, p, q):
        """row-cooperative point arithmetic on (n, 18) Jacobian points: (p + q, 2 p)"""
        p = np.ascontiguousarray(p, dtype=np.uint64)
        q = np.ascontiguousarray(q, dtype=np.uint64)
        out = np.zeros((2,) + p.shape, dtype=np.uint64)
        self._check(self.lib.zk_k_cl_add(self.ctx, u64p(out), u64p(p), u64p(q), ctypes.c_uint64(p.shape[0])), "zk_k_cl_add")
        return out[0], out[1]

    def cl_tree(self, pts, segs, n_in=64):
        """sums of `segs` equal runs of the (n, 18) Jacobian points through the row-cooperative reduction tree"""
        pts = np.ascontiguousarray(pts, dtype=np.uint64)
        out = np.zeros((segs, 18), dtype=np.uint64)
        self._check(self.lib.zk_k_cl_tree(self.ctx, u64p(out), u64p(pts), ctypes.c_uint64(pts.shape[0] // segs), ctypes.c_uint64(segs), ctypes.c_uint32(n_in)), "zk_k_cl_tree")
        return out

    def witness_ntt(self, src, logn, inverse, count):
        """count transforms of length 2^logn (forward: half-length inputs zero padded; inverse: first half kept)"""
        length = 1 << logn
        out_len = length // 2 if inverse else length
        dst = np.zeros((count * out_len, 4), dtype=np.uint64)
        self._check(self.lib.zk_witness_ntt(self.ctx, u64p(dst), u64p(src), ctypes.c_int32(logn), ctypes.c_int32(int(inverse)),
                                            ctypes.c_uint64(count)), "zk_witness_ntt")
        return dst

    UNI_GATE = np.dtype([("g", "<u4"), ("u", "<u4"), ("lu", "u1"), ("sc", "u1"), ("pad", "u1", (2,))])               # zk_uni_gate
    BIN_GATE = np.dtype([("g", "<u4"), ("u", "<u4"), ("v", "<u4"), ("sc", "u1"), ("l", "u1"), ("pad", "u1", (2,))])   # zk_bin_gate

    def witness_input(self, offset, values):
        """(re)writes entries [offset, offset + len) of the device copy of layer 0 (zk_witness_input)"""
        values = np.ascontiguousarray(values, dtype=np.uint64)
        self._check(self.lib.zk_witness_input(self.ctx, ctypes.c_uint64(offset), u64p(values), ctypes.c_uint64(values.shape[0])),
                    "zk_witness_input")

    def witness_gates(self, n_out, uni, bin_, prev, two_mul, scale):
        """value of every gate of a generic layer (zk_witness_gates); uni / bin_ are UNI_GATE / BIN_GATE record arrays"""
        out = np.zeros((n_out, 4), dtype=np.uint64)
        uni = np.ascontiguousarray(uni, dtype=self.UNI_GATE)
        bin_ = np.ascontiguousarray(bin_, dtype=self.BIN_GATE)
        two_mul = np.ascontiguousarray(two_mul, dtype=np.uint64)
        pp = u64p(prev) if prev is not None else None
        self._check(self.lib.zk_witness_gates(self.ctx, u64p(out), ctypes.c_uint64(n_out), uni.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.c_uint64(uni.shape[0]), bin_.ctypes.data_as(ctypes.c_void_p),
                                              ctypes.c_uint64(bin_.shape[0]), pp, ctypes.c_uint64(0 if prev is None else prev.shape[0]),
                                              u64p(two_mul), ctypes.c_uint32(two_mul.shape[0]), u64p(scale)), "zk_witness_gates")
        return out

    def witness_dotprod(self, F, n_in, n_out, gates, fft_bl):
        """witness of a DOT_PROD layer (zk_witness_dotprod): out[(g, t)] = sum over the gates of g of F[(u, t)] * F[(v, t)]"""
        out = np.zeros((n_out << fft_bl, 4), dtype=np.uint64)
        F = np.ascontiguousarray(F, dtype=np.uint64)
        gates = np.ascontiguousarray(gates, dtype=self.BIN_GATE)
        self._check(self.lib.zk_witness_dotprod(self.ctx, u64p(out), ctypes.c_uint64(n_out), u64p(F), ctypes.c_uint64(n_in),
                                                gates.ctypes.data_as(ctypes.c_void_p), ctypes.c_uint64(gates.shape[0]),
                                                ctypes.c_int32(fft_bl)), "zk_witness_dotprod")
        return out

    def commit_rows(self, scalars, bases, rows, cols):
        out = np.zeros((rows, 12), dtype=np.uint64)
        self._check(self.lib.zk_k_commit_rows(self.ctx, u64p(out), u64p(scalars), u64p(bases), ctypes.c_uint64(rows), ctypes.c_uint64(cols)),
                    "zk_k_commit_rows")
        return out

    def profile(self, mask):
        self._check(self.lib.zk_profile_enable(self.ctx, ctypes.c_uint32(mask)), "zk_profile_enable")

    def profile_report(self, reset=True):
        import json
        buf = ctypes.create_string_buffer(8192)
        self._check(self.lib.zk_profile_report(self.ctx, buf, ctypes.c_uint64(8192), ctypes.c_int32(int(reset))), "zk_profile_report")
        return json.loads(buf.value.decode())

    def bench_fr_mul(self, n_threads, muls, iters=5):
        sec = ctypes.c_double()
        self._check(self.lib.zk_bench_fr_mul(self.ctx, ctypes.c_uint64(n_threads), ctypes.c_uint32(muls), ctypes.c_uint32(iters),
                                             ctypes.byref(sec)), "zk_bench_fr_mul")
        return sec.value

    def bench_copy(self, nbytes, iters=10):
        sec = ctypes.c_double()
        self._check(self.lib.zk_bench_copy(self.ctx, ctypes.c_uint64(nbytes), ctypes.c_uint32(iters), ctypes.byref(sec)), "zk_bench_copy")
        return sec.value

    def bench_round_quadratic(self, log_n, iters=10):
        sec, nbytes = ctypes.c_double(), ctypes.c_double()
        self._check(self.lib.zk_bench_round_quadratic(self.ctx, ctypes.c_uint32(log_n), ctypes.c_uint32(iters), ctypes.byref(sec),
                                                      ctypes.byref(nbytes)), "zk_bench_round_quadratic")
        return sec.value, nbytes.value


class _SessionBase:
    _prefix = None

    def _fn(self, name):
        return getattr(self.lib, self._prefix + name)

    def __init__(self, lib, model, pic=(32, 32, 1), pic_cnt=1, data_seed=20260928, device=0, statement=None, picture_seed=0, calibrated=None):
        """picture_seed=0: picture, weights and biases all come from the data_seed stream; otherwise the picture has its own stream, so
        that sessions with one data_seed share their weights and differ in the picture.
        statement=None: circuit + witness from the (synthetic) data. statement=[ints]: a verifier-only session whose circuit is
        rebuilt from the model descriptor and the quantisation scales of a prover's session (no witness; prove() fails)."""
        self.lib = lib
        self.model, self.pic, self.pic_cnt, self.data_seed = model, tuple(pic), pic_cnt, data_seed
        self.picture_seed = picture_seed
        self.desc = ModelDesc(model.encode(), pic[0], pic[1], pic[2], pic_cnt, data_seed, picture_seed)
        if calibrated is not None:
            # circuit + witness under GIVEN quantisation scales (the statement() of an earlier session of the model): fails if the picture does not fit them
            arr = (ctypes.c_int32 * max(len(calibrated), 1))(*calibrated)
            self._fn("session_create_calibrated").restype = ctypes.c_void_p
            self.h = self._fn("session_create_calibrated")(ctypes.byref(self.desc), arr, ctypes.c_uint64(len(calibrated)), ctypes.c_int32(device))
        elif statement is None:
            self._fn("session_create").restype = ctypes.c_void_p
            self.h = self._fn("session_create")(ctypes.byref(self.desc), ctypes.c_int32(device))
        else:
            arr = (ctypes.c_int32 * max(len(statement), 1))(*statement)
            self._fn("verifier_create").restype = ctypes.c_void_p
            self.h = self._fn("verifier_create")(ctypes.byref(self.desc), arr, ctypes.c_uint64(len(statement)), ctypes.c_int32(device))
        if not self.h:
            raise RuntimeError(f"{self._prefix}session_create({model}) failed")

    def statement(self):
        """quantisation scales the circuit's shape depends on (together with the model descriptor: the public statement)"""
        fn = self._fn("session_statement")
        fn.restype = ctypes.c_int64
        n = fn(ctypes.c_void_p(self.h), None, ctypes.c_uint64(0))
        arr = (ctypes.c_int32 * max(n, 1))()
        fn(ctypes.c_void_p(self.h), arr, ctypes.c_uint64(n))
        return [int(x) for x in arr[:n]]

    def prove(self, seed=None, mode=MODE_VERIFY, want_transcript=True):
        """one proof. seed=None (default): the verifier's challenges come from the operating system's CSPRNG, as in the reference.
        An integer seed asks for a REPRODUCIBLE run (MODE_SEEDED: parity tests, benches) -- predictable challenges, not secure."""
        if seed is not None:
            mode |= MODE_SEEDED
        else:
            seed = 0
        cap = 0
        if want_transcript:
            if getattr(self, "_tbuf", None) is None:
                self._tbuf = (ctypes.c_uint8 * (4 << 20))()       # re-used across proofs; grown if a proof is larger
            cap = len(self._tbuf)
        buf = self._tbuf if want_transcript else (ctypes.c_uint8 * 1)()
        res = Result()
        rc = self._fn("session_prove")(ctypes.c_void_p(self.h), ctypes.c_uint64(seed), ctypes.c_uint32(mode), buf,
                                       ctypes.c_uint64(cap), ctypes.byref(res))
        if rc != 0:
            raise RuntimeError(f"{self._prefix}session_prove failed ({rc}): {res.message.decode(errors='replace')}")
        if want_transcript and res.transcript_len > cap:
            self._tbuf = (ctypes.c_uint8 * int(res.transcript_len + (1 << 20)))()
            if not mode & (MODE_SEEDED | MODE_FIAT_SHAMIR):
                raise RuntimeError("transcript buffer too small for a proof with fresh randomness; call prove() again")
            return self.prove(seed, mode, True)                     # deterministic: the same seed gives the same proof
        data = ctypes.string_at(buf, res.transcript_len) if want_transcript else b""
        return res, data

    def verify(self, proof, seed=None, mode=MODE_VERIFY):
        """checks a serialized proof (bytes returned by prove) without the prover; returns the Result (accepted = 1 / 0).
        Only Fiat-Shamir proofs are proofs off line. An interactive transcript verifies only as a REPLAY against the seeded challenge
        stream it was driven with (pass that seed): a parity / debugging aid, not evidence -- the library rejects it otherwise."""
        if seed is not None:
            mode |= MODE_SEEDED
        else:
            seed = 0
        res = Result()
        buf = (ctypes.c_uint8 * max(len(proof), 1)).from_buffer_copy(proof if proof else b"\0")
        rc = self._fn("session_verify")(ctypes.c_void_p(self.h), ctypes.c_uint64(seed), ctypes.c_uint32(mode), buf,
                                        ctypes.c_uint64(len(proof)), ctypes.byref(res))
        if rc != 0:
            raise RuntimeError(f"{self._prefix}session_verify failed ({rc}): {res.message.decode(errors='replace')}")
        return res

    def layer_size(self, layer):
        """(size, layerType) of one layer of the circuit"""
        fn = self._fn("session_layer_size")
        fn.restype = ctypes.c_int64
        ty = ctypes.c_int32(0)
        return int(fn(ctypes.c_void_p(self.h), ctypes.c_int32(layer), ctypes.byref(ty))), int(ty.value)

    def poke(self, layer, index, value):
        """test hook: overwrite one witness value (4 x u64 Montgomery limbs) -- an invalid witness on purpose"""
        arr = (ctypes.c_uint64 * 4)(*value)
        rc = self._fn("session_poke")(ctypes.c_void_p(self.h), ctypes.c_int32(layer), ctypes.c_uint64(index), arr)
        if rc != 0:
            raise RuntimeError(f"{self._prefix}session_poke failed ({rc})")

    def conv_paths(self, force_field=-1):
        """test hook: (convolutions new_image evaluates from their tensors, those the last new_image ran in 64-bit integers, index in layer 0 of the first one's weights); force_field = 1 / 0
        keeps them all in field arithmetic / lets the device-side bounds decide from now on"""
        out = (ctypes.c_uint64 * 3)()
        rc = self._fn("session_conv_paths")(ctypes.c_void_p(self.h), ctypes.c_int32(force_field), out)
        if rc != 0:
            raise RuntimeError(f"{self._prefix}session_conv_paths failed ({rc})")
        return int(out[0]), int(out[1]), int(out[2])

    def row(self):
        buf = ctypes.create_string_buffer(1024)
        self._fn("session_row")(ctypes.c_void_p(self.h), buf, ctypes.c_uint64(1024))
        return buf.value.decode()

    def close(self):
        if self.h:
            self._fn("session_destroy")(ctypes.c_void_p(self.h))
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


PROFILE_CLASSES = ["eq_table", "gather", "gate_reduce", "gate_fixup", "gate_sum", "sum_partials", "round_quad", "round_cubic", "fold",
                   "matvec", "phi", "dot_prod", "liu_scatter", "msm_planes", "msm_finish", "msm_tables", "ipa", "misc", "round_tail", "round_fine"]


def sharing_stats():
    """what the sessions of this process share per GPU: resident circuits built / attached to, generator window tables / byte tables built
    (include/zkcnn_hip.h: zk_sharing_stats, zk_generator_table_stats)"""
    lib = hip_lib()
    v = [ctypes.c_uint64() for _ in range(4)]
    lib.zk_sharing_stats.restype = None
    lib.zk_generator_table_stats.restype = None
    lib.zk_sharing_stats(ctypes.byref(v[0]), ctypes.byref(v[1]))
    lib.zk_generator_table_stats(ctypes.byref(v[2]), ctypes.byref(v[3]))
    lib.zk_shared_circuit_bytes.restype = ctypes.c_uint64
    return {"shared_circuit_gb": round(lib.zk_shared_circuit_bytes() / 1e9, 3), "circuit_builds": v[0].value, "circuit_attaches": v[1].value, "window_table_builds": v[2].value, "byte_table_builds": v[3].value}


class Session(_SessionBase):
    """Circuit + witness resident on one GPU; prove() runs verifier <-> HIP prover (include/zkcnn_api.h)."""
    _prefix = "zkcnn_"

    def __init__(self, model, pic=(32, 32, 1), pic_cnt=1, data_seed=20260928, device=0, statement=None, picture_seed=0, calibrated=None):
        super().__init__(host_lib(), model, pic, pic_cnt, data_seed, device, statement, picture_seed, calibrated)

    def clone(self):
        """a second session on the same resident circuit with a copy of this session's witness and no host copy of the circuit
        (zkcnn_session_clone): nothing is generated, sorted or uploaded. Give it a picture of its own with new_image."""
        other = object.__new__(Session)
        other.lib, other.model, other.pic, other.pic_cnt, other.data_seed, other.picture_seed, other.desc = self.lib, self.model, self.pic, self.pic_cnt, self.data_seed, self.picture_seed, self.desc
        self.lib.zkcnn_session_clone.restype = ctypes.c_void_p
        other.h = self.lib.zkcnn_session_clone(ctypes.c_void_p(self.h))
        if not other.h:
            raise RuntimeError("zkcnn_session_clone failed")
        return other

    def new_image(self, picture_seed=None, pixels=None):
        """the next picture on the resident circuit (zkcnn_session_new_image): the synthetic picture of `picture_seed`, or `pixels`
        (channel, x, y order). Layer values and auxiliary witnesses are recomputed in HBM -- no circuit generation, no upload.
        Returns (code, ms): 0 = the session now proves this picture (every value range fits the circuit's scales: same transcript as
        Session(..., calibrated=self.statement()) built for it); 1 = its input range does not fit (nothing changed); 2 = an activation range
        does not fit: refused, the session keeps proving the picture it proved before."""
        ms = ctypes.c_double(0)
        if pixels is not None:
            arr = (ctypes.c_double * len(pixels))(*pixels)
            rc = self.lib.zkcnn_session_new_image(ctypes.c_void_p(self.h), ctypes.c_uint64(0), arr, ctypes.c_uint64(len(pixels)), ctypes.byref(ms))
        else:
            rc = self.lib.zkcnn_session_new_image(ctypes.c_void_p(self.h), ctypes.c_uint64(picture_seed), None, ctypes.c_uint64(0), ctypes.byref(ms))
        if rc < 0:
            raise RuntimeError(f"zkcnn_session_new_image failed ({rc})")
        if rc == 0 and pixels is None:
            self.picture_seed = picture_seed
        return rc, ms.value

    def structured_layers(self):
        """direct-convolution layers whose gate sums run in factored form (hint checked against the gate list at upload)"""
        return int(self.lib.zkcnn_session_structured_layers(ctypes.c_void_p(self.h)))

    def factored_dot_layers(self):
        """DOT_PROD layers (FFT convolutions over pic_cnt >= 2 pictures) whose phase-1 table is summed once over channel_out, not once per picture"""
        return int(self.lib.zkcnn_session_factored_dot_layers(ctypes.c_void_p(self.h)))

    def dot_deferred_phases(self):
        """DOT_PROD phases proved so far whose Y table was evaluated once behind X's live prefix instead of folded every round"""
        self.lib.zkcnn_session_dot_deferred_phases.restype = ctypes.c_uint64
        return int(self.lib.zkcnn_session_dot_deferred_phases(ctypes.c_void_p(self.h)))

    def synthetic_picture(self, picture_seed):
        n = self.pic[0] * self.pic[1] * self.pic[2]
        arr = (ctypes.c_double * n)()
        self.lib.zkcnn_session_synthetic_picture.restype = ctypes.c_int64
        got = self.lib.zkcnn_session_synthetic_picture(ctypes.c_void_p(self.h), ctypes.c_uint64(picture_seed), arr, ctypes.c_uint64(n))
        if got != n:
            raise RuntimeError("zkcnn_session_synthetic_picture failed")
        return list(arr)

    def profile(self, classes="all"):
        """HIP-event timing of the selected kernel classes (list of names, "all", or None to switch off)"""
        if classes is None:
            mask = 0
        elif classes == "all":
            mask = 0xFFFFFFFF
        else:
            mask = sum(1 << PROFILE_CLASSES.index(c) for c in classes)
        rc = self.lib.zkcnn_session_profile(ctypes.c_void_p(self.h), ctypes.c_uint32(mask))
        if rc != 0:
            raise RuntimeError("zkcnn_session_profile failed")

    def host_tail_rounds(self):
        """sumcheck rounds the hybrid tail has run on the host on this session so far (lanes of a batch: a phase's tables of <= 32 entries)"""
        r = ctypes.c_uint64()
        if self.lib.zkcnn_session_host_tail_rounds(ctypes.c_void_p(self.h), ctypes.byref(r)) != 0:
            raise RuntimeError("zkcnn_session_host_tail_rounds failed")
        return r.value

    def fs_stats(self):
        """(rounds, phases) the GPU has run by itself in Fiat-Shamir mode on this session so far"""
        r, p = ctypes.c_uint64(), ctypes.c_uint64()
        if self.lib.zkcnn_session_fs_stats(ctypes.c_void_p(self.h), ctypes.byref(r), ctypes.byref(p)) != 0:
            raise RuntimeError("zkcnn_session_fs_stats failed")
        return r.value, p.value

    def profile_report(self, reset=True):
        import json
        buf = ctypes.create_string_buffer(8192)
        rc = self.lib.zkcnn_session_profile_report(ctypes.c_void_p(self.h), buf, ctypes.c_uint64(8192), ctypes.c_int32(int(reset)))
        if rc != 0:
            raise RuntimeError("zkcnn_session_profile_report failed")
        return json.loads(buf.value.decode())


class BatchSession:
    """K sessions of one model on one GPU proving in lock step (include/zkcnn_api.h: zkcnn_batch_*): one host thread drives the K verifier
    loops, every sumcheck round of the K proofs is ONE kernel launch. The sessions must share their weights (same data_seed; the pictures
    differ by picture_seed or new_image), i.e. one resident circuit; they stay usable for new_image between batch proofs and are closed
    by their owner AFTER the batch."""

    def __init__(self, sessions, _handle=None):
        self.lib = host_lib()
        self.sessions = list(sessions)
        if _handle is None:
            arr = (ctypes.c_void_p * len(self.sessions))(*[s.h for s in self.sessions])
            self.lib.zkcnn_batch_create.restype = ctypes.c_void_p
            _handle = self.lib.zkcnn_batch_create(arr, ctypes.c_int32(len(self.sessions)))
            if not _handle:
                raise RuntimeError("zkcnn_batch_create failed (sessions of one model on one device, at most 8)")
        self.h = _handle
        self._bufs = None
        self.wall_s = 0.0

    @classmethod
    def group(cls, session_lists):
        """several batches at once (zkcnn_batch_create_group): every batch's stream exists before any lane is attached, so the streams spread evenly over the
        hardware queues -- batches that share a queue run their kernels one after the other, and the step time is the most loaded queue's"""
        lib = host_lib()
        lists = [list(x) for x in session_lists]
        flat = [s.h for x in lists for s in x]
        arr = (ctypes.c_void_p * len(flat))(*flat)
        counts = (ctypes.c_int32 * len(lists))(*[len(x) for x in lists])
        out = (ctypes.c_void_p * len(lists))()
        rc = lib.zkcnn_batch_create_group(arr, counts, ctypes.c_int32(len(lists)), out)
        if rc != 0:
            raise RuntimeError(f"zkcnn_batch_create_group failed ({rc})")
        return [cls(x, _handle=out[j]) for j, x in enumerate(lists)]

    def prove(self, seeds=None, mode=MODE_VERIFY, want_transcript=True):
        """one proof per lane; returns [(Result, transcript bytes)] in lane order. seeds: one integer per lane (reproducible, MODE_SEEDED)
        or None (operating system's CSPRNG)."""
        k = len(self.sessions)
        if seeds is not None:
            mode |= MODE_SEEDED
        seed_arr = (ctypes.c_uint64 * k)(*(seeds if seeds is not None else [0] * k))
        if want_transcript and self._bufs is None:
            self._bufs = [(ctypes.c_uint8 * (4 << 20))() for _ in range(k)]
        ptrs = (ctypes.c_void_p * k)(*[ctypes.addressof(b) if want_transcript else None for b in (self._bufs or [None] * k)])
        caps = (ctypes.c_uint64 * k)(*[len(b) if want_transcript else 0 for b in (self._bufs or [()] * k)])
        res = (Result * k)()
        wall = ctypes.c_double(0)
        rc = self.lib.zkcnn_batch_prove(ctypes.c_void_p(self.h), seed_arr, ctypes.c_uint32(mode), ptrs, caps, res, ctypes.byref(wall))
        if rc != 0:
            raise RuntimeError(f"zkcnn_batch_prove failed ({rc}): " + " | ".join(r.message.decode(errors="replace") for r in res))
        self.wall_s = wall.value
        out = []
        for i in range(k):
            if want_transcript and res[i].transcript_len > caps[i]:
                raise RuntimeError("transcript buffer too small")
            r = Result.from_buffer_copy(res[i])
            out.append((r, ctypes.string_at(self._bufs[i], r.transcript_len) if want_transcript else b""))
        return out

    def stats(self):
        v = (ctypes.c_uint64 * 5)()
        if self.lib.zkcnn_batch_stats(ctypes.c_void_p(self.h), v) != 0:
            raise RuntimeError("zkcnn_batch_stats failed")
        return {"fused_launches": int(v[0]), "lane_launches": int(v[1]), "flushes": int(v[2]), "lanes": int(v[3]), "driver_passes": int(v[4])}

    def close(self):
        if self.h:
            self.lib.zkcnn_batch_destroy(ctypes.c_void_p(self.h))
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
