"""Parity of the path bench.py actually times, and of the BASELINE configurations the round-1 suite did not reach.

* bench.py proves with MODE_REUSE_GENS | MODE_DRIVE_ONLY: from the second proof of a session on, the commitment and every
  inner-product round go through the byte-table MSM kernels (k_msm_bytes / k_tree_reduce / k_ipa_scalars with index rows). Here the
  transcripts of that path are compared byte for byte with the CPU oracle run in the same mode on the same generators.
* BASELINE.json configs[3] / [4] per-GPU workloads at full size (vgg11 pic_cnt=8 as one circuit, vgg16 pic_cnt=4): acceptance, rejection
  of a corrupted message, determinism, GPU predicates == host predicates; oracle parity on reduced-width circuits of the same shape
  whose FFT-layer tables are large enough for the multi-pair branch of the cubic round kernel.
* the RCCL gather of zkcnn_amd/dp.py with the `nccl` backend on one rank; the DOT_PROD witness kernel against the oracle's field ops.
"""
import hashlib
import os
import socket

import numpy as np
import pytest

import zkcnn_amd
from tests import oracle_ffi

pytestmark = pytest.mark.gpu

REUSE, DRIVE, CROSS, TAMPER, FULL_IPA = (zkcnn_amd.MODE_REUSE_GENS, zkcnn_amd.MODE_DRIVE_ONLY, zkcnn_amd.MODE_CROSS_PRED, zkcnn_amd.MODE_TAMPER,
                                           zkcnn_amd.MODE_FULL_IPA)

TIMED_CASES = [
    ("lenet", (32, 32, 1), 1),
    ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2),      # FFT-conv block: pad -> FFT -> dot -> IFFT -> bias
    ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),        # vgg11 at quarter width
]


@pytest.mark.parametrize("model,pic,pp", TIMED_CASES)
def test_reused_generators_byte_table_path_identical_to_oracle(built, model, pic, pp):
    """one session, four proofs on the session's generators: the first builds window / digit tables, from the second on the full byte
    table serves the commitment and all IPA rounds (what bench.py times); then the same seeds drive-only. Every transcript must equal
    the oracle's under the same mode bits."""
    seeds = [0x5EED0001, 0x5EED0002, 0x5EED0003]
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        want = []
        for sd in seeds:
            ores, tr = o.prove(seed=sd, mode=REUSE)
            assert ores.accepted == 1
            want.append(tr)
    with zkcnn_amd.Session(model, pic, pp) as s:
        for k, sd in enumerate(seeds):
            res, tr = s.prove(seed=sd, mode=REUSE)
            assert res.accepted == 1, res.message.decode()
            assert tr == want[k], f"proof {k} (seed {sd:#x}) with re-used generators differs from the oracle"
        for k, sd in enumerate(seeds):                    # bench mode: same calls, same challenges, verifier checks skipped
            res, tr = s.prove(seed=sd, mode=REUSE | DRIVE)
            assert res.accepted == -1 and tr == want[k], f"drive-only proof {k} differs"
        # the inner-product argument run down to length 1 (bench.py's conservative companion), on the byte table and on fresh generators
        res, full_reuse = s.prove(seed=seeds[1], mode=REUSE | FULL_IPA)
        assert res.accepted == 1 and len(full_reuse) != len(want[1])
        # fresh generators on a session whose byte table is live: the cache must notice the change
        res, tr = s.prove(seed=seeds[0])
        assert res.accepted == 1
        res, full_fresh = s.prove(seed=seeds[2], mode=FULL_IPA)
        assert res.accepted == 1
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        assert o.prove(seed=seeds[0])[1] == tr
        assert o.prove(seed=seeds[1], mode=REUSE | FULL_IPA)[1] == full_reuse
        assert o.prove(seed=seeds[2], mode=FULL_IPA)[1] == full_fresh


REDUCED = [
    # same topology as vgg16 (13 conv, 5 pool, 3 fc; 99 layers with FFT convolutions), every width / 16, four pictures in one circuit
    ("vgg:4 4 M 8 8 M 16 16 16 M 32 32 32 M 32 32 32 M", (32, 32, 3), 4),
    # two 3x3 FFT convolutions on 32x32 pictures, 4 pictures: the second FFT layer has 2^21 entries, so the cubic rounds of its
    # DOT_PROD layer run the grid-stride (several pairs per thread) branch of k_round_cubic
    ("custom:C16:3:1:f C16:3:1:f M F10", (32, 32, 3), 4),
]


@pytest.mark.parametrize("model,pic,pp", REDUCED)
def test_fft_conv_batches_identical_to_oracle(built, model, pic, pp):
    with zkcnn_amd.Session(model, pic, pp) as s:
        res, gpu = s.prove(seed=0x5EED0001)
        assert res.accepted == 1, res.message.decode()
        res2, gpu2 = s.prove(seed=0x5EED0001, mode=REUSE)
        res3, gpu3 = s.prove(seed=0x5EED0001, mode=REUSE | DRIVE)
        assert res2.accepted == 1 and gpu3 == gpu2
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        ores, cpu = o.prove(seed=0x5EED0001)
        assert ores.accepted == 1
        _, cpu2 = o.prove(seed=0x5EED0001, mode=REUSE | DRIVE)
    assert hashlib.sha256(gpu).hexdigest() == hashlib.sha256(cpu).hexdigest()
    assert hashlib.sha256(gpu2).hexdigest() == hashlib.sha256(cpu2).hexdigest()



def test_dot_prod_dead_region_is_folded_in_one_pass(built, monkeypatch):
    """Round 5: a DOT_PROD phase leaves Y unfolded behind X's live prefix while that prefix stays even and catches up in one pass (k_dead_rows_f) before
    the tables are small -- transcripts must not move (reference src/prover.cpp:103-153 folds every entry in every round). With the library's thresholds
    only the full-size circuits get there (test_full_size_fft_conv_circuits); the hook lowers the fill threshold so that this 2^21-entry phase skips
    several folds (s >= 6: the chunked branch of the kernel) and, with a higher one, few (s < 6: the butterfly branch)."""
    model, pic, pp = "custom:C16:3:1:f C16:3:1:f M F10", (32, 32, 3), 4
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        _, cpu = o.prove(seed=0x5EED0021, mode=REUSE | DRIVE)
    monkeypatch.setenv("ZKCNN_TEST_HOOKS", "1")
    for fill_log, batch in ((6, False), (17, False), (12, True)):
        monkeypatch.setenv("ZKCNN_TEST_DOT_FILL_LOG", str(fill_log))
        if not batch:
            with zkcnn_amd.Session(model, pic, pp) as s:
                res, gpu = s.prove(seed=0x5EED0021, mode=REUSE)
                assert res.accepted == 1, res.message.decode()
                assert s.dot_deferred_phases() >= 1, "no DOT_PROD phase took the deferred fold"
                assert gpu == cpu, f"fill threshold 2^{fill_log}: transcript differs from the oracle's"
        else:
            ss = [zkcnn_amd.Session(model, pic, pp) for _ in range(2)]
            try:
                with zkcnn_amd.BatchSession(ss) as b:
                    out = b.prove(seeds=[0x5EED0021, 0x5EED0021], mode=REUSE | DRIVE)
                for res, tr in out:
                    assert tr == cpu
                assert ss[0].dot_deferred_phases() >= 1
            finally:
                for x in ss:
                    x.close()


@pytest.mark.parametrize("model,pic,pp,must_defer", [("custom:C64:3:1:f F10", (48, 48, 3), 1, True), ("custom:C24:3:1:f F10", (40, 40, 5), 1, False),
                                                     ("custom:C10:3:1:f F10", (32, 32, 6), 3, False)])
def test_dot_prod_deferred_fold_when_the_live_prefix_stops_being_a_multiple_of_four(built, monkeypatch, model, pic, pp, must_defer):
    """Round-5 advisor finding: the cubic round kernel folds QUADS, so a deferred Y (read as 0 behind X's live prefix) is only right while that prefix is a
    multiple of 4, not merely even -- with prefixes like 6 = (pictures + output channels) x input channels x 2^-k the last live quad reached behind it and
    claim_1 = Y(u) came out wrong (the verifier rejects an honest proof). Shapes whose prefix has a small 2-adic valuation, every fill threshold the hook
    allows: accepted, and the bytes of the oracle."""
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        _, cpu = o.prove(seed=0x5EED0022, mode=REUSE | DRIVE)
    monkeypatch.setenv("ZKCNN_TEST_HOOKS", "1")
    deferred = 0
    for fill_log in (4, 7, 10, 13):
        monkeypatch.setenv("ZKCNN_TEST_DOT_FILL_LOG", str(fill_log))
        with zkcnn_amd.Session(model, pic, pp) as s:
            res, gpu = s.prove(seed=0x5EED0022, mode=REUSE)
            assert res.accepted == 1, f"fill threshold 2^{fill_log}: {res.message.decode()}"
            assert gpu == cpu, f"fill threshold 2^{fill_log}: transcript differs from the oracle's"
            deferred += s.dot_deferred_phases()
    assert deferred >= 1 or not must_defer, "no DOT_PROD phase took the deferred fold: the test does not reach the code it is about"


FULL = [
    ("vgg16", (32, 32, 3), 4, dict(n_layers=99, input_bits=26)),      # BASELINE configs[4]: the per-GPU workload (4 images per GPU)
    ("vgg11", (32, 32, 3), 8, dict(n_layers=69, input_bits=26)),      # configs[3] as the reference would fold it (one circuit, pic_cnt=8)
]


FULL_GOLDEN = __import__("json").load(open(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "full_size.json")))["full_size"]


@pytest.mark.parametrize("model,pic,pp,shape", FULL)
def test_full_size_fft_conv_circuits(built, model, pic, pp, shape):
    """BYTE parity at full size (layer 0 = 2^26 entries, cubic rounds on 2^26-entry tables) through committed fixtures: the CPU oracle's transcripts
    of these two circuits take ~10 minutes and ~30 GB each, so their SHA-256 and length are kept in tests/golden/full_size.json (made by
    tests/golden/make_golden_full.py) -- plus the size-independent properties: the verifier checks the round identity in every round, every wiring
    predicate (on the GPU and, cross-checked, on the host) and the Hyrax opening; corrupted messages are rejected."""
    golden = FULL_GOLDEN.get(f"{model}_pp{pp}")
    with zkcnn_amd.Session(model, pic, pp) as s:
        res, t1 = s.prove(seed=0x5EED0001)
        assert res.accepted == 1, res.message.decode()
        assert res.n_layers == shape["n_layers"] and res.input_bits == shape["input_bits"]
        if golden:
            g = golden["interactive"]
            assert g["challenge_seed"] == 0x5EED0001 and (hashlib.sha256(t1).hexdigest(), len(t1)) == (g["sha256"], g["transcript_len"]), \
                f"{model} pic_cnt={pp} at full size: GPU transcript differs from the CPU oracle's (tests/golden/full_size.json)"
        r2, t2 = s.prove(seed=0x5EED0001, mode=DRIVE)
        assert t2 == t1                                           # deterministic, drive-only makes the same calls
        r3, t3 = s.prove(seed=0x5EED0007, mode=REUSE)
        r4, t4 = s.prove(seed=0x5EED0007, mode=REUSE | DRIVE)      # byte-table path at full size
        assert r3.accepted == 1 and t4 == t3
        if golden:
            g = golden["reuse_gens"]
            assert g["challenge_seed"] == 0x5EED0007 and (hashlib.sha256(t3).hexdigest(), len(t3)) == (g["sha256"], g["transcript_len"]), \
                f"{model} pic_cnt={pp} at full size, public generators: GPU transcript differs from the CPU oracle's"
        assert s.verify(t3, seed=0x5EED0007, mode=REUSE).accepted == 1
        n = res.n_messages
        for k in (n // 3, n - 1, n + 2):                          # a sumcheck message in the middle, the last one, an opening message
            bad, _ = s.prove(seed=0x5EED0001, mode=TAMPER | (k << 8))
            assert bad.accepted == 0, f"message {k} corrupted but accepted"
        cross, _ = s.prove(seed=0x5EED0004, mode=CROSS)
        assert cross.accepted == 1, cross.message.decode()


def test_dot_prod_witness_kernel_matches_oracle(hip, oracle):
    """zk_witness_dotprod (k_dot_witness; reference src/neuralNetwork.cpp:937-948, calcDotProdLayer) against the oracle's field
    multiplication / addition applied gate by gate; includes an output vector without gates and repeated (u, v) pairs"""
    rng = np.random.default_rng(3)
    fft_bl, n_in, n_out, n_gates = 5, 13, 9, 60
    length = 1 << fft_bl
    F = oracle.random(n_in * length, 901)
    gates = np.zeros(n_gates, dtype=zkcnn_amd.HipContext.BIN_GATE)
    gates["g"] = rng.choice(np.array([0, 1, 2, 3, 5, 6, 7, 8]), size=n_gates)        # output 4 has no gate
    gates["u"] = rng.integers(0, n_in, size=n_gates)
    gates["v"] = rng.integers(0, n_in, size=n_gates)
    gates[7] = gates[3]                                                           # a repeated gate counts twice
    got = hip.witness_dotprod(F, n_in, n_out, gates, fft_bl)
    want = np.zeros((n_out * length, 4), dtype=np.uint64)
    for gt in gates:
        g, u, v = int(gt["g"]), int(gt["u"]), int(gt["v"])
        prod = oracle.binop("mul", np.ascontiguousarray(F[u * length:(u + 1) * length]), np.ascontiguousarray(F[v * length:(v + 1) * length]))
        want[g * length:(g + 1) * length] = oracle.binop("add", np.ascontiguousarray(want[g * length:(g + 1) * length]), prod)
    assert np.array_equal(got, want)
    assert not got[4 * length:5 * length].any()
    bad = gates.copy()
    bad["u"][0] = n_in
    with pytest.raises(RuntimeError):
        hip.witness_dotprod(F, n_in, n_out, bad, fft_bl)


def test_rccl_gather_single_rank(built):
    """the step's only collective on the real backend: `nccl` (= RCCL) process group of one rank on cuda:0, AsyncGather of packed
    transcripts produced by the GPU prover, unpacked proofs verify. (N > 1 ranks: tests/test_dp_cpu.py with gloo; the driver's scaling run.)"""
    import torch
    import torch.distributed as dist
    from zkcnn_amd import dp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        with zkcnn_amd.Session("custom:C2:3:1:f M F4", (8, 8, 1), 2) as s:
            proofs = [s.prove(seed=70 + k, mode=REUSE | DRIVE)[1] for k in range(3)]
            g = dp.AsyncGather(dist, "cuda", 1 << 19)
            for k, tr in enumerate(proofs):
                g.submit(0, dp.pack([(k, tr)]))
            got = g.wait()
            assert len(got) == 3
            for k, step in enumerate(got):
                (rank, blob), = step
                (img, tr), = dp.unpack(blob)
                assert rank == 0 and img == k and tr == proofs[k]
                assert s.verify(tr, seed=70 + k, mode=REUSE).accepted == 1
        assert dp.gather_proofs([(1, b"b"), (0, b"a")], dist, "cuda") == [(0, b"a"), (1, b"b")]
        dist.barrier()
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()
