"""Byte parity with the CPU oracle at FULL size: vgg11 / CIFAR shape, pic_cnt = 1 -- BASELINE.json configs[2], the workload bench.py times.

The reduced-width circuits of the other GPU tests reach every kernel, but not the 2^24-entry tables of the real workload: the large-table
round kernel's live-prefix path with its fill / guard-quad switching, the 24 wide rows of the commitment riding as virtual rows of the 4096-column
MSM, the factored sums of 512-channel convolutions. Here the canonical transcripts of the full circuit are compared byte for byte
(reference loop being restated: src/prover.cpp:155-239, 396-426), in the plain interactive mode and in the modes bench.py times, plus one
INVALID witness at full size (a non-zero constraint row behind the first RELU: the measured live prefix moves to the end of a 2^21-entry table).

The oracle needs ~27 s of one host core per full-size proof (+ ~10 s for its gate-walking verifier): the three oracle runs go side by side in
three processes.
"""
import hashlib
import multiprocessing as mp

import numpy as np
import pytest

import zkcnn_amd

pytestmark = pytest.mark.gpu

MODEL, PIC, PP = "vgg11", (32, 32, 3), 1
REUSE, DRIVE = zkcnn_amd.MODE_REUSE_GENS, zkcnn_amd.MODE_DRIVE_ONLY
SEED, SEED_BAD = 0x5EED0001, 0x5EED0041
RELU = 4


def _oracle_job(job):
    """(mode, seed, poke or None) -> (accepted, sha256, length) of the oracle's transcript; runs in its own process"""
    from tests import oracle_ffi
    mode, seed, poke = job
    with oracle_ffi.OracleSession(MODEL, PIC, PP) as o:
        if poke is not None:
            o.poke(*poke)
        res, tr = o.prove(seed=seed, mode=mode)
    return int(res.accepted), hashlib.sha256(tr).hexdigest(), len(tr)


def test_full_size_vgg11_identical_to_oracle(built, oracle):
    one = [int(v) for v in oracle.from_canonical(np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]]
    with zkcnn_amd.Session(MODEL, PIC, PP) as s:
        # the first RELU layer (1.5e6 entries: outputs, then constraint rows that are zero for a valid witness); its last row gets a 1
        i, relu = 0, None
        while relu is None:
            size, ty = s.layer_size(i)
            assert size >= 0, "no RELU layer found"
            if ty == RELU:
                relu = (i, size - 1, one)
            i += 1
        jobs = [(0, SEED, None), (REUSE | DRIVE, SEED, None), (REUSE, SEED_BAD, relu)]
        with mp.get_context("spawn").Pool(3) as pool:
            pending = pool.map_async(_oracle_job, jobs)

            def digest(tr):
                return hashlib.sha256(tr).hexdigest(), len(tr)
            res, plain = s.prove(seed=SEED)                               # fresh generators drawn by the verifier, full verification
            assert res.accepted == 1, res.message.decode()
            res, first_use = s.prove(seed=SEED, mode=REUSE)               # session generators, first use: window / digit tables
            assert res.accepted == 1, res.message.decode()
            res, timed = s.prove(seed=SEED, mode=REUSE | DRIVE)           # second use: the byte table -- the path bench.py times
            assert res.accepted == -1
            assert first_use == timed, "drive-only transcript differs from the verified one"
            s.poke(*relu)
            bad_res, bad = s.prove(seed=SEED_BAD, mode=REUSE)
            (acc0, sha0, len0), (acc1, sha1, len1), (acc2, sha2, len2) = pending.get(timeout=900)
    assert acc0 == 1 and (sha0, len0) == digest(plain), "full-size vgg11, interactive mode: GPU transcript differs from the CPU oracle's"
    assert acc1 == -1 and (sha1, len1) == digest(timed), "full-size vgg11, REUSE_GENS | DRIVE_ONLY (bench mode): GPU transcript differs from the CPU oracle's"
    assert acc2 == 0 and bad_res.accepted == 0, "a non-zero constraint row at full size must be rejected (oracle and GPU)"
    assert (sha2, len2) == digest(bad), "full-size vgg11 with a corrupted witness: GPU transcript differs from the CPU oracle's"
