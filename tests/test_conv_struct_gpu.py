"""Factored gate sums of direct-convolution layers (hip/conv_kernels.cuh): the same field elements as the gate-by-gate sums of reference
src/prover.cpp:224-233 / 297-305, so every transcript must stay byte-identical to the CPU oracle's (which sums gate by gate)."""
import hashlib

import pytest

import zkcnn_amd
from tests import oracle_ffi

pytestmark = pytest.mark.gpu
REUSE = zkcnn_amd.MODE_REUSE_GENS

CASES = [
    # model, picture, pic_cnt, expected structured layers
    ("custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5", (8, 8, 2), 1, 2),           # power-of-two channels: conv 2 and 3 factor (conv 1 reads layer 0)
    ("custom:C4:3:1:s C8:3:1:s M C8:3:1:s F5", (8, 8, 2), 3, 2),           # three pictures in one circuit (p is not a power of two)
    ("custom:C4:3:0:s C4:5:2:s A F3", (18, 18, 1), 2, 1),                 # 3x3 without padding (18 -> 16), then 5x5 with padding 2 on 16x16
    ("custom:C3:3:1:s C6:3:1:s F4", (8, 8, 1), 1, 0),                     # channel counts that are not powers of two: generic path
    ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1, 7),     # vgg11 at quarter width
]


@pytest.mark.parametrize("model,pic,pp,n_struct", CASES)
def test_factored_conv_layers_identical_to_oracle(built, model, pic, pp, n_struct):
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        ores, want = o.prove(seed=0x5EED0031, mode=REUSE)
        assert ores.accepted == 1
        _, want2 = o.prove(seed=0x5EED0032)
    with zkcnn_amd.Session(model, pic, pp) as s:
        assert s.structured_layers() == n_struct
        res, got = s.prove(seed=0x5EED0031, mode=REUSE)
        assert res.accepted == 1, res.message.decode()
        assert hashlib.sha256(got).hexdigest() == hashlib.sha256(want).hexdigest()
        res, got2 = s.prove(seed=0x5EED0032, mode=zkcnn_amd.MODE_CROSS_PRED)
        assert res.accepted == 1 and got2 == want2


def test_full_vgg11_uses_the_factored_path(built):
    with zkcnn_amd.Session("vgg11", (32, 32, 3), 1) as s:
        assert s.structured_layers() == 7          # 8 convolutions; the first reads the picture from layer 0 and stays generic
        res, tr = s.prove(seed=0x5EED0033, mode=REUSE)
        assert res.accepted == 1, res.message.decode()
        n = res.n_messages
        bad, _ = s.prove(seed=0x5EED0033, mode=REUSE | zkcnn_amd.MODE_TAMPER | ((n // 2) << 8))
        assert bad.accepted == 0


@pytest.mark.parametrize("model,pic,pp,n_struct", [CASES[0], CASES[1], CASES[2], CASES[4]])
def test_convolution_parameters_are_found_without_hints(built, monkeypatch, model, pic, pp, n_struct):
    """an unmodified reference generator passes no hints: the upload infers channel counts, sides, kernel size, padding and stride from
    the gate list (and checks them against every gate) -- same layers factored, same transcript"""
    with zkcnn_amd.Session(model, pic, pp) as s:
        _, want = s.prove(seed=0x5EED0035, mode=REUSE)
    monkeypatch.setenv("ZKCNN_CONV_HINTS", "0")
    with zkcnn_amd.Session(model, pic, pp) as s:
        assert s.structured_layers() == n_struct
        res, got = s.prove(seed=0x5EED0035, mode=REUSE)
        assert res.accepted == 1 and got == want


def test_full_vgg16_factored_and_live_prefix_paths(built):
    """vgg16 / CIFAR shape, pic_cnt = 1 (13 direct convolutions, layer 0 = 2^25 entries): twelve convolutions factor; acceptance, rejection of a
    corrupted message, a Fiat-Shamir proof that verifies off line, and the GPU predicates agreeing with the host loops"""
    with zkcnn_amd.Session("vgg16", (32, 32, 3), 1) as s:
        assert s.structured_layers() == 12
        res, tr = s.prove(seed=0x5EED0037, mode=REUSE)
        assert res.accepted == 1, res.message.decode()
        assert res.input_bits == 25
        bad, _ = s.prove(seed=0x5EED0037, mode=REUSE | zkcnn_amd.MODE_TAMPER | ((res.n_messages // 3) << 8))
        assert bad.accepted == 0
        r2, tr2 = s.prove(seed=0x5EED0037, mode=REUSE | zkcnn_amd.MODE_DRIVE_ONLY)
        assert tr2 == tr
        res, proof = s.prove(mode=zkcnn_amd.MODE_FIAT_SHAMIR)
        assert res.accepted == 1 and s.verify(proof, mode=zkcnn_amd.MODE_FIAT_SHAMIR).accepted == 1
        cross, _ = s.prove(seed=0x5EED0038, mode=zkcnn_amd.MODE_CROSS_PRED)
        assert cross.accepted == 1, cross.message.decode()
