"""Whole-proof parity on the GPU: the HIP prover driven by the verifier must (1) be accepted and (2) produce
the byte-identical canonical transcript as the CPU oracle prover under the same seeded challenge stream."""
import hashlib

import pytest

import zkcnn_amd
from tests import oracle_ffi

pytestmark = pytest.mark.gpu

CASES = [
    ("custom:F4", (4, 4, 1), 1),
    ("custom:F8 F4", (4, 4, 1), 1),
    ("custom:C2:3:1:s F4", (4, 4, 1), 1),
    ("custom:C2:3:1:s A F4", (4, 4, 1), 1),
    ("custom:C2:3:1:s M F4", (4, 4, 1), 1),
    ("custom:C2:3:1:n A F4", (4, 4, 2), 1),
    ("custom:C2:3:1:f F4", (4, 4, 1), 1),
    ("custom:C2:3:1:f M F4", (8, 8, 1), 2),
    ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2),
    ("lenet", (32, 32, 1), 1),
    ("custom:C3:3:1:f A C2:3:1:f M F6 F3", (8, 8, 1), 3),          # ragged batch (3 pictures), FFT conv + avg + max pool
    ("lenetCifar", (32, 32, 3), 1),                                 # three FFT conv blocks, 28 layers
    ("lenet.avg", (28, 28, 1), 1),                                  # MNIST-size input, padded first conv, average pooling
    ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),  # vgg11 at quarter width: 8e6 mul gates, 2^21 inputs
]


@pytest.mark.parametrize("model,pic,pp", CASES)
def test_gpu_transcript_identical_to_oracle(built, model, pic, pp):
    with zkcnn_amd.Session(model, pic, pp) as s:
        res, gpu = s.prove(seed=0x5EED0001)
        assert res.accepted == 1, res.message.decode()
        res2, gpu2 = s.prove(seed=0x5EED0002)          # same session, second proof, other challenges
        assert res2.accepted == 1 and gpu2 != gpu
        res3, gpu3 = s.prove(seed=0x5EED0001, mode=zkcnn_amd.MODE_DRIVE_ONLY)
        assert res3.accepted == -1 and gpu3 == gpu       # drive-only makes the same calls with the same challenges
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        ores, cpu = o.prove(seed=0x5EED0001)
        assert ores.accepted == 1
    assert len(gpu) == len(cpu) and hashlib.sha256(gpu).hexdigest() == hashlib.sha256(cpu).hexdigest()
    assert abs(res.proof_kb - ores.proof_kb) < 1e-9 and abs(res.poly_proof_kb - ores.poly_proof_kb) < 1e-9
