"""Host logic of the lock-step batch driver (zkcnn_amd/csrc/host/session.hpp: batchSessionT; fiber.hpp) without a GPU: K verifier loops
run as fibers of ONE thread, each against a CPU prover that gives up the thread at every round call (the points at which a GPU lane yields).
Whatever the interleaving, a lane's transcript must be what the same session produces alone -- the lanes share the thread, not their
challenge streams, Fiat-Shamir chains, private coins or result rows."""
import pytest

import zkcnn_amd
from tests import oracle_ffi

M = zkcnn_amd

LANES = [  # (model, picture shape, pictures per circuit, picture seed): different circuits on purpose -- the driver assumes nothing about lock step
    ("custom:C2:3:1:s M F4", (4, 4, 1), 1, 0),
    ("custom:C2:3:1:s M F4", (4, 4, 1), 1, 11),
    ("custom:C2:3:1:f M F4", (8, 8, 1), 2, 0),          # FFT convolution: cubic rounds, more rounds than the others
    ("custom:F8 F4", (4, 4, 1), 1, 5),
    ("custom:C2:3:1:s M F4", (4, 4, 1), 1, 12),
]


@pytest.fixture(scope="module")
def lanes(built):
    ss = [oracle_ffi.YieldingOracleSession(m, pic, pp, picture_seed=ps) for m, pic, pp, ps in LANES]
    yield ss
    for s in ss:
        s.close()


@pytest.mark.parametrize("mode", [M.MODE_SEEDED, M.MODE_SEEDED | M.MODE_REUSE_GENS | M.MODE_FULL_IPA, M.MODE_FIAT_SHAMIR,
                                  M.MODE_SEEDED | M.MODE_ZK, M.MODE_FIAT_SHAMIR | M.MODE_ZK | M.MODE_SEEDED])
@pytest.mark.parametrize("k", [1, 2, 3, 5])
def test_lanes_of_a_batch_prove_what_they_prove_alone(lanes, mode, k):
    seeds = [0x5EED0100 + 7 * i for i in range(k)]
    alone = [lanes[i].prove(seeds[i], mode) for i in range(k)]
    together, passes = oracle_ffi.oracle_batch_prove(lanes[:k], seeds, mode)
    for i in range(k):
        (ra, ta), (rb, tb) = alone[i], together[i]
        assert ra.accepted == 1 and rb.accepted == 1, (ra.message, rb.message)
        assert ta == tb, f"lane {i}: transcript differs from the session's own"
        assert rb.n_rounds == ra.n_rounds and rb.proof_kb == ra.proof_kb
    assert passes > max(r.n_rounds for r, _ in together)            # the lanes really were interleaved: at least one pass per round
    if k > 1 and not mode & M.MODE_FIAT_SHAMIR:
        assert len({t for _, t in together}) == k                   # and are different proofs


def test_a_rejected_lane_does_not_disturb_the_others(lanes):
    """every lane's verifier corrupts message 9 (the shared mode word): all reject -- each at its own pace, lanes ending early leave the
    others running -- and the next batch proof is clean again"""
    seeds = [3, 4, 5]
    bad, _ = oracle_ffi.oracle_batch_prove(lanes[:3], seeds, M.MODE_SEEDED | M.MODE_TAMPER | (9 << 8))
    assert [r.accepted for r, _ in bad] == [0, 0, 0]
    good, _ = oracle_ffi.oracle_batch_prove(lanes[:3], seeds, M.MODE_SEEDED)
    assert [r.accepted for r, _ in good] == [1, 1, 1]
    for i in range(3):
        assert good[i][1] == lanes[i].prove(seeds[i], M.MODE_SEEDED)[1]
