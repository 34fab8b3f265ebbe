"""Zero-knowledge mode on the GPU: under a seed (challenges and prover coins reproducible) the HIP prover's
transcript -- blinded row commitments (k_msm_codes + blinding column through the window kernels), masked round polynomials, the proofs of
dot product whose commitments go through zk_commit_vector -- is byte-identical to the CPU oracle's; full-size vgg11 is accepted."""
import pytest

import zkcnn_amd
from tests import oracle_ffi

pytestmark = pytest.mark.gpu
ZK, FS, REUSE, DRIVE, TAMPER = zkcnn_amd.MODE_ZK, zkcnn_amd.MODE_FIAT_SHAMIR, zkcnn_amd.MODE_REUSE_GENS, zkcnn_amd.MODE_DRIVE_ONLY, zkcnn_amd.MODE_TAMPER

CASES = [
    ("custom:F8 F4", (4, 4, 1), 1),
    ("custom:C2:3:1:f M F4", (8, 8, 1), 2),
    ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2),
    ("lenet", (32, 32, 1), 1),
    ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),
]


@pytest.mark.parametrize("model,pic,pp", CASES)
def test_zero_knowledge_transcripts_identical_to_oracle(built, model, pic, pp):
    modes = [ZK, ZK | REUSE, ZK | REUSE, ZK | FS]          # REUSE twice: the second proof runs on the byte table of (g, H)
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        want = [o.prove(seed=0x5EED0001 + k, mode=m)[1] for k, m in enumerate(modes)]
    with zkcnn_amd.Session(model, pic, pp) as s:
        for k, m in enumerate(modes):
            res, tr = s.prove(seed=0x5EED0001 + k, mode=m)
            assert res.accepted == 1, res.message.decode()
            assert tr == want[k], f"zero-knowledge transcript {k} (mode {m:#x}) differs from the oracle"
        res, tr = s.prove(seed=0x5EED0002, mode=ZK | REUSE | DRIVE)
        assert res.accepted == -1 and tr == want[1]
        assert s.verify(want[3], mode=ZK | FS).accepted == 1
        plain, t_plain = s.prove(seed=0x5EED0001)             # the plain mode still works on the same session afterwards
        assert plain.accepted == 1
        n = res.n_messages
        for k in (n // 2, n - 1, n, n + 1, n + 2):
            bad, _ = s.prove(seed=3, mode=ZK | TAMPER | (k << 8))
            assert bad.accepted == 0, k
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        assert o.prove(seed=0x5EED0001)[1] == t_plain


def test_every_message_of_the_gpu_prover_depends_on_the_coins(built, monkeypatch):
    """the product path's twin of tests/test_zk_cpu.py::test_every_message_depends_on_the_coins: the same challenges with other prover coins
    (ZKCNN_TEST_COIN_SALT) change every commitment, round polynomial, evaluation claim (V + Z M: zk_sumcheck_tail_pairs / zk_sumcheck_claims_adjust)
    and revealed value -- what stays is the public output's value and the zero claims of operands a layer does not have"""
    model, pic, pp = CASES[2]
    with zkcnn_amd.Session(model, pic, pp) as s:
        monkeypatch.setenv("ZKCNN_TEST_COIN_SALT", "0")
        r1, t1 = s.prove(seed=21, mode=ZK | REUSE)
        monkeypatch.setenv("ZKCNN_TEST_COIN_SALT", "777")
        r2, t2 = s.prove(seed=21, mode=ZK | REUSE)
        r3, t3 = s.prove(seed=21, mode=ZK | REUSE)
        assert r1.accepted == 1 and r2.accepted == 1 and t2 == t3 and len(t1) == len(t2)
        same = [k for k in range(0, len(t1), 16) if t1[k:k + 16] == t2[k:k + 16] and any(t1[k:k + 16])]
        assert len(same) <= 2, (len(same), len(t1) // 16)


def test_gpu_transcripts_pass_the_independent_python_verifier(built):
    """tests/test_verifier_python_cpu.py restates the verifier (the reference's loop + the zero-knowledge extension) in Python; here it reads what the HIP
    prover wrote: a fully connected model and an FFT convolution over two pictures, plain protocol and zero-knowledge mode"""
    from tests.test_verifier_python_cpu import python_verify
    orc = oracle_ffi.load()
    for model, pic, pp in (("custom:F8 F4", (4, 4, 1), 1), ("custom:C2:3:1:f M F4", (4, 4, 1), 2)):
        with zkcnn_amd.Session(model, pic, pp) as s, oracle_ffi.OracleSession(model, pic, pp) as o:       # (the oracle session: the circuit's structure as data)
            for zk in (False, True):
                mode = (ZK if zk else 0) | REUSE
                res, tr = s.prove(seed=0x5EED0050, mode=mode)
                assert res.accepted == 1
                assert python_verify(orc, o, tr, 0x5EED0050, res.n_layers, zk=zk) == res.n_rounds + 1


def test_full_size_vgg11_zero_knowledge(built):
    with zkcnn_amd.Session("vgg11", (32, 32, 3), 1) as s:
        a, ta = s.prove(mode=ZK | REUSE)                      # OS randomness
        b, tb = s.prove(mode=ZK | REUSE)
        assert a.accepted == 1 and b.accepted == 1 and ta[:4096 * 48] != tb[:4096 * 48]
        c, tc = s.prove(seed=9, mode=ZK | REUSE | DRIVE)
        d, td = s.prove(seed=9, mode=ZK | REUSE | DRIVE)
        assert tc == td
        print(f"vgg11 zero-knowledge: prover {1e3 * (d.prove_s + d.poly_prove_s):.1f} ms (sumcheck {1e3 * d.prove_s:.1f} + commitments / openings "
              f"{1e3 * d.poly_prove_s:.1f}), proof {d.proof_kb + d.poly_proof_kb:.0f} KB")
        assert s.verify(tc, seed=9, mode=ZK | REUSE).accepted == 1


def test_zero_knowledge_refuses_another_tail_policy(built):
    """Round-5 advisor finding: the zero-knowledge mode fixes the host tail itself (the masks' share of a phase's last round polynomial is added on the
    host from the last table pairs); ZKCNN_MODE_GPU_TAIL ("every round a kernel") in the same proof used to be ignored silently -- the call is refused,
    for a session on its own and for the lanes of a batch, and the session proves on afterwards; ZKCNN_MODE_HOST_TAIL next to it changes nothing."""
    model, pic, pp = CASES[1]
    with zkcnn_amd.Session(model, pic, pp) as s:
        with pytest.raises(RuntimeError, match="host tail"):
            s.prove(seed=5, mode=ZK | REUSE | zkcnn_amd.MODE_GPU_TAIL)
        res, tr = s.prove(seed=5, mode=ZK | REUSE)
        assert res.accepted == 1
        res2, tr2 = s.prove(seed=5, mode=ZK | REUSE | zkcnn_amd.MODE_HOST_TAIL)      # a host tail either way: accepted, the same bytes
        assert res2.accepted == 1 and tr2 == tr
        with zkcnn_amd.Session(model, pic, pp) as s2, zkcnn_amd.BatchSession([s, s2]) as b:
            with pytest.raises(RuntimeError, match="host tail"):
                b.prove(seeds=[5, 6], mode=ZK | REUSE | DRIVE | zkcnn_amd.MODE_GPU_TAIL)
            out = b.prove(seeds=[5, 6], mode=ZK | REUSE)
            assert all(r.accepted == 1 for r, _ in out)
