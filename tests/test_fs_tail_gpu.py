"""Device-side Fiat-Shamir rounds (hip/fs_tail.cuh): with ZKCNN_MODE_FS_DEVICE the GPU runs the small rounds of every
phase of a non-interactive proof by itself -- fold, round sums, add_term bookkeeping, BLAKE2s chain step, next challenge. The transcript must be
the one of the default mode (challenges derived on the host, rounds in the resident kernels), of the host-driven rounds
(ZKCNN_MODE_HOST_ROUNDS) and of the CPU oracle, whose verifier derives every challenge on the host."""
import hashlib
import time

import pytest

import zkcnn_amd
from tests import oracle_ffi

pytestmark = pytest.mark.gpu
FS, HOST, REUSE, DRIVE = zkcnn_amd.MODE_FIAT_SHAMIR, zkcnn_amd.MODE_HOST_ROUNDS, zkcnn_amd.MODE_REUSE_GENS, zkcnn_amd.MODE_DRIVE_ONLY
FSD = FS | zkcnn_amd.MODE_FS_DEVICE

CASES = [
    ("custom:F8 F4", (4, 4, 1), 1),                                   # every table tiny: whole phases on the device, from round 0
    ("custom:C2:3:1:s M F4", (4, 4, 1), 1),
    ("custom:C2:3:1:n A F4", (4, 4, 2), 1),
    ("custom:C2:3:0:f C3:3:1:f M C2:3:1:s A F5 F3", (10, 10, 2), 2),   # FFT / DOT_PROD layers: cubic rounds stay on the host, the rest not
    ("lenet", (32, 32, 1), 1),                                         # 2^18-entry layer 0: the tail starts in the middle of the Liu sumcheck
    ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),
]


@pytest.mark.parametrize("model,pic,pp", CASES)
def test_device_rounds_give_the_host_transcript(built, model, pic, pp):
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        ores, want = o.prove(mode=FS)
        assert ores.accepted == 1
    with zkcnn_amd.Session(model, pic, pp) as s:
        r0, p0 = s.fs_stats()
        res, tr = s.prove(mode=FSD)
        r1, p1 = s.fs_stats()
        assert res.accepted == 1, res.message.decode()
        assert tr == want, "device-side rounds changed the Fiat-Shamir transcript"
        assert p1 > p0 and r1 - r0 >= p1 - p0, "no phase ran on the device"
        resd, trd = s.prove(mode=FS)                              # default: host-derived challenges, resident round kernels
        r1, p1 = s.fs_stats()
        assert resd.accepted == 1 and trd == want
        res2, tr2 = s.prove(mode=FS | HOST)
        assert s.fs_stats() == (r1, p1) and tr2 == want          # every round driven from the host: same bytes
        res3, tr3 = s.prove(mode=FS | DRIVE)
        assert res3.accepted == -1 and tr3 == want
        assert s.verify(tr, mode=FS).accepted == 1
        print(f"{model}: {r1 - r0} of {res.n_rounds} rounds in {p1 - p0} device phases")


@pytest.mark.parametrize("model,pic,pp", CASES)
def test_resident_round_kernel_equals_launch_per_round(built, model, pic, pp):
    """Interactive protocol: by default the small rounds of every phase run in ONE resident kernel that trades polynomials and challenges with
    the host through mailboxes (k_tail<true>); ZKCNN_MODE_HOST_ROUNDS launches a kernel per round instead. Same transcripts, equal to the
    oracle's; a verifier that rejects in the middle of a phase sends the resident kernel home and the session keeps working."""
    with oracle_ffi.OracleSession(model, pic, pp) as o:
        _, want = o.prove(seed=0x5EED0051, mode=REUSE)
        _, want_zk = o.prove(seed=0x5EED0052, mode=REUSE | zkcnn_amd.MODE_ZK)
    with zkcnn_amd.Session(model, pic, pp) as s:
        r0, p0 = s.fs_stats()
        res, tr = s.prove(seed=0x5EED0051, mode=REUSE)
        r1, p1 = s.fs_stats()
        assert res.accepted == 1, res.message.decode()
        assert tr == want, "resident round kernel: transcript differs from the oracle's"
        assert r1 > r0 and p1 > p0, "no round ran in the resident kernel"
        res2, tr2 = s.prove(seed=0x5EED0051, mode=REUSE | HOST)
        assert s.fs_stats() == (r1, p1), "ZKCNN_MODE_HOST_ROUNDS must keep every round a launch"
        assert res2.accepted == 1 and tr2 == want
        # masked round polynomials (the masks are added on the host) on top of the resident kernel
        res3, tr3 = s.prove(seed=0x5EED0052, mode=REUSE | zkcnn_amd.MODE_ZK)
        assert res3.accepted == 1 and tr3 == want_zk
        # a corrupted message in the middle of the proof: the verifier stops calling in the middle of a phase
        n, rejected = res.n_messages, 0
        for k in (n // 5, n // 3, n // 2, n - 5):
            bad, _ = s.prove(seed=0x5EED0051, mode=REUSE | zkcnn_amd.MODE_TAMPER | (k << 8))
            rejected += bad.accepted == 0           # (a few messages are never read back: the bound on them is tests/test_zk_cpu.py's)
            again, tr4 = s.prove(seed=0x5EED0051, mode=REUSE | DRIVE)
            assert again.accepted == -1 and tr4 == want
        assert rejected >= 2
        print(f"{model}: {r1 - r0} of {res.n_rounds} rounds in {p1 - p0} resident kernels")


def test_full_size_vgg11_fiat_shamir_latency(built):
    with zkcnn_amd.Session("vgg11", (32, 32, 3), 1) as s:
        s.prove(mode=FS | DRIVE, want_transcript=False)           # tables of the public generators
        s.prove(mode=FS | DRIVE, want_transcript=False)
        lat = {}
        for name, mode in (("device", FSD | DRIVE), ("resident", FS | DRIVE), ("host", FS | DRIVE | HOST)):
            best = 1e9
            for _ in range(3):
                r, tr = s.prove(mode=mode)
                best = min(best, 1e3 * (r.prove_s + r.poly_prove_s))
            lat[name] = (best, hashlib.sha256(tr).hexdigest(), 1e3 * r.prove_s)
        rounds, phases = s.fs_stats()
        print(f"vgg11 Fiat-Shamir prover latency: {lat['resident'][0]:.1f} ms with host-derived challenges over the resident kernels (default), "
              f"{lat['device'][0]:.1f} ms with device-side rounds (sumcheck {lat['device'][2]:.1f}), {lat['host'][0]:.1f} ms with a launch per round; "
              f"{rounds} rounds in {phases} device / resident phases so far")
        assert lat["device"][1] == lat["host"][1] == lat["resident"][1]
        assert lat["resident"][0] < lat["host"][0] * 1.05
        full, _ = s.prove(mode=FS)
        assert full.accepted == 1, full.message.decode()
        assert lat["device"][0] < lat["host"][0] * 1.05          # (a few percent of run-to-run noise either way)


@pytest.mark.parametrize("model,pic,pp", CASES[2:5])
def test_hybrid_host_tail_gives_the_same_transcripts(built, model, pic, pp):
    """ZKCNN_MODE_HOST_TAIL: tables of <= 64 entries finish their phase on the host; interactive and Fiat-Shamir transcripts unchanged"""
    HT = zkcnn_amd.MODE_HOST_TAIL
    with zkcnn_amd.Session(model, pic, pp) as s:
        a, ta = s.prove(seed=77)
        b, tb = s.prove(seed=77, mode=HT)
        assert a.accepted == 1 and b.accepted == 1 and ta == tb
        _, tc = s.prove(seed=78, mode=HT | REUSE | DRIVE)
        _, td = s.prove(seed=78, mode=REUSE | DRIVE)
        assert tc == td
        _, fa = s.prove(mode=FS)
        fb, fbt = s.prove(mode=FS | HT)
        assert fb.accepted == 1 and fa == fbt
        z, zt = s.prove(seed=5, mode=zkcnn_amd.MODE_ZK | HT)
        assert z.accepted == 1 and zt == s.prove(seed=5, mode=zkcnn_amd.MODE_ZK)[1]
