"""A lone proof runs the small and mid-size rounds of every phase in RESIDENT kernels that trade polynomials and challenges with the host through
mailboxes. Such a kernel can be lost -- it gives up after its time-out when nobody answers, or never gets the CUs its workgroups wait
for. Round 3 failed the proof then; since round 4 the phase is run AGAIN from its initialisation with a launch per round, the challenges answered so
far replayed (reference src/prover.cpp:360-426 computes the same field elements either way). The test hook ZKCNN_TEST_LIVE_FAIL=n sends the n-th
resident round of the process home; the transcript must still be the CPU oracle's, for failures in a first round, a middle round, a tail, in phase 1,
phase 2 and the layer-0 combine."""
import hashlib
import os
import subprocess
import sys

import pytest

from tests import oracle_ffi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODEL, PIC, PP = "vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1

CHILD = r"""
import hashlib, sys
sys.path.insert(0, %r)
import zkcnn_amd
with zkcnn_amd.Session(%r, %r, %d) as s:
    res, tr = s.prove(seed=0x5EED0F00, mode=zkcnn_amd.MODE_REUSE_GENS)
    print("RESULT", res.accepted, hashlib.sha256(tr).hexdigest(), len(tr), s.fs_stats()[0], flush=True)
"""


@pytest.fixture(scope="module")
def want(built):
    with oracle_ffi.OracleSession(MODEL, PIC, PP) as o:
        res, tr = o.prove(seed=0x5EED0F00, mode=2)
    assert res.accepted == 1
    return hashlib.sha256(tr).hexdigest(), len(tr), res.n_rounds


@pytest.mark.parametrize("fail_at", [0, 1, 7, 60, 333, 700, 900])
def test_lost_resident_kernel_phase_is_replayed_with_launches(want, fail_at):
    env = dict(os.environ, ZKCNN_TEST_HOOKS="1", ZKCNN_TEST_LIVE_FAIL=str(fail_at))
    r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, MODEL, PIC, PP)], capture_output=True, text=True, env=env, timeout=600)
    line = [x for x in r.stdout.splitlines() if x.startswith("RESULT")]
    assert line, r.stderr[-2000:]
    _, accepted, sha, n, resident = line[0].split()
    assert int(accepted) == 1 and (sha, int(n)) == want[:2], "the replayed phase changed the transcript"
    # most rounds of a lone proof are resident (all but the large ones and, since round 5, the last five of a phase, which the tail kernel hands to the host):
    # the hook fired unless the proof has fewer resident rounds than fail_at -- the child reports how many it started
    assert int(resident) > 0.5 * want[2]
    assert "run again with a launch per round" in r.stderr or fail_at >= int(resident) - 64


LOOP = r"""
import hashlib, sys, time
sys.path.insert(0, %r)
import zkcnn_amd
with zkcnn_amd.Session("lenet", (32, 32, 1), 1) as s:
    print("READY", flush=True)
    t0 = time.time()
    n = 0
    while time.time() - t0 < %f:
        res, tr = s.prove(seed=0x5EED0001)
        print("PROOF", res.accepted, hashlib.sha256(tr).hexdigest(), flush=True)
        n += 1
"""


def test_two_processes_on_one_gpu_each_believing_itself_alone(built):
    """The resident-kernel policy counts proofs per PROCESS: two processes on one GPU both run resident kernels (round 3: this combination
    deadlocked k_mid kernels of 1024 blocks once; the grid is bounded since). A second process that starts proving in the middle of the first
    one's proofs must not break either: every proof of both is accepted and is the committed golden transcript (a lost resident kernel is
    replayed with launches, not an error)."""
    import json
    import time
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "transcripts.json")))["lenet5_pic1"]["sha256"]
    env = dict(os.environ, ZKCNN_TEST_HOOKS="1")
    a = subprocess.Popen([sys.executable, "-c", LOOP % (ROOT, 12.0)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    assert a.stdout.readline().startswith("READY")
    time.sleep(1.0)                                   # the first process is proving by now
    b = subprocess.Popen([sys.executable, "-c", LOOP % (ROOT, 6.0)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)
    out_b, err_b = b.communicate(timeout=300)
    out_a, err_a = a.communicate(timeout=300)
    assert a.returncode == 0 and b.returncode == 0, (err_a[-1500:], err_b[-1500:])
    for name, out in (("first", out_a), ("second", out_b)):
        proofs = [x.split() for x in out.splitlines() if x.startswith("PROOF")]
        assert len(proofs) >= 3, f"{name} process: {len(proofs)} proofs"
        assert all(p[1] == "1" and p[2] == golden for p in proofs), f"{name} process: a proof was rejected or differs from the golden transcript"
