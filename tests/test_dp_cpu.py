"""The N > 1 path on CPU: world_size 2 over gloo. Each rank proves its shard of a batch with the CPU oracle
standing in for the GPU prover (the sharding / packing / gather code is the product's zkcnn_amd.dp)."""
import hashlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import zkcnn_amd
from zkcnn_amd import dp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

MODEL, PIC = "custom:C2:3:1:s M F4", (4, 4, 1)


def test_shard_covers_every_image_once():
    for n in (1, 2, 7, 8, 32):
        for world in (1, 2, 3, 8):
            ids = sum((dp.shard(n, world, r) for r in range(world)), [])
            assert ids == list(range(n))
            sizes = [len(dp.shard(n, world, r)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1


def test_pack_roundtrip():
    items = [(3, b"abc"), (0, b""), (9, bytes(range(200)))]
    assert dp.unpack(dp.pack(items)) == items


def _session(img):
    from tests import oracle_ffi
    return oracle_ffi.OracleSession(MODEL, PIC, 1, data_seed=1000 + img)


def _worker(rank, world, port, n_images, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    local, stats = dp.prove_images(_session, dp.shard(n_images, world, rank), challenge_seed=77)
    assert all(s.accepted == 1 for s in stats)
    allp = dp.gather_proofs(local, dist, "cpu")
    if rank == 0:
        q.put([(i, hashlib.sha256(t).hexdigest()) for i, t in allp])
    else:
        assert allp is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gather_equals_single_process(oracle):
    n_images = 5
    single, _ = dp.prove_images(_session, list(range(n_images)), challenge_seed=77)
    want = [(i, hashlib.sha256(t).hexdigest()) for i, t in single]
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_images, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == want
    assert len({h for _, h in got}) == n_images          # different pictures give different proofs


def _async_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = dp.AsyncGather(dist, "cpu", 4096)
    for step in range(3):
        g.submit(rank, bytes([rank, step]) * (10 + step))
    res = g.wait()
    if rank == 0:
        q.put(res)
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


def test_async_gather_keeps_steps_and_ranks_apart():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_async_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == [[(0, bytes([0, s]) * (10 + s)), (1, bytes([1, s]) * (10 + s))] for s in range(3)]


def _streams_worker(rank, world, port, q):
    """what bench.py does per rank: K sessions proving on K host threads, one packed asynchronous gather per step"""
    import threading
    from tests import oracle_ffi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    K, steps = 2, 2
    sessions = [oracle_ffi.OracleSession(MODEL, PIC, 1, data_seed=2000 + rank * K + i) for i in range(K)]
    g = dp.AsyncGather(dist, "cpu", K << 16)
    out = [[None] * K for _ in range(steps)]

    def stream(i):
        for k in range(steps):
            out[k][i] = sessions[i].prove(seed=500 + k)[1]
    th = [threading.Thread(target=stream, args=(i,)) for i in range(K)]
    [t.start() for t in th]
    [t.join() for t in th]
    for k in range(steps):
        g.submit(rank, dp.pack([(rank * K + i, out[k][i]) for i in range(K)]))
    res = g.wait()
    if rank == 0:
        # every gathered proof is checked without its prover: replayed against a verifier-only session built from the statement
        ok = []
        stmts = {}
        for step, per_rank in enumerate(res):
            for r, blob in per_rank:
                for img, tr in dp.unpack(blob):
                    if img not in stmts:
                        with oracle_ffi.OracleSession(MODEL, PIC, 1, data_seed=2000 + img) as tmp:
                            stmts[img] = tmp.statement()
                    with oracle_ffi.OracleSession(MODEL, PIC, 1, statement=stmts[img]) as v:
                        ok.append((step, img, v.verify(tr, seed=500 + step).accepted))
        q.put(ok)
    for s in sessions:
        s.close()
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_two_streams_each_all_proofs_verify(oracle):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_streams_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(got) == [(step, img, 1) for step in range(2) for img in range(4)]


def _bench_shape_worker(rank, world, port, q):
    """bench.py's step loop per rank, shrunk: K sessions on K host threads, a step = K proofs, one packed asynchronous gather per step to rank 0"""
    import queue
    import threading
    from tests import oracle_ffi
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import time
    import psutil
    K, steps = 2, 10
    sessions = [oracle_ffi.OracleSession("custom:C4:3:1:n M F8", (8, 8, 2), 1, data_seed=9000 + rank * K + i) for i in range(K)]
    g = dp.AsyncGather(dist, "cpu", K << 16)
    done = [queue.Queue() for _ in range(K)]

    def stream(i):
        for k in range(steps):
            done[i].put(sessions[i].prove(seed=100 + k, mode=zkcnn_amd.MODE_REUSE_GENS | zkcnn_amd.MODE_DRIVE_ONLY)[1])
    th = [threading.Thread(target=stream, args=(i,)) for i in range(K)]
    dist.barrier()
    t0 = time.time()
    [t.start() for t in th]
    for k in range(steps):
        batch = [done[i].get() for i in range(K)]
        g.submit(rank, dp.pack([(rank * K + i, tr) for i, tr in enumerate(batch)]))
    [t.join() for t in th]
    busy = time.time() - t0
    res = g.wait()
    # what bench.py reports per rank: its own proving time and its resident memory, all-gathered
    import torch
    mine = torch.tensor([busy, float(psutil.Process().memory_info().rss)], dtype=torch.float64)
    allr = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(allr, mine)
    dist.barrier()
    if rank == 0:
        q.put(([(step, r, sorted(img for img, _ in dp.unpack(blob))) for step, per_rank in enumerate(res) for r, blob in per_rank],
               [float(a[0]) for a in allr], [float(a[1]) for a in allr]))
    for s in sessions:
        s.close()
    dist.destroy_process_group()


def test_eight_ranks_step_loop_shape(oracle):
    """the N = 8 launch of bench.py (8 ranks x K streams, a gather per step) on gloo: every rank's K proofs of every step arrive at rank 0"""
    world = 8
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bench_shape_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got, busy, rss = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(got) == [(step, r, [2 * r, 2 * r + 1]) for step in range(10) for r in range(world)]
    # no rank straggles: the ranks prove side by side (ranks that serialised behind one another would finish up to 8x apart); the slack
    # covers 16 proving threads on however few cores this box has
    # (round-4 review: spread of the ranks' proving times <= 1.1x; the constant covers thread start-up on a box with fewer cores than proving threads)
    assert max(busy) <= 1.1 * min(busy) + 0.3, busy
    # host-memory budget: the ranks together stay inside what bench.py's rule reserves for them (streams_that_fit: 6 GB per session)
    import bench
    assert sum(rss) <= world * 2 * bench.HOST_BYTES_PER_SESSION
    import psutil
    assert bench.streams_that_fit(psutil.virtual_memory().total, world, 2) >= 1


def test_session_count_respects_host_memory():
    """bench.py lowers the streams per rank to what the host's memory allows (a vgg11 session keeps ~10 GB of circuit + witness on the
    host): the rule, on made-up numbers -- 8 ranks x 8 streams need ~640 GB"""
    import bench
    k_fit = bench.streams_that_fit
    assert k_fit(2e12, 8, 8) == 8 and k_fit(256e9, 8, 8) == 5 and k_fit(40e9, 8, 8) == 1 and k_fit(40e9, 1, 8) == 6
    # lock-step batches: one full session per rank, the other lanes are clones (no host circuit): 8 ranks x 56 fit 256 GB, not 56 GB
    assert k_fit(256e9, 8, 56, clones=True) == 56 and k_fit(56e9, 8, 56, clones=True) == 5 and k_fit(40e9, 8, 56, clones=True) == 1


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus N` with no RANK in the environment becomes the torch.distributed.run launcher (one rank per GPU, loopback
    rendezvous); on a box without N GPUs it exits with a message instead of an assertion (round-2 review, next #1)"""
    import subprocess
    import sys as _sys
    import bench
    cmd = bench.launch_command(["--gpus", "8", "--steps", "3"], 8, port=29555, python="python3")
    assert cmd[:3] == ["python3", "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29555"
    k = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[k + 1:] == ["--gpus", "8", "--steps", "3"]
    assert bench.launch_command([], 2)[cmd.index("--master-port") + 1].isdigit()          # a free port is picked when none is given
    if bench.visible_gpus() < 2:
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
        r = subprocess.run([_sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "GPU(s)" in r.stderr and "AssertionError" not in r.stderr and "Traceback" not in r.stderr
        env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")      # a launcher that started the wrong number of ranks: a message, not an assert
        r = subprocess.run([_sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, env=env, timeout=300)
        assert r.returncode != 0 and "Traceback" not in r.stderr


def test_bench_parent_prints_the_last_line_its_child_left(monkeypatch, capsys):
    """`python bench.py` at N = 1 measures in a child process that leaves its line in a file after every stage; the parent prints the last one.
    A child that dies in the LAST, optional stage (the PMC counter passes: it marks the line before it starts them) costs roofline.traffic, not the
    run; a child that dies anywhere else -- a byte-parity mismatch, a crash in a companion -- is a FAILED run: the line is printed with the child's
    exit code in it and the parent exits non-zero (round-3 advisor finding: it used to exit 0)."""
    import json
    import bench

    def child(line, rc):
        def run(cmd, env):
            assert "--inner" in cmd and cmd[-2:] == ["--steps", "3"]
            with open(env["ZKCNN_BENCH_RESULT"], "w") as f:
                f.write(line)
            return rc
        return run

    monkeypatch.setattr(bench, "_run_child", child('{"metric": "m", "value": 1.5, "incomplete_after": "CPU baseline; PMC", "only_optional_stages_left": true}', -6))
    assert bench.supervise(["--steps", "3"]) == 0
    out = capsys.readouterr()
    d = json.loads(out.out.strip())
    assert d["value"] == 1.5 and d["child_rc"] == -6 and "-6" in out.err
    monkeypatch.setattr(bench, "_run_child", child('{"metric": "m", "value": 1.5, "incomplete_after": "FAILED: the GPU transcript differs", "transcript_equal_to_cpu_oracle": false}', 1))
    assert bench.supervise(["--steps", "3"]) == 1
    d = json.loads(capsys.readouterr().out.strip())
    assert d["child_rc"] == 1 and d["transcript_equal_to_cpu_oracle"] is False
    monkeypatch.setattr(bench, "_run_child", child('{"metric": "m", "value": 1.5, "incomplete_after": "companions"}', -11))
    assert bench.supervise(["--steps", "3"]) != 0                          # a crash in a companion stage: the line is there, the run failed
    capsys.readouterr()
    monkeypatch.setattr(bench, "_run_child", child('{"metric": "m", "value": 2.5}', 0))
    assert bench.supervise(["--steps", "3"]) == 0 and json.loads(capsys.readouterr().out.strip()) == {"metric": "m", "value": 2.5}
    monkeypatch.setattr(bench, "_run_child", lambda cmd, env: 3)            # a child that left nothing (no GPU, bad arguments)
    with pytest.raises(SystemExit) as e:
        bench.supervise([])
    assert e.value.code == 3
