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
    K, steps = 2, 2
    sessions = [oracle_ffi.OracleSession("custom:F4", (4, 4, 1), 1, data_seed=9000 + rank * K + i) for i in range(K)]
    g = dp.AsyncGather(dist, "cpu", K << 14)
    done = [queue.Queue() for _ in range(K)]

    def stream(i):
        for k in range(steps):
            done[i].put(sessions[i].prove(seed=100 + k, mode=zkcnn_amd.MODE_REUSE_GENS | zkcnn_amd.MODE_DRIVE_ONLY)[1])
    th = [threading.Thread(target=stream, args=(i,)) for i in range(K)]
    dist.barrier()
    [t.start() for t in th]
    for k in range(steps):
        batch = [done[i].get() for i in range(K)]
        g.submit(rank, dp.pack([(rank * K + i, tr) for i, tr in enumerate(batch)]))
    [t.join() for t in th]
    res = g.wait()
    dist.barrier()
    if rank == 0:
        q.put([(step, r, sorted(img for img, _ in dp.unpack(blob))) for step, per_rank in enumerate(res) for r, blob in per_rank])
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
    got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(got) == [(step, r, [2 * r, 2 * r + 1]) for step in range(2) for r in range(world)]


def test_session_count_respects_host_memory():
    """bench.py lowers the streams per rank to what the host's memory allows (a vgg11 session keeps ~10 GB of circuit + witness on the
    host): the rule, on made-up numbers -- 8 ranks x 8 streams need ~640 GB"""
    def k_fit(avail_bytes, world, asked):
        return max(1, min(asked, int(avail_bytes / (10e9 * max(world, 1)))))
    assert k_fit(2e12, 8, 8) == 8 and k_fit(512e9, 8, 8) == 6 and k_fit(60e9, 8, 8) == 1 and k_fit(60e9, 1, 8) == 6
