"""Data-parallel proving: pictures are independent proofs, so a batch shards across ranks with no
data-path collective; the only exchange is ONE gather of the finished per-image proofs (canonical
transcripts) to rank 0 -- RCCL over xGMI on GPUs ("nccl" backend), gloo in the CPU tests.

The reference has no multi-process path at all; its pic_cnt folds a batch into one
circuit, which is a different statement. Parity is therefore per image against the oracle.
"""
import struct


def shard(n_images, world, rank):
    """image ids of this rank: contiguous blocks, sizes differ by at most one"""
    base, extra = divmod(n_images, world)
    lo = rank * base + min(rank, extra)
    return list(range(lo, lo + base + (1 if rank < extra else 0)))


def pack(transcripts):
    """list[(image_id, bytes)] -> one byte string: count, then (id, length, payload) records"""
    out = [struct.pack("<I", len(transcripts))]
    for img, tr in transcripts:
        out.append(struct.pack("<IQ", img, len(tr)))
        out.append(tr)
    return b"".join(out)


def unpack(blob):
    (n,), off, res = struct.unpack_from("<I", blob, 0), 4, []
    for _ in range(n):
        img, ln = struct.unpack_from("<IQ", blob, off)
        off += 12
        res.append((img, bytes(blob[off:off + ln])))
        off += ln
    return res


def gather_proofs(local, dist=None, device="cpu"):
    """local: list[(image_id, transcript bytes)] of this rank. Returns the full list (sorted by image id) on rank 0,
    None elsewhere. One fixed-size gather: sizes are agreed with an all_reduce(MAX) first."""
    import torch
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(local)
    blob = pack(local)
    n = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    dist.all_reduce(n, op=dist.ReduceOp.MAX)
    cap = int(n.item())
    buf = torch.zeros(cap + 8, dtype=torch.uint8, device=device)
    buf[:8] = torch.frombuffer(bytearray(struct.pack("<Q", len(blob))), dtype=torch.uint8).to(device)
    buf[8:8 + len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    rank, world = dist.get_rank(), dist.get_world_size()
    if rank == 0:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.gather(buf, parts, dst=0)
        out = []
        for p in parts:
            raw = p.cpu().numpy().tobytes()
            (ln,) = struct.unpack_from("<Q", raw, 0)
            out += unpack(raw[8:8 + ln])
        return sorted(out)
    dist.gather(buf, None, dst=0)
    return None


class AsyncGather:
    """Per-step gather that overlaps with the next proof: the transcript goes to a device buffer and an asynchronous
    RCCL gather is queued on the collective stream; wait() drains everything before the final barrier."""

    def __init__(self, dist, device, cap_bytes):
        import torch
        self.dist, self.device, self.cap = dist, device, int(cap_bytes)
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.pending = []          # (work handle, send buffer, receive buffers)
        self.torch = torch

    def submit(self, image_id, transcript):
        torch = self.torch
        blob = struct.pack("<IQ", image_id, len(transcript)) + transcript
        if len(blob) > self.cap:
            raise ValueError("proof larger than the gather capacity")
        host = torch.frombuffer(bytearray(blob.ljust(self.cap, b"\0")), dtype=torch.uint8)
        send = host.to(self.device, non_blocking=False)
        recv = [torch.empty_like(send) for _ in range(self.world)] if self.rank == 0 else None
        work = self.dist.gather(send, recv, dst=0, async_op=True)
        self.pending.append((work, send, recv))

    def wait(self):
        """returns list[list[(image_id, bytes)]] per step on rank 0, None elsewhere"""
        out = []
        for work, _, recv in self.pending:
            work.wait()
            if recv is not None:
                step = []
                for t in recv:
                    raw = t.cpu().numpy().tobytes()
                    img, ln = struct.unpack_from("<IQ", raw, 0)
                    step.append((img, raw[12:12 + ln]))
                out.append(sorted(step))
        self.pending = []
        return out if self.rank == 0 else None


def prove_images(make_session, image_ids, challenge_seed, mode=0):
    """proves the given images one after the other on this rank; make_session(image_id) -> session object"""
    out, stats = [], []
    for img in image_ids:
        with make_session(img) as s:
            res, tr = s.prove(seed=challenge_seed + img, mode=mode)
        out.append((img, tr))
        stats.append(res)
    return out, stats
