"""On-disk / wire form of a proof. The proof proper is the canonical transcript the verifier records
(Fr: 32 bytes little-endian canonical, G1: 48 bytes compressed, in protocol order); this module wraps it with what a verifier
needs to replay it: the statement (model descriptor) and how the challenges were made.

  magic "ZKCNNPF1" | u32 header length | header (UTF-8 JSON) | u64 transcript length | transcript | sha256(all of the above)
"""
import hashlib
import json
import struct

MAGIC = b"ZKCNNPF1"


def dumps(transcript, model, pic, pic_cnt, data_seed, challenge_seed, mode, statement=None):
    from . import MODE_FIAT_SHAMIR, MODE_REUSE_GENS, MODE_ZK, MODE_FULL_IPA
    header = {"model": model, "pic": list(pic), "pic_cnt": pic_cnt, "data_seed": data_seed, "statement": statement,
              "challenges": "fiat-shamir/sha256" if mode & MODE_FIAT_SHAMIR else "seeded-stream",
              "challenge_seed": None if mode & MODE_FIAT_SHAMIR else challenge_seed,
              "session_generators": bool(mode & MODE_REUSE_GENS),
              # the bits that shape the protocol (what messages the transcript holds); TAMPER / DRIVE_ONLY / SEEDED are properties of a run, not of a proof
              "mode": mode & (MODE_FIAT_SHAMIR | MODE_REUSE_GENS | MODE_ZK | MODE_FULL_IPA)}
    h = json.dumps(header, sort_keys=True).encode()
    body = MAGIC + struct.pack("<I", len(h)) + h + struct.pack("<Q", len(transcript)) + bytes(transcript)
    return body + hashlib.sha256(body).digest()


def loads(blob):
    """-> (header dict, transcript bytes); raises ValueError on a damaged file"""
    if len(blob) < len(MAGIC) + 4 + 8 + 32 or blob[:8] != MAGIC:
        raise ValueError("not a zkcnn proof file")
    if hashlib.sha256(blob[:-32]).digest() != blob[-32:]:
        raise ValueError("proof file checksum mismatch")
    (hl,) = struct.unpack_from("<I", blob, 8)
    header = json.loads(blob[12:12 + hl].decode())
    (tl,) = struct.unpack_from("<Q", blob, 12 + hl)
    tr = blob[20 + hl:20 + hl + tl]
    if len(tr) != tl or 20 + hl + tl + 32 != len(blob):
        raise ValueError("proof file length mismatch")
    return header, tr


def save(path, *args, **kw):
    with open(path, "wb") as f:
        f.write(dumps(*args, **kw))


def load(path):
    with open(path, "rb") as f:
        return loads(f.read())


def dumps_from(session, transcript, challenge_seed, mode):
    """proof file of a proof made by `session`, carrying the statement a stand-alone verifier needs"""
    return dumps(transcript, session.model, session.pic, session.pic_cnt, session.data_seed, challenge_seed, mode, session.statement())


class NotAProofError(ValueError):
    """the file holds an interactive transcript: its challenges were the verifier's private coins, off line it proves nothing"""


def _replay_args(header, allow_seeded_replay):
    """(seed, mode) for Session.verify. A header is prover-controlled input: a file that says "seeded-stream" names the very seed its
    challenges came from, so whoever wrote it knew every challenge before sending a single message. Such a file is only ever
    replayed on request (parity / debugging), never accepted as a proof."""
    from . import MODE_FIAT_SHAMIR, MODE_REUSE_GENS, MODE_ZK, MODE_FULL_IPA
    mode = int(header["mode"]) & (MODE_FIAT_SHAMIR | MODE_REUSE_GENS | MODE_ZK | MODE_FULL_IPA)
    if mode & MODE_FIAT_SHAMIR:
        return None, mode
    if not allow_seeded_replay:
        raise NotAProofError("interactive transcript (seeded challenge stream): not a proof; pass allow_seeded_replay=True to replay it for debugging")
    return int(header["challenge_seed"] or 0), mode


def verify_standalone(blob, session_cls, allow_seeded_replay=False, **kw):
    """verifies a proof file with nothing but the file: the circuit is rebuilt from the header's model descriptor + statement
    (session_cls: zkcnn_amd.Session for GPU predicates, or the oracle's session class in tests). Returns the Result.
    Only Fiat-Shamir proofs are accepted; see _replay_args."""
    header, tr = loads(blob)
    if header.get("statement") is None:
        raise ValueError("proof file carries no statement")
    seed, mode = _replay_args(header, allow_seeded_replay)
    with session_cls(header["model"], tuple(header["pic"]), header["pic_cnt"], statement=header["statement"], **kw) as v:
        return v.verify(tr, seed=seed, mode=mode)


def verify_with(session, blob, allow_seeded_replay=False):
    """checks a proof file against `session` (whose circuit must be the file's statement); returns the Result"""
    header, tr = loads(blob)
    seed, mode = _replay_args(header, allow_seeded_replay)
    return session.verify(tr, seed=seed, mode=mode)
