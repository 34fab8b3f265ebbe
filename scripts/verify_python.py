"""The independent Python verifier (tests/test_verifier_python_cpu.py) on a model of one's choice: the CPU oracle proves, Python checks.
usage: python scripts/verify_python.py [model] [x y c] [pic_cnt] [mode: plain|zk|fresh|reference]   [challenge seed]   (reference = fresh generators + the inner-product argument down to length 1: the reference's semantics)      e.g.  python scripts/verify_python.py vgg11 32 32 3 1 plain"""
import hashlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zkcnn_amd  # noqa: E402
from tests import oracle_ffi  # noqa: E402
from tests.test_verifier_python_cpu import python_verify  # noqa: E402

model = sys.argv[1] if len(sys.argv) > 1 else "lenet"
pic = tuple(int(x) for x in sys.argv[2:5]) if len(sys.argv) > 4 else (32, 32, 1)
pp = int(sys.argv[5]) if len(sys.argv) > 5 else 1
kind = sys.argv[6] if len(sys.argv) > 6 else "plain"
seed = int(sys.argv[7], 0) if len(sys.argv) > 7 else 0x5EED0001
mode = {"plain": zkcnn_amd.MODE_REUSE_GENS, "zk": zkcnn_amd.MODE_REUSE_GENS | zkcnn_amd.MODE_ZK, "fresh": 0, "reference": zkcnn_amd.MODE_FULL_IPA}[kind]
orc = oracle_ffi.load()
t0 = time.time()
with oracle_ffi.OracleSession(model, pic, pp) as o:
    res, tr = o.prove(seed=seed, mode=mode)
    t1 = time.time()
    print(f"{model} {pic} pic_cnt={pp} [{kind}]: oracle accepted {res.accepted}, {res.n_layers} layers, {res.n_rounds} rounds, input 2^{res.input_bits}, "
          f"transcript {len(tr)} B sha256 {hashlib.sha256(tr).hexdigest()} ({t1 - t0:.0f} s)", flush=True)
    n = python_verify(orc, o, tr, seed, res.n_layers, zk=kind == "zk", fresh_gens=kind in ("fresh", "reference"), full_ipa=kind == "reference")
    print(f"the Python verifier ACCEPTS: {n} round messages checked, {time.time() - t1:.0f} s", flush=True)
