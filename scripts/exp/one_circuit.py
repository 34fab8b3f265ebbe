#!/usr/bin/env python3
"""One large circuit on one GPU (the stretch item of round 4: vgg16 with pic_cnt = 32 as the reference itself folds it -- reference
src/main_demo_vgg.cpp:36, src/models.cpp:43-146 -- layer 0 = 2^28 entries): build, one fully verified proof, property checks, drive-only timing.
   python scripts/exp/one_circuit.py [model] [pic_cnt]"""
import json
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("ZKCNN_TEST_HOOKS", "1")      # (the corrupted-message check below)
import torch  # noqa: E402
import zkcnn_amd as M  # noqa: E402
torch.cuda.init()
model = sys.argv[1] if len(sys.argv) > 1 else "vgg16"
pp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
out = {"model": model, "pic_cnt": pp}
t0 = time.time()
free0, _ = torch.cuda.mem_get_info(0)
s = M.Session(model, (32, 32, 3), pp)
out["setup_s"] = round(time.time() - t0, 1)
free1, _ = torch.cuda.mem_get_info(0)
out["hbm_gb"] = round((free0 - free1) / 1e9, 1)
print("[one] built", out, flush=True)
t = time.time()
res, tr = s.prove(seed=0x5EED0001, mode=M.MODE_REUSE_GENS)
out.update(accepted=res.accepted, n_layers=res.n_layers, input_bits=res.input_bits, rounds=res.n_rounds, proof_kb=round(res.proof_kb + res.poly_proof_kb, 1),
           first_proof_wall_s=round(time.time() - t, 1), verify_s=round(res.verify_s + res.poly_verify_s, 2), message=res.message.decode())
print("[one] verified", out, flush=True)
r2, tr2 = s.prove(seed=0x5EED0001, mode=M.MODE_REUSE_GENS | M.MODE_DRIVE_ONLY)
out["drive_only_equal"] = bool(tr2 == tr)
ms = []
for k in range(3):
    r, _ = s.prove(seed=0x5EED0010 + k, mode=M.MODE_REUSE_GENS | M.MODE_DRIVE_ONLY, want_transcript=False)
    ms.append(1e3 * (r.prove_s + r.poly_prove_s))
out["prover_ms"] = round(min(ms), 1)
out["pictures_per_s_single_stream"] = round(pp / (min(ms) * 1e-3), 1)
bad, _ = s.prove(seed=0x5EED0001, mode=M.MODE_REUSE_GENS | M.MODE_TAMPER | ((res.n_messages // 2) << 8))
out["tampered_rejected"] = bad.accepted == 0
out["replay_verifies"] = s.verify(tr, seed=0x5EED0001, mode=M.MODE_REUSE_GENS).accepted == 1
free2, _ = torch.cuda.mem_get_info(0)
out["hbm_gb_after_proofs"] = round((free0 - free2) / 1e9, 1)
s.close()
print(json.dumps(out), flush=True)
