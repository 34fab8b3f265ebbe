#!/usr/bin/env python3
"""bench.py -- prover throughput of the MI355X-native zkCNN GKR prover.

A step = ONE proof of the workload circuit per GPU (vgg11 / CIFAR-10 shape, pic_cnt = 1, synthetic picture
and weights), circuit + witness already resident in HBM when the timed region starts. Steps run in
"drive-only" mode: the reference verifier's challenge stream and call sequence drive the GPU prover, the
verifier's own CPU checks (which re-walk every gate) are skipped inside the timed region; the first warm-up
step runs the full verifier and must print acceptance.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

One JSON line on rank 0:
  value        whole-job proofs/s over all N GPUs (weak scaling: one image per GPU per step)
  roofline     dominant kernel: algorithmic bytes / HIP-event time measured inside the timed steps
  cpu_baseline the CPU oracle (port of the reference prover) on a bounded sample, 1 core
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a copy kernel reaches
WORKLOADS = {
    # name: (model string, picture, pictures per circuit)
    "vgg11": ("vgg11", (32, 32, 3), 1),
    "lenet": ("lenet", (32, 32, 1), 1),
    "vgg16": ("vgg16", (32, 32, 3), 1),
    "vgg11_pp8": ("vgg11", (32, 32, 3), 8),      # 8 pictures folded into ONE circuit (the reference's pic_cnt = 8; FFT convolutions)
    "vgg16_pp4": ("vgg16", (32, 32, 3), 4),
    # bounded CPU sample: vgg11 with every channel width divided by 4 (1/16 of the multiplication gates)
    "vgg11_quarter": ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),
}
# kernel classes whose algorithmic byte count is defined; the dominant one is reported
ROOFLINE_CLASSES = ["gate_reduce", "round_quad", "round_cubic", "msm_planes"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="vgg11", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default="vgg11_quarter", choices=sorted(WORKLOADS))
    ap.add_argument("--profile-all", action="store_true", help="print the per-kernel-class table of one extra proof to stderr")
    args = ap.parse_args()

    import torch
    import zkcnn_amd
    from zkcnn_amd import dp

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:        # launched by torch.distributed.run (also with one rank: same code path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    model, pic, pp = WORKLOADS[args.workload]
    t0 = time.time()
    sess = zkcnn_amd.Session(model, pic, pp, data_seed=20260928 + rank, device=local_rank)
    setup_s = time.time() - t0
    drive = zkcnn_amd.MODE_DRIVE_ONLY | zkcnn_amd.MODE_REUSE_GENS

    # ---- warm-up: first step with the full verifier (acceptance), the rest as the timed steps run ----
    accepted = None
    first = None
    for w in range(max(args.warmup, 1)):
        if w == 0:
            first, _ = sess.prove(seed=0x5EED0001, mode=zkcnn_amd.MODE_REUSE_GENS, want_transcript=False)
            accepted = first.accepted == 1
            if not accepted:
                raise SystemExit(f"verifier rejected the GPU proof: {first.message.decode()}")
        else:
            sess.prove(seed=0x5EED0001 + w, mode=drive, want_transcript=False)

    # one profiled proof to find the dominant kernel class (events on every launch; not timed)
    sess.profile("all")
    sess.prove(seed=0x5EED00FF, mode=drive, want_transcript=False)
    table = sess.profile_report(reset=True)
    dominant = max(ROOFLINE_CLASSES, key=lambda c: table[c]["ms"])
    if args.profile_all and rank == 0:
        for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            if v["launches"]:
                print(f"  {k:14s} {v['ms']:10.3f} ms {v['launches']:7d} launches {v['bytes'] / 1e9:10.3f} GB(alg)", file=sys.stderr)
    sess.profile([dominant])           # during the timed steps only the dominant class carries events

    # ---- timed region ----
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    prove_s = poly_s = 0.0
    gatherer = dp.AsyncGather(dist, "cuda", 1 << 20) if dist is not None else None
    for k in range(args.steps):
        res, tr = sess.prove(seed=0x5EED1000 + k, mode=drive, want_transcript=dist is not None)
        prove_s += res.prove_s
        poly_s += res.poly_prove_s
        if gatherer is not None:
            # the step's only exchange: this image's proof to rank 0 over RCCL, overlapped with the next proof (zkcnn_amd/dp.py)
            gatherer.submit(rank, tr)
    if gatherer is not None:
        gathered = gatherer.wait()
        if rank == 0:
            assert len(gathered) == args.steps and all(len(g) == world for g in gathered)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    prof = sess.profile_report(reset=True)[dominant]
    sess.profile(None)
    roofline = None
    if prof["launches"]:
        sec = prof["ms"] * 1e-3 / prof["launches"]
        achieved = prof["bytes"] / prof["launches"] / sec / 1e9
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "avg_launch_ms": round(sec * 1e3, 4), "launches_per_step": prof["launches"] / args.steps,
                    "algorithmic_bytes_per_launch": prof["bytes"] / prof["launches"],
                    "share_of_prover_time": round(prof["ms"] * 1e-3 / max(prove_s + poly_s, 1e-12), 3)}

    # the same kernel family on one large streaming launch (two 2^24-entry tables, device resident): this is the figure
    # to read against the HBM roof; the whole-proof average above is dominated by ~1.1 k tiny latency-bound rounds
    if roofline is not None and rank == 0:
        try:
            hc = zkcnn_amd.HipContext(local_rank)
            sec, nbytes = hc.bench_round_quadratic(24, 10)
            mul_sec = hc.bench_fr_mul(1 << 20, 256, 3)
            hc.close()
            roofline["streaming_launch"] = {"kernel": "k_round_quad, 2 x 2^24 entries", "ms": round(sec * 1e3, 4),
                                            "algorithmic_bytes": nbytes, "achieved": round(nbytes / sec / 1e9, 1),
                                            "frac": round(nbytes / sec / 1e9 / HBM_PEAK_GBS, 4),
                                            "traffic_pmc_bytes": 1.7037e9,      # profiles/r01_round_quad_kernel.md (FETCH_SIZE x2 + WRITE_SIZE)
                                            "fr_mul_ceiling_G_per_s": round((1 << 20) * 256 / mul_sec / 1e9, 1)}
        except Exception as e:      # the headline numbers do not depend on this extra measurement
            roofline["streaming_launch"] = {"error": str(e)}

    if rank != 0:
        sess.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- CPU baseline: the oracle (port of the reference prover), 1 core, bounded sample ----
    cpu = None
    if not args.no_cpu_baseline:
        from tests import oracle_ffi
        cm, cpic, cpp = WORKLOADS[args.cpu_sample]
        with oracle_ffi.OracleSession(cm, cpic, cpp, data_seed=20260928) as o:
            ores, _ = o.prove(seed=0x5EED0001, mode=drive, want_transcript=False)
        cpu_ms = 1e3 * (ores.prove_s + ores.poly_prove_s)
        # the GPU on the same sample, for a like-for-like ratio
        with zkcnn_amd.Session(cm, cpic, cpp, data_seed=20260928, device=local_rank) as gs:
            gs.prove(seed=0x5EED0001, mode=drive, want_transcript=False)
            gres, _ = gs.prove(seed=0x5EED0001, mode=drive, want_transcript=False)
        gpu_ms = 1e3 * (gres.prove_s + gres.poly_prove_s)
        cpu = {"value": round(cpu_ms, 1), "unit": "prover ms/image", "cores": 1, "kind": "port",
               "sample": f"{args.cpu_sample} ({cm}), pic_cnt=1, {ores.gate_cnt_bin} mul gates: CPU oracle prover time "
                         f"(sumcheck {1e3 * ores.prove_s:.0f} ms + Hyrax {1e3 * ores.poly_prove_s:.0f} ms); full vgg11 measured offline: see BASELINE.md",
               "gpu_same_sample_ms": round(gpu_ms, 2), "gpu_speedup_same_sample": round(cpu_ms / gpu_ms, 1),
               "host_cores_available": os.cpu_count()}

    steps = args.steps
    out = {
        "metric": "proofs/s (prover, vgg11 pic_cnt=1 per GPU); prover_ms_per_image alongside",
        "value": round(world * steps / elapsed, 4),
        "unit": "proofs/s",
        "n_gpus": world, "steps": steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u256 (BLS12-381 Fr, 8x32-bit Montgomery limbs)", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {model} {pic[0]}x{pic[1]}x{pic[2]} pic_cnt={pp}, one image per GPU",
                   "layers": first.n_layers, "input_size": first.input_size, "rounds": first.n_rounds,
                   "mul_gates": first.gate_cnt_bin, "add_gates": first.gate_cnt_uni, "parallelism": f"dp{world} (independent proofs, RCCL gather)"},
        "prover_ms_per_image": round(1e3 * (prove_s + poly_s) / steps, 3),
        "prover_ms_sumcheck": round(1e3 * prove_s / steps, 3), "prover_ms_commit": round(1e3 * poly_s / steps, 3),
        "verifier_pass": bool(accepted), "proof_kb": round(first.proof_kb + first.poly_proof_kb, 1),
        "setup_s": round(setup_s, 1), "witness_s": round(first.witness_s, 1), "upload_s": round(first.upload_s, 2),
        "roofline": roofline, "cpu_baseline": cpu,
    }
    print(json.dumps(out), flush=True)
    sess.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
