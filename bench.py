#!/usr/bin/env python3
"""bench.py -- prover throughput of the MI355X-native zkCNN GKR prover.

A step = ONE proof of the workload circuit (vgg11 / CIFAR-10 shape, pic_cnt = 1, synthetic picture and weights) for each
of the --streams images kept in flight on every GPU (own session, host thread and HIP stream per image: the interactive
rounds of one proof are latency bound and leave the GPU idle, so independent proofs overlap), circuits + witnesses already
resident in HBM when the timed region starts. The single-stream latency (prover ms/image) is measured next to it. Steps run in
"drive-only" mode: the reference verifier's challenge stream and call sequence drive the GPU prover, the
verifier's own CPU checks (which re-walk every gate) are skipped inside the timed region; the first warm-up
step runs the full verifier and must print acceptance.

  python bench.py --gpus N --steps K --warmup W      (N > 1: launched by torch.distributed.run, one rank per GPU)

One JSON line on rank 0:
  value        whole-job proofs/s over all N GPUs (weak scaling: --streams images per GPU per step)
  roofline     dominant kernel: algorithmic bytes / HIP-event time measured inside the timed steps
  cpu_baseline the CPU oracle (port of the reference prover) on the SAME workload, one core, plus the aggregate of several
               independent single-threaded provers running side by side on the host cores (the reference has no threads)

Modes of the timed proofs (stated in `metric`). Default, --semantics reference (since round 6): the REFERENCE's protocol -- the verifier draws
new random generators for every proof (reference src/verifier.cpp:119-128: nothing built for one proof survives it, every table is built inside
the prover's clock) and the inner-product argument runs down to length 1 -- SEEDED (reproducible challenges and generator scalars, a seed of its
own for every proof) | DRIVE_ONLY | FULL_IPA. `value`, `prover_ms_per_image`, the roofline and the CPU baseline are all in that mode. The
public-generator variant of rounds 1-5 (--semantics public: hash-to-curve generators with a resident byte table, argument cut at 256) is the
`public_generators` object of the line. upload_sort_s = gate sort + upload, which the reference pays inside its prover timer.
"""
import argparse
import json
import os
import queue
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# HIP's default of 4 hardware queues is what the 8 streams want (measured 28 / 44 / 51 / 59 / 45 / 45 proofs/s with 1 / 2 / 3 / 4 / 5 / 6-8
# queues): pin it against a different default in the environment
os.environ.setdefault("GPU_MAX_HW_QUEUES", "4")

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is what a copy kernel reaches
WORKLOADS = {
    # name: (model string, picture, pictures per circuit)
    "vgg11": ("vgg11", (32, 32, 3), 1),
    "lenet": ("lenet", (32, 32, 1), 1),
    "vgg16": ("vgg16", (32, 32, 3), 1),
    "vgg11_pp8": ("vgg11", (32, 32, 3), 8),      # 8 pictures folded into ONE circuit (the reference's pic_cnt = 8; FFT convolutions)
    "vgg16_pp4": ("vgg16", (32, 32, 3), 4),
    "vgg16_pp32": ("vgg16", (32, 32, 3), 32),    # BASELINE configs[4] as the reference itself would run it: ONE circuit over 32 pictures (layer 0 = 2^28 entries)
    # bounded CPU sample: vgg11 with every channel width divided by 4 (1/16 of the multiplication gates)
    "vgg11_quarter": ("vgg:16 M 32 M 64 64 M 128 128 M 128 128 M", (32, 32, 3), 1),
    # vgg11 with every channel width halved (1/4 of the multiplication gates): ~7 s of CPU prover time on one core
    "vgg11_half": ("vgg:32 M 64 M 128 128 M 256 256 M 256 256 M", (32, 32, 3), 1),
}
# kernel classes whose algorithmic byte count is defined; the dominant one is reported
ROOFLINE_CLASSES = ["gate_reduce", "round_quad", "round_fine", "round_cubic", "msm_planes"]
HBM_CLASSES = ["gate_reduce", "round_quad", "round_fine", "round_cubic"]
DATA_SEED = 20260928         # synthetic picture + weights
PARITY_SEED = 0x5EED0001     # challenge stream of the proof whose transcript is compared with the CPU oracle's, byte for byte


def launch_command(argv, n_gpus, port=None, python=None):
    """the command `bench.py --gpus N` re-executes itself as when nobody launched the ranks: one process per GPU under torch.distributed.run"""
    import socket
    if port is None:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
    return [python or sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


# kernels behind each byte-counted class of the built-in profiler (names as rocprofv3 reports them)
# (kernels are functors launched through k_run<F> / k_run_b<F>: the functor's name is part of the kernel's)
CLASS_KERNELS = {"round_quad": ("k_round_quad2_f",), "round_fine": ("k_round_fine_f",), "round_tail": ("k_tail",), "round_cubic": ("k_round_cubic",),
                 "gate_reduce": ("k_gate_multi", "k_conv_wa", "k_conv_m1", "k_conv_e", "k_conv_ae", "k_conv_m2"),
                 "msm_planes": ("k_msm_codes", "k_planes_acc", "k_bytes_acc", "k_msm_planes", "k_scalar_codes", "k_scalar_mags", "k_bit_masks", "k_compact_flags", "k_cl_blind_rows")}


CALIBRATION_KERNEL = "k_round_quad2<0>(round2_args)"      # the plain kernel zk_bench_round_quadratic launches (RQ_FOLD): 2 x 2^24 entries, known byte count
CALIBRATION_LOG_N = 24


def measure_pmc_traffic(workload, kernels, timeout_s=200):
    """HBM bytes per launch of `kernels`, measured NOW: two short child runs of this script under `rocprofv3 --pmc` (FETCH_SIZE and WRITE_SIZE in
    separate passes, as MI355X_MICROARCH.md prescribes: they do not fit one pass; KB counters). The guide's gfx950 correction (FETCH_SIZE x 2) is
    calibrated for 16-byte-per-lane streaming reads only and tells callers to calibrate their own access pattern: the child therefore ALSO runs the
    round kernel on two device-resident 2^24-entry tables (1 GiB read, 0.5 GiB written per launch: known, and far beyond the 256 MiB Infinity Cache),
    and the raw counters of the proof's launches are scaled by (known bytes / raw counter) of those calibration launches -- no hard-coded factor.
    One session, four proofs with every round a launch. None if rocprofv3 is missing or anything goes wrong."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    known = {"FETCH_SIZE": 64.0 * (1 << CALIBRATION_LOG_N), "WRITE_SIZE": 32.0 * (1 << CALIBRATION_LOG_N)}     # bytes per calibration launch
    total, launches, factor = {}, 0, {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(prefix="zkcnn_pmc_") as tmp:
                cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", tmp, "-o", "pmc", "--",
                       sys.executable, os.path.abspath(__file__), "--pmc-child", "--workload", workload]
                env = dict(os.environ, TMPDIR=tmp)
                env.pop("RANK", None)
                # (its own process group: a pass that does not come back is killed WITH the python process rocprofv3 started)
                proc = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
                try:
                    proc.wait(timeout=timeout_s)
                except subprocess.TimeoutExpired:
                    import signal
                    os.killpg(proc.pid, signal.SIGKILL)
                    proc.wait()
                    return None
                val, n, cal, n_cal = 0.0, 0, 0.0, 0
                for f in glob.glob(os.path.join(tmp, "**", "*counter_collection.csv"), recursive=True):
                    for r in csv.DictReader(open(f)):
                        if r.get("Counter_Name") != counter:
                            continue
                        name = r["Kernel_Name"].replace("void ", "")
                        if name.startswith(CALIBRATION_KERNEL.split("<")[0] + "<"):
                            cal += float(r["Counter_Value"])
                            n_cal += 1
                        elif any(k in name for k in kernels):
                            val += float(r["Counter_Value"])
                            n += 1
                if not n or not n_cal or cal <= 0:
                    return None
                factor[counter] = known[counter] / (cal / n_cal)          # bytes per counter unit, for this kernel's access pattern
                total[counter], launches = val * factor[counter], n
        return {"bytes_per_launch": (total["FETCH_SIZE"] + total["WRITE_SIZE"]) / launches, "fetch_bytes": total["FETCH_SIZE"], "write_bytes": total["WRITE_SIZE"],
                "launches": launches, "kernels": list(kernels),
                "bytes_per_counter_unit": {k: round(v, 1) for k, v in factor.items()}}
    except Exception:       # noqa: BLE001 - an optional measurement
        return None


def pmc_child(workload):
    """what measure_pmc_traffic profiles: the calibration launches (known bytes), then one session, four proofs with every round a launch"""
    import zkcnn_amd
    model, pic, pp = WORKLOADS[workload]
    hc = zkcnn_amd.HipContext(0)
    hc.bench_round_quadratic(CALIBRATION_LOG_N, 4)
    hc.close()
    with zkcnn_amd.Session(model, pic, pp, data_seed=DATA_SEED) as s:
        for k in range(4):
            s.prove(seed=0x5EED3000 + k, mode=zkcnn_amd.MODE_DRIVE_ONLY | zkcnn_amd.MODE_REUSE_GENS | zkcnn_amd.MODE_HOST_ROUNDS, want_transcript=False)


HOST_BYTES_PER_SESSION = 6e9      # a vgg11 session holds 2.7 GB on the host, 4.3 GB at its peak while it is built (host_peak_rss_gb_while_building)


HOST_BYTES_PER_CLONE = 0.25e9     # a clone holds no circuit on the host: buffers of its picture, its transcript, its verifier's state (measured 0.03-0.1 GB)


def streams_that_fit(available_bytes, world, asked, clones=False):
    """sessions per rank the host's memory allows: every rank of the node builds its sessions side by side on the same host.
    clones: one full session per rank, the others are clones of it (lock-step batches)"""
    if clones:
        per_rank = available_bytes / max(world, 1) - HOST_BYTES_PER_SESSION
        return max(1, min(int(asked), 1 + int(max(per_rank, 0) / HOST_BYTES_PER_CLONE)))
    return max(1, min(int(asked), int(available_bytes / (HOST_BYTES_PER_SESSION * max(world, 1)))))


def visible_gpus():
    """number of GPUs this process could use, without creating a HIP context in it (the ranks are separate processes)"""
    try:
        import torch
        return int(torch.cuda.device_count()) if torch.cuda.is_available() else 0
    except Exception:       # noqa: BLE001
        return 0


def self_launch_if_needed(args, argv):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: become the launcher (exec, same stdout: rank 0 prints the line)"""
    if args.gpus <= 1 or "RANK" in os.environ:
        return
    have = visible_gpus()
    if have < args.gpus and not (args.rehearse_shared_gpu and have >= 1):
        raise SystemExit(f"bench.py --gpus {args.gpus}: this box has {have} GPU(s); one rank per GPU is needed (no CPU fallback, no oversubscription)")
    cmd = launch_command(argv, args.gpus)
    print("[bench] launching the ranks: " + " ".join(cmd), file=sys.stderr, flush=True)
    os.execv(cmd[0], cmd)


def _run_child(cmd, env):
    """the measuring process; a SIGTERM / SIGINT sent to this process (a driver's time limit) ends it too, so that it does not keep the GPU"""
    import signal
    import subprocess
    proc = subprocess.Popen(cmd, env=env)
    old = {sig: signal.signal(sig, lambda signum, frame: proc.kill()) for sig in (signal.SIGTERM, signal.SIGINT)}
    try:
        return proc.wait()
    finally:
        for sig, h in old.items():
            signal.signal(sig, h)


def supervise(argv):
    """N = 1 without a launcher: the measurement runs in a child process that leaves its line in a file -- a provisional one as soon as the timed
    region and its replay verification are behind it, updated after every later stage (companions, roofline, PMC passes, CPU baseline), the final
    one at the end. This process touches no GPU and prints the last line the child left: a failure in a LATE, optional stage (seen once: a GPU
    memory fault 2.5 minutes into a run, after the timed region) costs the stages behind it -- named in "incomplete_after" -- not the line."""
    import tempfile
    with tempfile.TemporaryDirectory(prefix="zkcnn_bench_") as tmp:
        res = os.path.join(tmp, "line.json")
        rc = _run_child([sys.executable, os.path.abspath(__file__), "--inner"] + list(argv), dict(os.environ, ZKCNN_BENCH_RESULT=res))
        line = open(res).read().strip() if os.path.exists(res) else ""
    if line:
        if rc == 0:
            print(line, flush=True)
            return 0
        # the child died: only a failure in the LAST, optional stage (the PMC counter passes, marked by the child before it starts them) is
        # tolerated -- the line is complete up to roofline.traffic. Anything earlier (a parity mismatch, a crash in a companion) is a failed run:
        # the line is still printed, with the child's exit code in it, and this process exits non-zero.
        try:
            d = json.loads(line)
        except ValueError:
            d = {}
        tolerated = bool(d.get("only_optional_stages_left"))
        d["child_rc"] = rc
        print(f"[bench] the measuring process ended with code {rc}; printing the last line it left ({'optional stage failed' if tolerated else 'FAILED RUN'})", file=sys.stderr, flush=True)
        print(json.dumps(d), flush=True)
        return 0 if tolerated else (rc if 0 < rc < 256 else 1)
    raise SystemExit(rc if rc else 1)


def _cpu_prover_worker(args):
    """one single-threaded CPU prover (the oracle) on the workload: returns (prover seconds, wall seconds incl. circuit + witness, ...,
    sha256 of the canonical transcript -- the GPU proof of the same picture, challenge seed and modes must hash to the same value)"""
    model, pic, pp, seed, mode = args
    sys.path.insert(0, ROOT)
    import hashlib
    from tests import oracle_ffi
    t0 = time.time()
    with oracle_ffi.OracleSession(model, pic, pp, data_seed=seed) as o:
        res, tr = o.prove(seed=PARITY_SEED, mode=mode, want_transcript=True)
    return res.prove_s + res.poly_prove_s, time.time() - t0, int(res.gate_cnt_bin), res.prove_s, res.poly_prove_s, hashlib.sha256(tr).hexdigest(), len(tr)


def pin_to_gpu_numa_node(torch, local_rank):
    """best effort: the host threads of a rank spin on a slot their GPU writes over PCIe; keep them on the GPU's NUMA node"""
    if os.environ.get("ZKCNN_BENCH_NOPIN"):
        return -1
    try:
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        base = f"/sys/bus/pci/devices/{bdf.lower()}"
        node = int(open(base + "/numa_node").read())
        if node < 0:
            return -1
        cpus = set()
        for part in open(f"/sys/devices/system/node/node{node}/cpulist").read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if len(cpus) >= 16:
            os.sched_setaffinity(0, cpus)
        return node
    except Exception:       # noqa: BLE001 - placement is an optimisation only
        return -1


def physical_cores(cpus):
    """one logical CPU per physical core among `cpus` (the first SMT sibling of each), sorted"""
    seen, out = set(), []
    for c in sorted(cpus):
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{c}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(c)
        if sib not in seen:
            seen.add(sib)
            out.append(c)
    return out


def pin_worker(j, n_workers):
    """ZKCNN_BENCH_PIN_WORKERS=1 (experiment): worker j of this rank on a physical core of its own inside the rank's CPU set -- the workers spin on their
    GPU's mailboxes and never sleep, so wherever the scheduler put them first is where they stay"""
    if not os.environ.get("ZKCNN_BENCH_PIN_WORKERS"):
        return
    try:
        cores = physical_cores(os.sched_getaffinity(0))
        if len(cores) > n_workers + 1:
            os.sched_setaffinity(0, {cores[1 + j]})          # (core 0 of the set is left to the main thread)
    except Exception:       # noqa: BLE001
        pass


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="vgg11", choices=sorted(WORKLOADS))
    ap.add_argument("--streams", type=int, default=None, help="proofs in flight per GPU (default: 8 x --lanes, i.e. 8 lock-step batches, two per hardware queue; with --lanes 1: 8)")
    ap.add_argument("--lanes", type=int, default=8, help="lanes of a lock-step batch: that many proofs share ONE host thread, ONE HIP stream and ONE kernel launch per round "
                                                          "(zkcnn_batch_*); --streams / --lanes batches per GPU. 1 = every proof its own thread and stream (the round-3 shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", default=None, choices=sorted(WORKLOADS), help="CPU baseline workload (default: the bench workload itself)")
    ap.add_argument("--cpu-procs", type=int, default=8, help="independent CPU provers run side by side for the host throughput figure (0 = skip)")
    ap.add_argument("--semantics", default="reference", choices=["reference", "public"],
                    help="reference: new random generators for every proof + inner-product argument down to length 1 (reference src/verifier.cpp:119-128; the default "
                         "since round 6); public: hash-to-curve generators with a resident byte table, argument cut at 256 (the headline of rounds 1-5)")
    ap.add_argument("--hybrid-tail", action="store_true", help="timed proofs with ZKCNN_MODE_HOST_TAIL (experiment; not the headline configuration)")
    ap.add_argument("--host-rounds", action="store_true", help="timed proofs with ZKCNN_MODE_HOST_ROUNDS: a kernel launch per round, no resident round kernel (experiment)")
    ap.add_argument("--fiat-shamir", action="store_true", help="timed proofs non-interactive (device-side rounds; experiment)")
    ap.add_argument("--no-companions", action="store_true", help="skip the extra single-stream measurements of other modes (profiling runs)")
    ap.add_argument("--profile-all", action="store_true", help="print the per-kernel-class table of one extra proof to stderr")
    ap.add_argument("--rehearse-shared-gpu", action="store_true",
                    help="REHEARSAL of the multi-rank path on a box with fewer GPUs than ranks: ranks share GPUs (rank %% GPUs) and exchange over gloo "
                         "(RCCL refuses two ranks on one device). Not a scaling measurement; the line says so")
    ap.add_argument("--no-pmc", action="store_true", help="skip the in-run rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--inner", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.pmc_child:
        return pmc_child(args.workload)
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    self_launch_if_needed(args, sys.argv[1:])
    if args.gpus == 1 and "RANK" not in os.environ and not args.inner:
        return supervise(sys.argv[1:])

    import torch
    import zkcnn_amd
    from zkcnn_amd import dp

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if args.rehearse_shared_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    numa_node = pin_to_gpu_numa_node(torch, local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:        # launched by torch.distributed.run (also with one rank: same code path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # no device_id: the RCCL communicator (its streams and hardware queues) is then created by the first collective -- the barrier in front of
        # the timed region, AFTER the sessions and their streams exist. Created first it costs the proving streams a fifth of their throughput
        # (71.7 vs 88.1 proofs/s on one rank; the same creation-order effect as for the sessions themselves).
        if args.rehearse_shared_gpu:
            dist.init_process_group(backend="gloo")
        elif os.environ.get("ZKCNN_BENCH_EAGER_RCCL"):
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="nccl")
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: launch one rank per GPU (or run it without RANK set: it launches them itself)")

    model, pic, pp = WORKLOADS[args.workload]
    LANES = max(1, min(args.lanes, 8))
    # round 6: 8 batches of 8 -- two per hardware queue, evenly, since all batch streams are created back to back (BatchSession.group): every batch then proves at the same
    # pace (profiles/r06_variance.md, r06_queues_grouped.txt: 4 / 6 / 8 batches 162.6 / 165.7 / 172.4 proofs/s over public generators, 84.6 / 98.0 / 101.8 in the reference's
    # semantics). Round 5: 6 batches (made one by one they landed 2 + 2 + 1 + 1 on the four queues: 4 / 5 / 6 / 7 / 8 batches 147 / 155 / 161-163 / 154-157).
    K = max(1, args.streams if args.streams else (8 * LANES if LANES > 1 else 8))
    try:                                # do not overcommit a small node
        import psutil
        K_ram = streams_that_fit(psutil.virtual_memory().available, world, K, clones=LANES > 1)
        if K_ram < K:
            print(f"[bench] host memory allows {K_ram} sessions per rank, not the {K} asked for", file=sys.stderr)
            K = K_ram
    except ImportError:
        pass
    REF = args.semantics == "reference" and not args.fiat_shamir          # (Fiat-Shamir proofs are always over the public generators: their digest is part of the hashed statement)
    public_mode = zkcnn_amd.MODE_DRIVE_ONLY | zkcnn_amd.MODE_REUSE_GENS
    reference_mode = zkcnn_amd.MODE_DRIVE_ONLY | zkcnn_amd.MODE_FULL_IPA
    drive = reference_mode if REF else public_mode
    other_mode = public_mode if REF else reference_mode                   # the companion configuration

    def proof_seed(base, i, k):
        """challenge seed of proof k of session i. Reference semantics: a seed of its own for every proof -- the verifier draws the generators from its seeded
        stream, and two proofs under one seed would share a generator set (and its tables)."""
        return ((base << 20) | (i << 10) | k) if REF else base + k
    if args.hybrid_tail:
        drive |= zkcnn_amd.MODE_HOST_TAIL
    if args.fiat_shamir:
        drive |= zkcnn_amd.MODE_FIAT_SHAMIR
    if args.host_rounds or args.rehearse_shared_gpu:       # (ranks that share a GPU are not alone on it: the policy counts proofs per PROCESS)
        drive |= zkcnn_amd.MODE_HOST_ROUNDS

    def stage(name):
        """progress on stderr: if a run dies, the log says where"""
        if rank == 0:
            print(f"[bench] {time.strftime('%H:%M:%S')} {name}", file=sys.stderr, flush=True)
        if os.environ.get("ZKCNN_BENCH_ABORT_AT") == name:       # test hook of the supervising process: die here
            os.abort()

    result_file = os.environ.get("ZKCNN_BENCH_RESULT") if args.inner else None

    def leave(make, incomplete_after=None):
        """the line so far (or the final one) for the supervising process"""
        if result_file and rank == 0:
            out = make()
            if incomplete_after:
                out = dict(out, incomplete_after=incomplete_after)
            with open(result_file + ".tmp", "w") as f:
                f.write(json.dumps(out))
            os.replace(result_file + ".tmp", result_file)

    # at most this many sessions prove INDEPENDENTLY at a time (set-up, warm-up: every session on its own stream): the timed region drives 7 batches
    # from 7 threads, and 56 threads launching side by side is a shape nothing needs -- under rocprofv3 it dies inside hipLaunchKernel, twice out
    # of twice (profiles/r04_rocprof_sigsegv_56threads.log)
    side_by_side = threading.Semaphore(16)

    def in_threads(fn):
        """fn(i) for every stream i, at most 16 at a time, on a POOL of 16 host threads (the C calls release the GIL); re-raises the first failure.
        (Until round 5 every i had a thread of its own behind a semaphore: K threads alive at once, each of which launches kernels sooner or later --
        the one thing that separates the shapes rocprofv3 traces from those it dies in: 32 sessions trace, 40 / 48 / 56 die inside hipLaunchKernel
        during this warm-up, old and new tree alike, `profiles/r05_rocprof_threads.md`.)"""
        errs = []
        todo = queue.Queue()
        for i in range(K):
            todo.put(i)

        def worker():
            while True:
                try:
                    i = todo.get_nowait()
                except queue.Empty:
                    return
                try:
                    fn(i)
                except BaseException as e:      # noqa: BLE001 - reported below
                    errs.append(e)
        if os.environ.get("ZKCNN_BENCH_THREAD_PER_SESSION"):      # the shape of rounds 3-5 (experiment switch)
            def run(i):
                try:
                    with side_by_side:
                        fn(i)
                except BaseException as e:      # noqa: BLE001
                    errs.append(e)
            th = [threading.Thread(target=run, args=(i,)) for i in range(K)]
        else:
            # (under rocprofv3 -- its tool library is preloaded -- four: the fewer threads launch side by side, the rarer its crash; see the docstring)
            traced = any("rocprof" in os.environ.get(v, "").lower() for v in ("LD_PRELOAD", "ROCP_TOOL_LIBRARIES", "HSA_TOOLS_LIB"))
            th = [threading.Thread(target=worker) for _ in range(min(int(os.environ.get("ZKCNN_BENCH_POOL", "4" if traced else "16")), K))]
        [t.start() for t in th]
        [t.join() for t in th]
        if errs:
            raise errs[0]

    # ---- K sessions per GPU: one image (circuit + witness), host thread and HIP stream each ----
    t0 = time.time()
    sessions = [None] * K

    # One model = one circuit: every session of the job proves pictures under the SAME weights (data seed) and quantisation scales. Session 0 is
    # built from the data; the others are built for the same statement (calibrated: circuit + witness under session 0's scales) and ATTACH to the
    # circuit already resident on the GPU -- no second gate sort / upload, one set of generator tables per GPU -- and then take pictures of their
    # own through new_image below. (Rounds 1-2 built K unrelated circuits per GPU: K uploads of 1.9 GB, K byte tables of 3.2 GB.)
    def build(i):
        sessions[i] = zkcnn_amd.Session(model, pic, pp, data_seed=DATA_SEED, device=local_rank)
    # Every session is built from the same data: the same circuit, so ONE of them sorts and uploads it and the others -- built side by side, on
    # their own host threads -- wait for it and attach. (Side by side matters: with the sessions created one after the other the same 8 streams
    # reach 79 instead of 97 proofs/s, reproducibly since round 1 -- how HIP maps streams to hardware queues depends on it; cause not established.)
    # Large single-circuit workloads do not fit 8 times: a probe session measures what the first and what one more session cost in HBM.
    free0, _ = torch.cuda.mem_get_info(local_rank)
    hbm_first_gb = hbm_extra_gb = None
    # Session 0 generates the circuit (host), sorts and uploads it; with lock-step batches every other session is a CLONE of it (zkcnn_session_clone):
    # a context of its own on the resident circuit with a copy of the witness in HBM and NO host copy of the circuit -- 0.1 s and ~0.1 GB of host
    # memory each instead of 3.5 s and 2.7 GB (round 3 built K full sessions side by side: K host circuits per rank). A probe clone measures what one
    # more session costs in HBM: large single-circuit workloads do not fit 32 times. Independent streams (--lanes 1) keep the round-3 shape: there the
    # order in which the sessions' streams are created matters.
    if LANES > 1:
        build(0)
        free1, _ = torch.cuda.mem_get_info(local_rank)
        probe = sessions[0].clone()
        free2, _ = torch.cuda.mem_get_info(local_rank)
        probe.close()
        hbm_first_gb, hbm_extra_gb = round((free0 - free1) / 1e9, 2), round((free1 - free2) / 1e9, 2)
        K_fit = 1 + max(0, int((0.9 * free1 - 8e9) / (max(free1 - free2, 1) * 1.2)))        # margin: MSM scratch (~1.5 GB per session at 2^26 inputs) and the byte table come with the first proofs
        if K_fit < K:
            print(f"[bench] HBM allows {K_fit} sessions of this workload, not the {K} asked for", file=sys.stderr)
            K = K_fit
        LANES = min(LANES, K)
        K = (K // LANES) * LANES                # whole batches
        B = K // LANES
        sessions = sessions[:K]

        def build_clone(i):
            if i:
                sessions[i] = sessions[0].clone()
        in_threads(build_clone)
    else:
        if pp > 1 or "vgg16" in model:
            build(0)
            free1, _ = torch.cuda.mem_get_info(local_rank)
            probe = zkcnn_amd.Session(model, pic, pp, data_seed=DATA_SEED, device=local_rank)
            free2, _ = torch.cuda.mem_get_info(local_rank)
            probe.close()
            sessions[0].close()
            sessions[0] = None
            hbm_first_gb, hbm_extra_gb = round((free0 - free1) / 1e9, 2), round((free1 - free2) / 1e9, 2)
            K_fit = 1 + max(0, int((0.9 * free1 - 8e9) / (max(free1 - free2, 1) * 1.2)))
            if K_fit < K:
                print(f"[bench] HBM allows {K_fit} sessions of this workload, not the {K} asked for", file=sys.stderr)
                K = K_fit
        B = K
        sessions = sessions[:K]
        in_threads(build)
    if any(x is None for x in sessions):
        raise SystemExit("a session could not be built")
    setup_s = time.time() - t0
    free3, _ = torch.cuda.mem_get_info(local_rank)
    shared_gb = zkcnn_amd.sharing_stats()["shared_circuit_gb"]
    hbm_all_gb = round((free0 - free3) / 1e9, 2)
    if hbm_extra_gb is None:            # (no probe: what the sessions hold for themselves, evenly)
        hbm_extra_gb = round((hbm_all_gb - shared_gb) / K, 2)
        hbm_first_gb = round(shared_gb + hbm_extra_gb, 2)
    sess = sessions[0]
    try:
        import psutil
        import resource
        host_rss_gb = round(psutil.Process().memory_info().rss / 1e9, 2)                      # all K sessions of this rank, resident now
        host_peak_gb = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss * 1024 / 1e9, 2)     # peak while they were being built side by side
    except Exception:       # noqa: BLE001
        host_rss_gb = host_peak_gb = None

    stage("pictures")
    # ---- a picture of its own for every session but the first (which keeps the data stream's picture: the parity proof below is compared with
    # the CPU oracle's): the recorded witness program replays in HBM (zkcnn_session_new_image); pictures whose activation ranges ask for other
    # quantisation scales than the circuit's are refused and the next seed is tried ----
    valid = [[] for _ in range(K)]
    tried = [0] * K

    def scan(i):
        p = 0
        while len(valid[i]) < 2 and p < 32:
            p += 1
            seed = 100000 * (rank * K + i + 1) + p
            if sessions[i].new_image(seed)[0] == 0:
                valid[i].append(seed)
        tried[i] = p
        if len(valid[i]) < 2:
            raise RuntimeError("no two pictures with the circuit's quantisation scales among 32")
        sessions[i].new_image(valid[i][0])

    def scan_rest(i):
        if i:
            scan(i)
    t_scan = time.time()
    if pp == 1:                         # (new pictures for one-picture circuits; a pic_cnt > 1 circuit keeps the batch it was built for)
        try:
            in_threads(scan_rest)
        except Exception as e:          # noqa: BLE001 - the sessions then all prove the first picture: say so
            print(f"[bench] no distinct pictures: {e}", file=sys.stderr)
            valid = [[] for _ in range(K)]
    t_scan = time.time() - t_scan
    distinct_pictures = all(len(v) >= 2 for v in valid[1:])

    stage("warm-up")
    # ---- warm-up: first step with the full verifier on every image (acceptance), the rest as the timed steps run ----
    firsts = [None] * K

    def warm(i):
        for w in range(max(args.warmup, 2)):      # at least two: the second use of a generator set builds the MSM byte table
            if w == 0:
                firsts[i], _ = sessions[i].prove(seed=proof_seed(0x5EED0001, i, 0), mode=zkcnn_amd.MODE_FULL_IPA if REF else zkcnn_amd.MODE_REUSE_GENS, want_transcript=False)
                if firsts[i].accepted != 1:
                    raise SystemExit(f"verifier rejected the GPU proof: {firsts[i].message.decode()}")
            else:
                sessions[i].prove(seed=proof_seed(0x5EED0001, i, w), mode=drive, want_transcript=False)
    in_threads(warm)
    first, accepted = firsts[0], True
    # full-size byte parity inside the driver's evidence: session 0 of rank 0 proves its picture under the challenge seed and the modes the CPU
    # oracle leg below uses (same data seed): the two canonical transcripts must be the same bytes (compared through their SHA-256)
    import hashlib
    parity_sha = parity_len = None
    if rank == 0:
        _, ptr = sess.prove(seed=PARITY_SEED, mode=drive, want_transcript=True)
        parity_sha, parity_len = hashlib.sha256(ptr).hexdigest(), len(ptr)
        del ptr

    # one profiled proof to find the dominant kernel class (events on every launch; not timed)
    sess.profile("all")
    sess.prove(seed=0x5EED00FF, mode=drive, want_transcript=False)
    table = sess.profile_report(reset=True)
    # the class the HBM roofline is reported for: the byte-streaming classes (the commitment's MSM is integer-ALU work by construction, SURVEY 8(d):
    # it gets its own entry, roofline.msm, against the multiplier ceiling)
    dominant = max(HBM_CLASSES, key=lambda c: table[c]["ms"])
    alg_bytes_per_proof = sum(table[c]["bytes"] for c in ROOFLINE_CLASSES)     # every class with a defined byte count, one proof
    if args.profile_all and rank == 0:
        for k, v in sorted(table.items(), key=lambda kv: -kv[1]["ms"]):
            if v["launches"]:
                print(f"  {k:14s} {v['ms']:10.3f} ms {v['launches']:7d} launches {v['bytes'] / 1e9:10.3f} GB(alg)", file=sys.stderr)
    # single-stream latency (the "prover ms/image" half of the metric): one proof at a time on an otherwise idle GPU,
    # with the same events on the dominant class as the timed steps carry
    sess.profile([dominant])
    lat_prove = lat_poly = 0.0
    LAT_REPS = 3
    for k in range(LAT_REPS):
        r, _ = sess.prove(seed=0x5EED0100 + k, mode=drive, want_transcript=False)
        lat_prove += r.prove_s / LAT_REPS
        lat_poly += r.poly_prove_s / LAT_REPS
    lat_prof = sess.profile_report(reset=True)[dominant]
    sess.profile([dominant])           # during the timed steps the dominant class carries events on ONE of the streams (stream 0)
    if os.environ.get("ZKCNN_BENCH_NOEVENTS"):
        sess.profile(None)

    # ---- the round-3 shape for comparison (not the headline): the first 8 sessions as independent proofs, a host thread and a HIP stream each ----
    indep = {}
    if LANES > 1 and rank == 0 and world == 1 and not args.no_companions:
        n_ind = min(8, K)
        torch.cuda.synchronize()
        t_i = time.perf_counter()
        errs_i = []

        def indep_stream(i):
            try:
                for k in range(args.steps):
                    sessions[i].prove(seed=proof_seed(0x5EED0900, i, k), mode=drive, want_transcript=True)
            except BaseException as e:      # noqa: BLE001
                errs_i.append(e)
        th_i = [threading.Thread(target=indep_stream, args=(i,)) for i in range(n_ind)]
        [t.start() for t in th_i]
        [t.join() for t in th_i]
        torch.cuda.synchronize()
        if not errs_i:
            indep = {"proofs_per_s_independent_streams": round(n_ind * args.steps / (time.perf_counter() - t_i), 3), "independent_streams": n_ind}

    # ---- lock-step batches: LANES sessions share one host thread, one stream and one launch per round (zkcnn_amd.BatchSession) ----
    batches, batch_table = [], None
    if LANES > 1:
        # (all batch streams first, then the lanes: an even spread over the hardware queues -- profiles/r06_variance.md)
        batches = (zkcnn_amd.BatchSession.group([sessions[j * LANES:(j + 1) * LANES] for j in range(B)]) if not os.environ.get("ZKCNN_BENCH_BATCHES_ONE_BY_ONE")
                   else [zkcnn_amd.BatchSession(sessions[j * LANES:(j + 1) * LANES]) for j in range(B)])

        def warm_batch(j):
            batches[j].prove(seeds=[proof_seed(0x5EED0A00, j * LANES + i, 0) if REF else 0x5EED0A00 + i for i in range(LANES)], mode=drive, want_transcript=False)
        errs_b = []

        def run_b(j):
            try:
                warm_batch(j)
            except BaseException as e:      # noqa: BLE001
                errs_b.append(e)
        th_b = [threading.Thread(target=run_b, args=(j,)) for j in range(B)]
        if os.environ.get("ZKCNN_BENCH_SERIAL_WARM"):          # experiment: the batches' first proofs (and with them the streams' first launches) one after the other
            for t in th_b:
                t.start()
                t.join()
        else:
            [t.start() for t in th_b]
            [t.join() for t in th_b]
        if errs_b:
            raise errs_b[0]
        # one batch proof with nothing else on the GPU and events on every class: what a FUSED launch of each kernel class costs uncontended
        sess.profile_report(reset=True)          # (events the session collected on its own before it became a lane)
        sess.profile("all")
        batches[0].prove(seeds=[proof_seed(0x5EED0A80, i, 0) if REF else 0x5EED0A80 + i for i in range(LANES)], mode=drive, want_transcript=False)
        batch_table = sess.profile_report(reset=True)
        sess.profile([dominant])
        if os.environ.get("ZKCNN_BENCH_NOEVENTS"):
            sess.profile(None)
        fusion0 = batches[0].stats()
        host_rounds0 = sess.host_tail_rounds()

    stage("timed steps")
    # ---- timed region: `steps` steps, a step = K proofs in flight on this GPU (B batches of LANES lanes; or one per stream) ----
    coll_dev = "cpu" if args.rehearse_shared_gpu else "cuda"        # gloo exchanges host tensors

    def barrier():
        if dist is not None:
            dist.barrier() if args.rehearse_shared_gpu else dist.barrier(device_ids=[local_rank])
    barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    prove_s = poly_s = 0.0
    gatherer = dp.AsyncGather(dist, coll_dev, K << 19) if dist is not None and not os.environ.get("ZKCNN_BENCH_NOGATHER") else None
    done = [queue.Queue() for _ in range(K)]
    fail = []

    EVENT_STEPS = min(3, args.steps)       # stream 0 carries the events of the dominant class for its first timed proofs only
    prof_box = {}

    # every timed proof produces its canonical transcript (N = 1 like N > 1: the same work at every point of a scaling curve)
    def stream(i):
        try:
            for k in range(args.steps):
                if i == 0 and k == EVENT_STEPS:
                    prof_box["report"] = sess.profile_report(reset=True)[dominant]
                    sess.profile(None)
                done[i].put([sessions[i].prove(seed=proof_seed(0x5EED1000, i, k), mode=drive, want_transcript=True)])
        except BaseException as e:          # noqa: BLE001
            fail.append(e)
            done[i].put(None)

    batch_wall = [0.0] * max(B, 1)

    def batch_stream(j):
        try:
            pin_worker(j, B)
            for k in range(args.steps):
                if j == 0 and k == EVENT_STEPS:
                    prof_box["report"] = sess.profile_report(reset=True)[dominant]
                    sess.profile(None)
                done[j].put(batches[j].prove(seeds=[proof_seed(0x5EED1000, j * LANES + i, k) for i in range(LANES)], mode=drive, want_transcript=True))
                batch_wall[j] += batches[j].wall_s
        except BaseException as e:          # noqa: BLE001
            fail.append(e)
            done[j].put(None)
    n_workers = B if LANES > 1 else K
    workers = [threading.Thread(target=batch_stream if LANES > 1 else stream, args=(i,)) for i in range(n_workers)]
    [t.start() for t in workers]
    for k in range(args.steps):
        got = [done[i].get() for i in range(n_workers)]
        if fail:
            raise fail[0]
        batch = [x for part in got for x in part]           # the K proofs of the step, in session order
        prove_s += sum(r.prove_s for r, _ in batch)
        poly_s += sum(r.poly_prove_s for r, _ in batch)
        if gatherer is not None:
            # the step's only exchange: this GPU's K proofs to rank 0 over RCCL, overlapped with the next proofs (zkcnn_amd/dp.py)
            gatherer.submit(rank, dp.pack([(rank * K + i, tr) for i, (_, tr) in enumerate(batch)]))
    [t.join() for t in workers]
    rank_busy_s = time.perf_counter() - t_start          # this rank's own proving time (no collective, no other rank in it)
    fusion = {}
    if batches:
        f1 = batches[0].stats()
        fusion = {"fused_launches_per_proof": round((f1["fused_launches"] - fusion0["fused_launches"]) / args.steps / LANES, 1),
                  "fused_launches_per_batch_proof": round((f1["fused_launches"] - fusion0["fused_launches"]) / args.steps, 1),
                  "lanes_per_fused_launch": round((f1["lane_launches"] - fusion0["lane_launches"]) / max(f1["fused_launches"] - fusion0["fused_launches"], 1), 2),
                  "rounds_per_proof": firsts[0].n_rounds,
                  "rounds_on_host_per_proof": round((sess.host_tail_rounds() - host_rounds0) / args.steps, 1),
                  "host_tail": "a lane hands a phase's tables to the host once they hold <= 32 entries (2 KB): the phase's last 5 rounds are ~150 multiplications on the batch's "
                               "host thread instead of 5 launches (sumcheck.hip: policy::LANE_TAIL_LOG; same field elements, transcripts byte-identical); "
                               "proofs_per_s_every_round_on_gpu is the same run without it (ZKCNN_MODE_GPU_TAIL)",
                  "note": "batch 0 over the timed steps: every deferred launch of a batch proof is ONE launch over all its lanes"}
    last_proofs = [tr for _, tr in batch]          # the last timed proof of every stream, checked after the clock stops
    gather_wait_s = 0.0
    if gatherer is not None:
        t_g = time.perf_counter()
        gathered = gatherer.wait()
        gather_wait_s = time.perf_counter() - t_g      # what was left of the asynchronous gathers once the last proof was done
        if rank == 0:
            assert len(gathered) == args.steps and all(len(g) == world for g in gathered)
            assert all(len(dp.unpack(blob)) == K for g in gathered for _, blob in g)
    torch.cuda.synchronize()
    barrier()
    elapsed = time.perf_counter() - t_start
    # what a scaling run needs to attribute a slow rank: its own rate, sessions, gather wait, set-up time, NUMA node of its GPU and the HBM it has left
    headroom_gb = torch.cuda.mem_get_info(local_rank)[0] / 1e9
    per_rank = [{"rank": 0, "proofs_per_s": round(K * args.steps / rank_busy_s, 3), "streams": K, "gather_wait_s": round(gather_wait_s, 4),
                 "setup_s": round(setup_s, 1), "numa_node": int(numa_node if numa_node is not None else -1), "hbm_free_gb": round(headroom_gb, 1)}]
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
        # attribution of a sub-linear scaling figure: every rank's own rate, its stream count and how long it waited for the gather
        mine = torch.tensor([rank_busy_s, float(K), gather_wait_s, setup_s, float(numa_node if numa_node is not None else -1), headroom_gb], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": r, "proofs_per_s": round(float(t[1]) * args.steps / max(float(t[0]), 1e-9), 3), "streams": int(t[1]),
                     "gather_wait_s": round(float(t[2]), 4), "setup_s": round(float(t[3]), 1), "numa_node": int(t[4]), "hbm_free_gb": round(float(t[5]), 1)}
                    for r, t in enumerate(a.cpu() for a in allr)]
        rates = [p["proofs_per_s"] / max(p["streams"], 1) for p in per_rank]
        if rank == 0 and min(rates) > 0 and max(rates) / min(rates) > 1.1:
            print(f"[bench] per-session rates of the ranks differ by {max(rates) / min(rates):.2f}x: " + ", ".join(f"rank {p['rank']} (NUMA {p['numa_node']}): {p['proofs_per_s']}" for p in per_rank),
                  file=sys.stderr)

    # the timed steps ran drive-only: replay the last proof of every stream through the full verifier (not timed)
    replay_mode = (zkcnn_amd.MODE_FULL_IPA if REF else zkcnn_amd.MODE_REUSE_GENS) | (zkcnn_amd.MODE_FIAT_SHAMIR if args.fiat_shamir else 0)
    timed_ok = all(sessions[i].verify(last_proofs[i], seed=proof_seed(0x5EED1000, i, args.steps - 1), mode=replay_mode).accepted == 1 for i in range(K))
    if not timed_ok:
        raise SystemExit("a proof produced inside the timed region does not verify")

    prof = {"ms": 0.0, "launches": 0, "bytes": 0.0}
    pr = prof_box.get("report") or sess.profile_report(reset=True)[dominant]
    for key in prof:
        prof[key] += pr[key]
    sess.profile(None)

    roofline, cpu, parity, extras = None, None, {}, {}

    def build_out():
        """the bench line from what has been measured so far (stages that have not run yet leave their defaults)"""
        steps = args.steps
        out = {
            "metric": ("proofs/s of the GKR prover in the REFERENCE's semantics: the verifier draws new random generators for every proof (reference src/verifier.cpp:119-128; "
                       "every commitment table is built inside the prover's clock, none survives the proof; the verifier's own loop gens[i] = k_i G multiplies on the prover's GPU, "
                       "zk_fixed_base_mul, ZKCNN_HOST_GENERATORS=1 keeps it on host threads) and the inner-product argument runs down to length 1; " if REF else
                       "proofs/s of the GKR prover, PUBLIC-GENERATOR VARIANT (hash-to-curve generators with a resident byte table, inner-product argument cut at 256; the reference's "
                       "own semantics are the `reference_semantics` object of this line); ") +
                      f"{args.workload} pic_cnt={pp} proofs, {K} in flight per GPU" +
                      (f" as {B} lock-step batches of {LANES} lanes: one host thread, one HIP stream and ONE kernel launch per sumcheck round per batch" if LANES > 1 else "") +
                      ("; modes SEEDED|DRIVE_ONLY|FULL_IPA, a challenge seed (= a generator set) of its own for every proof" if REF else "; modes SEEDED|DRIVE_ONLY|REUSE_GENS") +
                      ("; EXPERIMENT: hybrid host tail" if args.hybrid_tail else "") + ("; EXPERIMENT: Fiat-Shamir (challenges hashed on the host)" if args.fiat_shamir else "") +
                      "; sessions share one resident circuit, a picture each; every timed proof returns its transcript, the last of every session is replay-verified; prover_ms_per_image = "
                      "single-stream latency in the same semantics (a lone proof runs its rounds in resident kernels); proofs_per_s_every_round_on_gpu = the same shape without the "
                      "lanes' host tail" + ("; `public_generators` = the variant rounds 1-5 reported as `value`" if REF else ""),
            "value": round(sum(p["streams"] for p in per_rank) * steps / elapsed, 4),         # every rank's proofs (a rank may hold fewer streams than asked for)
            "unit": "proofs/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u256 (BLS12-381 Fr, 8x32-bit Montgomery limbs)", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {model} {pic[0]}x{pic[1]}x{pic[2]} pic_cnt={pp}, one proof per image, {K} images in flight per GPU",
                       "images_per_step": sum(p["streams"] for p in per_rank), "streams_per_gpu": K,
                       "layers": first.n_layers, "input_size": first.input_size, "rounds": first.n_rounds,
                       "mul_gates": first.gate_cnt_bin, "add_gates": first.gate_cnt_uni, "lanes_per_batch": LANES, "batches_per_gpu": B if LANES > 1 else 0,
                       "parallelism": (f"dp{world} x {B} lock-step batches x {LANES} lanes (independent proofs, one launch per round per batch, RCCL gather)" if LANES > 1
                                       else f"dp{world} x {K} streams (independent proofs, RCCL gather)")},
            "prover_ms_per_image": round(1e3 * (lat_prove + lat_poly), 3),
            "prover_ms_sumcheck": round(1e3 * lat_prove, 3), "prover_ms_commit": round(1e3 * lat_poly, 3),
            "prover_ms_per_image_in_flight": round(1e3 * (sum(batch_wall) / (steps * B) if LANES > 1 else (prove_s + poly_s) / (steps * K)), 3),
            # wall per batch proof of every batch (its host thread's view): batches that share a hardware queue with another show up here (variance study, profiles/r06_variance.md)
            "batch_wall_ms": [round(1e3 * w / steps, 1) for w in batch_wall] if LANES > 1 else None,
            "verifier_pass": bool(accepted), "timed_proofs_replay_verified": K, "proof_kb": round(first.proof_kb + first.poly_proof_kb, 1),
            "setup_s": round(setup_s, 1), "witness_s": round(first.witness_s, 1), "upload_sort_s": round(max(f.upload_s for f in firsts), 2),
            "hbm_gb_all_sessions": hbm_all_gb, "hbm_gb_first_session": hbm_first_gb, "hbm_gb_per_extra_session": hbm_extra_gb, "sharing": dict(zkcnn_amd.sharing_stats(), shared_circuit_gb=shared_gb), "distinct_picture_per_session": bool(distinct_pictures),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        out.update(parity)
        out["per_rank"] = per_rank
        if args.rehearse_shared_gpu:
            out["rehearsal"] = f"{world} ranks share {torch.cuda.device_count()} GPU(s), gloo exchange: a rehearsal of the launch / gather / timing path, NOT a scaling measurement"
        out["host_rss_gb_all_sessions"] = host_rss_gb
        out["host_peak_rss_gb_while_building"] = host_peak_gb
        out.update(extras)
        if REF:
            # the variant rounds 1-5 reported as `value`, next to the reference's own protocol
            pub = {"proofs_per_s": extras.get("proofs_per_s_public_generators"), "prover_ms_per_image": extras.get("prover_ms_public_generators"),
                   "cpu_oracle_ms": (cpu or {}).get("other_mode_ms"), "speedup_vs_one_core": (cpu or {}).get("gpu_speedup_vs_one_core_other_mode"),
                   "note": "public hash-to-curve generators (nobody knows a discrete log) shared by every proof, their byte table resident in HBM (built once, outside the clock), "
                           "inner-product argument cut at 256 (the last 256 scalars travel in the clear): the configuration rounds 1-5 reported as `value`"}
            out = {"metric": out["metric"], "value": out["value"], "unit": out["unit"], "semantics": "reference", "public_generators": pub,
                   **{k: v for k, v in out.items() if k not in ("metric", "value", "unit")}}
        else:
            ref = {"proofs_per_s": extras.get("proofs_per_s_fresh_gens_full_ipa"), "prover_ms_per_image": extras.get("prover_ms_fresh_gens_full_ipa"),
                   "cpu_oracle_ms": (cpu or {}).get("other_mode_ms"), "speedup_vs_one_core": (cpu or {}).get("gpu_speedup_vs_one_core_other_mode"),
                   "note": "fresh random generators drawn by the verifier for every proof (no table survives a proof), full inner-product argument; same circuit, same pictures, "
                           "transcripts equal to the CPU oracle's in this mode (tests/test_full_size_gpu.py)"}
            out = {"metric": out["metric"], "value": out["value"], "unit": out["unit"], "semantics": "public", "reference_semantics": ref,
                   **{k: v for k, v in out.items() if k not in ("metric", "value", "unit")}}
        return out

    leave(build_out, "timed region (replay-verified); companions, roofline, PMC passes and CPU baseline not run yet")
    stage("new-picture companion")
    # ---- companion: a NEW picture for every proof (the timed steps above re-prove K resident witnesses). zkcnn_session_new_image replays
    # the recorded witness program in HBM: quantise the picture on the host, recompute every layer value and auxiliary witness on the GPU.
    # Same K streams, same modes; each step = new_image + prove per stream. Pictures whose activation ranges ask for other quantisation
    # scales need a new circuit (a new session) and are skipped by the scan; the scan's first call uploads the program (not timed). ----
    new_image = {}
    if rank == 0 and world == 1 and not args.no_companions:          # companions belong to the single-GPU line (like the CPU baseline)
        try:
            if not distinct_pictures or pp != 1:
                raise RuntimeError("no pictures with the circuit's scales were found")
            scan(0)                   # (session 0 kept the data stream's picture until here: its parity proof is behind us)
            single = sorted(sess.new_image(valid[0][k % 2])[1] for k in range(7))
            ni_last = [None] * K

            def stream_new(i):
                for k in range(args.steps):
                    if sessions[i].new_image(valid[i][k % 2])[0] != 0:
                        raise RuntimeError("new_image refused a picture it accepted before")
                    ni_last[i] = sessions[i].prove(seed=proof_seed(0x5EED2000, i, k), mode=drive, want_transcript=True)[1]

            def batch_new(j):            # a new picture in every lane (its witness is replayed in HBM on the batch's stream), then the batch's proof
                for k in range(args.steps):
                    for i in range(j * LANES, (j + 1) * LANES):
                        if sessions[i].new_image(valid[i][k % 2])[0] != 0:
                            raise RuntimeError("new_image refused a picture it accepted before")
                    for i, (_, tr) in enumerate(batches[j].prove(seeds=[proof_seed(0x5EED2000, j * LANES + i, k) for i in range(LANES)], mode=drive, want_transcript=True)):
                        ni_last[j * LANES + i] = tr
            torch.cuda.synchronize()
            t_ni = time.perf_counter()
            if LANES > 1:
                errs_n = []

                def run_n(j):
                    try:
                        batch_new(j)
                    except BaseException as e:      # noqa: BLE001
                        errs_n.append(e)
                th_n = [threading.Thread(target=run_n, args=(j,)) for j in range(B)]
                [t.start() for t in th_n]
                [t.join() for t in th_n]
                if errs_n:
                    raise errs_n[0]
            else:
                in_threads(stream_new)
            torch.cuda.synchronize()
            t_ni = time.perf_counter() - t_ni
            ok = all(sessions[i].verify(ni_last[i], seed=proof_seed(0x5EED2000, i, args.steps - 1), mode=replay_mode).accepted == 1 for i in range(K))
            new_image = {"new_image_ms": round(single[len(single) // 2], 3),
                         "proofs_per_s_new_picture_each_proof": round(K * args.steps / t_ni, 3),
                         "new_picture_proofs_replay_verified": K if ok else 0,
                         "pictures_tried_per_accepted": round(sum(tried) / (2.0 * K), 2),
                         "program_upload_and_scan_s": round(t_scan, 2)}
        except Exception as e:      # noqa: BLE001 - the headline does not depend on this
            new_image = {"new_image_error": str(e)}
    # ---- companion: the timed configuration WITHOUT the lanes' hybrid tail: every round of every phase is a kernel launch (round 4's first shape) ----
    gpu_tail = {}
    if rank == 0 and world == 1 and not args.no_companions and LANES > 1:
        try:
            gt_steps = max(2, min(args.steps, 6))
            for x in batches:            # (untimed: first use of the mode)
                x.prove(seeds=[proof_seed(0x5EED5000, batches.index(x) * LANES + i, 0) for i in range(LANES)], mode=drive | zkcnn_amd.MODE_GPU_TAIL, want_transcript=False)
            torch.cuda.synchronize()
            t_gt = time.perf_counter()
            errs_g = []
            last_g = [None] * B

            def run_g(j):
                try:
                    for k in range(gt_steps):
                        last_g[j] = batches[j].prove(seeds=[proof_seed(0x5EED5100, j * LANES + i, k) for i in range(LANES)], mode=drive | zkcnn_amd.MODE_GPU_TAIL, want_transcript=True)
                except BaseException as e:      # noqa: BLE001
                    errs_g.append(e)
            th_g = [threading.Thread(target=run_g, args=(j,)) for j in range(B)]
            [t.start() for t in th_g]
            [t.join() for t in th_g]
            torch.cuda.synchronize()
            t_gt = time.perf_counter() - t_gt
            if errs_g:
                raise errs_g[0]
            # the two configurations produce the same bytes: lane 0 of batch 0 against a proof of the default configuration under the same seed
            same = batches[0].prove(seeds=[proof_seed(0x5EED5100, i, gt_steps - 1) for i in range(LANES)], mode=drive, want_transcript=True)[0][1] == last_g[0][0][1]
            gpu_tail = {"proofs_per_s_every_round_on_gpu": round(K * gt_steps / t_gt, 3), "every_round_on_gpu_transcript_identical": bool(same)}
        except Exception as e:      # noqa: BLE001 - the headline does not depend on this
            gpu_tail = {"every_round_on_gpu_error": str(e)}
    # ---- companion: the OTHER semantics. Timed steps in the reference's semantics (the default): the public-generator variant (hash-to-curve generators
    # with a resident byte table, argument cut at 256: the headline of rounds 1-5). Timed steps over public generators: the reference's semantics
    # (reference src/verifier.cpp:119-128: new random generators for every proof, argument down to length 1, every table built inside the clock) ----
    ref_mode = {}
    ref_steps = 0
    other_key = "proofs_per_s_public_generators" if REF else "proofs_per_s_fresh_gens_full_ipa"
    if rank == 0 and world == 1 and not args.no_companions:
        try:
            ref_steps = max(2, min(args.steps, 4))

            def other_seeds(j, k):
                # (fresh generators: a seed of its own for every proof -- the verifier draws the generators from its seeded stream)
                return [0x5EED4000 + k] * LANES if REF else [(0x5EED4000 << 20) | ((j * LANES + i) << 10) | k for i in range(LANES)]
            if REF:                 # (untimed: the second use of the public generator set builds its byte table)
                for w in range(2):
                    for j in range(B if LANES > 1 else K):
                        if LANES > 1:
                            batches[j].prove(seeds=[0x5EED3F00 + w] * LANES, mode=other_mode, want_transcript=False)
                        else:
                            sessions[j].prove(seed=0x5EED3F00 + w, mode=other_mode, want_transcript=False)
            torch.cuda.synchronize()
            t_rf = time.perf_counter()
            errs_r = []

            def run_r(j):
                try:
                    for k in range(ref_steps):
                        if LANES > 1:
                            batches[j].prove(seeds=other_seeds(j, k), mode=other_mode, want_transcript=True)
                        else:
                            sessions[j].prove(seed=other_seeds(j, k)[0], mode=other_mode, want_transcript=True)
                except BaseException as e:      # noqa: BLE001
                    errs_r.append(e)
            th_r = [threading.Thread(target=run_r, args=(j,)) for j in range(B if LANES > 1 else K)]
            [t.start() for t in th_r]
            [t.join() for t in th_r]
            torch.cuda.synchronize()
            if errs_r:
                raise errs_r[0]
            ref_mode = {other_key: round(K * ref_steps / (time.perf_counter() - t_rf), 3)}
        except Exception as e:      # noqa: BLE001 - the headline does not depend on this
            ref_mode = {other_key + "_error": str(e)}
    for x in batches:
        x.close()
    batches = []
    for x in sessions[1:]:
        x.close()

    roofline = None
    if prof["launches"]:
        sec = prof["ms"] * 1e-3 / prof["launches"]
        achieved = prof["bytes"] / prof["launches"] / sec / 1e9
        roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": None,
                    "avg_launch_ms": round(sec * 1e3, 4), "launches_per_step": prof["launches"] / EVENT_STEPS,
                    "note": (f"HIP events on batch 0 of {B} batches during its first {EVENT_STEPS} timed batch proofs: a launch is FUSED over {LANES} lanes (bytes = all lanes'); durations include contention between the batches"
                             if LANES > 1 else f"HIP events on stream 0 of {K} streams during its first {EVENT_STEPS} timed proofs: launch durations include contention between the streams"),
                    "algorithmic_bytes_per_launch": prof["bytes"] / prof["launches"],
                    "share_of_prover_time": round(prof["ms"] * 1e-3 / EVENT_STEPS / max((sum(batch_wall) / (args.steps * B)) if LANES > 1 else (prove_s + poly_s) / (K * args.steps), 1e-12), 3)}
        if lat_prof["launches"]:
            s1 = lat_prof["ms"] * 1e-3 / lat_prof["launches"]
            a1 = lat_prof["bytes"] / lat_prof["launches"] / s1 / 1e9
            roofline["single_stream"] = {"achieved": round(a1, 1), "frac": round(a1 / HBM_PEAK_GBS, 4), "avg_launch_ms": round(s1 * 1e3, 4),
                                         "note": "the same class with one proof at a time (the latency measurement above)"}

    # the same kernel family on one large streaming launch (two 2^24-entry tables, device resident): this is the figure
    # to read against the HBM roof; the whole-proof average above is dominated by ~1.1 k tiny latency-bound rounds
    if roofline is not None and rank == 0:
        try:
            hc = zkcnn_amd.HipContext(local_rank)
            sec, nbytes = hc.bench_round_quadratic(24, 10)
            mul_sec = hc.bench_fr_mul(1 << 20, 256, 3)
            hc.close()
            roofline["streaming_launch"] = {"kernel": "k_round_quad2 (the product round kernel: fold + sums + grid finish + host slot), V and M of 2^24 entries", "ms": round(sec * 1e3, 4),
                                            "algorithmic_bytes": nbytes, "achieved": round(nbytes / sec / 1e9, 1),
                                            "frac": round(nbytes / sec / 1e9 / HBM_PEAK_GBS, 4),
                                            "fr_mul_ceiling_G_per_s": round((1 << 20) * 256 / mul_sec / 1e9, 1)}
            # products the launch performs: a quad (4 + 4 entries in, 2 + 2 out = 384 B) is 4 folds + 2 sum products (b comes from the running claim;
            # 3 without it) -> 6 per quad; in multiply-accumulate steps 4 x 120 + 2 x 64 = 608 against the ceiling kernel's 128 per product
            quads = (1 << 24) / 4
            sl = roofline["streaming_launch"]
            sl["fr_mul_per_s"] = round(quads * 6 / sec / 1e9, 1)
            sl["frac_of_mul_ceiling"] = round(quads * 6 / sec / ((1 << 20) * 256 / mul_sec), 4)
            sl["mac_steps_frac_of_ceiling"] = round(quads * 608 / sec / (128 * (1 << 20) * 256 / mul_sec), 4)
            sl["note"] = ("6 Montgomery products per 384-B quad (fr_mul_per_s in G/s); the products of the streaming kernel are cheaper than the ceiling kernel's "
                          "(uniform factor, 512-bit sums): mac_steps_frac_of_ceiling counts multiply-accumulate steps. The launch is bound by its memory side "
                          "(profiles/r05_round_lab.md: the same accesses without arithmetic take 0.30 ms, the arithmetic without accesses 0.25 ms)")
        except Exception as e:      # the headline numbers do not depend on this extra measurement
            roofline["streaming_launch"] = {"error": str(e)}

    if roofline is not None and batch_table:
        # every byte-counted class of ONE batch proof that had the GPU to itself (fused launches over all lanes): launches, time, bytes -> GB/s
        cls = {}
        for c in ROOFLINE_CLASSES:
            t = batch_table[c]
            if t["launches"]:
                gbs = t["bytes"] / max(t["ms"] * 1e-3, 1e-12) / 1e9
                cls[c] = {"launches_per_batch_proof": t["launches"], "ms_per_batch_proof": round(t["ms"], 3), "algorithmic_GB": round(t["bytes"] / 1e9, 3),
                          "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4), "avg_launch_ms": round(t["ms"] / t["launches"], 4)}
        roofline["classes"] = cls
        roofline["classes_note"] = (f"one batch proof ({LANES} lanes) alone on the GPU, HIP events on every launch: round_quad = the product kernel on tables above 2^16 entries "
                                    "(the streaming class: read against the HBM roof / the multiplier ceiling), round_fine = the 4-lanes-per-quad latency kernel of the small rounds "
                                    "(one launch per round for all lanes: a dependent chain, not a stream), msm_planes = the commitment (integer ALU)")
    if roofline is not None:
        # whole-GPU view of the multi-stream regime: algorithmic bytes of one proof (all byte-counted classes) x proofs/s of this GPU
        per_gpu = K * args.steps / elapsed
        roofline["whole_gpu"] = {"algorithmic_GB_per_proof": round(alg_bytes_per_proof / 1e9, 2), "achieved": round(alg_bytes_per_proof * per_gpu / 1e9, 1),
                                 "frac": round(alg_bytes_per_proof * per_gpu / 1e9 / HBM_PEAK_GBS, 4),
                                 "note": "ALGORITHMIC bytes (SURVEY 8(d)) of gate_reduce + round_quad + round_cubic + msm of one proof times proofs/s per GPU; "
                                         "most gate_reduce gathers hit L2, so this is NOT HBM traffic (PMC: see profiles/)"}
    if rank != 0:
        sess.close()
        if dist is not None:
            dist.destroy_process_group()
        return

    extras = dict(new_image)
    extras.update(indep)
    extras.update(ref_mode)
    extras.update(gpu_tail)
    if fusion:
        extras["fusion"] = fusion
    leave(build_out, "new-picture companion")
    stage("companions")
    # ---- conservative companions of the headline (not timed steps): nothing pre-built, nothing cut ----
    try:
        if args.no_companions or world > 1:
            raise KeyboardInterrupt
        fresh = []
        for k in range(2):          # new random generators per proof (no REUSE_GENS): tables rebuilt inside the prover's clock; IPA down to length 1
            r, _ = sess.prove(seed=0x5EED0200 + k, mode=reference_mode, want_transcript=False)
            fresh.append(1e3 * (r.prove_s + r.poly_prove_s))
        extras["prover_ms_fresh_gens_full_ipa"] = round(min(fresh), 3)
        r, _ = sess.prove(seed=0x5EED0210, mode=public_mode, want_transcript=False)      # back on the session generators (tables are rebuilt once)
        r, _ = sess.prove(seed=0x5EED0211, mode=public_mode, want_transcript=False)
        pub = [sess.prove(seed=0x5EED0213 + k, mode=public_mode, want_transcript=False)[0] for k in range(3)]
        extras["prover_ms_public_generators"] = round(min(1e3 * (x.prove_s + x.poly_prove_s) for x in pub), 3)
        r, _ = sess.prove(seed=0x5EED0212, mode=public_mode | zkcnn_amd.MODE_FULL_IPA, want_transcript=False)
        extras["prover_ms_session_gens_full_ipa"] = round(1e3 * (r.prove_s + r.poly_prove_s), 3)
        # other modes of the same prover, single stream, best of 3 (hybrid tail / every round on the GPU: the timed semantics; Fiat-Shamir and zero knowledge: public generators)
        def best(mode):
            print(f"[bench] companion mode {mode:#x}", file=sys.stderr, flush=True)
            return round(min(1e3 * (x.prove_s + x.poly_prove_s) for x in (sess.prove(seed=0x5EED0300 + k, mode=mode, want_transcript=False)[0] for k in range(3))), 3)
        extras["prover_ms_hybrid_tail"] = best(drive | zkcnn_amd.MODE_HOST_TAIL)          # tables of <= 64 entries finish their phase on the host
        # the lone proof with EVERY round on the GPU (no hand-over of the last <= 32 entries of a phase to the host: the resident tail kernel runs each phase to its
        # end, the shape of rounds 3-5): what prover_ms_per_image would be without the host's share of the small rounds (round-4 review: quote both)
        extras["prover_ms_every_round_on_gpu"] = best(drive | zkcnn_amd.MODE_GPU_TAIL)
        fs = zkcnn_amd.MODE_FIAT_SHAMIR | zkcnn_amd.MODE_DRIVE_ONLY
        extras["prover_ms_fiat_shamir"] = best(fs)                                        # non-interactive: challenges hashed on the host, rounds in the resident kernels
        extras["prover_ms_fiat_shamir_device_rounds"] = best(fs | zkcnn_amd.MODE_FS_DEVICE)  # ... small rounds and their hash chain on the GPU by themselves
        extras["prover_ms_fiat_shamir_host_rounds"] = best(fs | zkcnn_amd.MODE_HOST_ROUNDS)
        extras["prover_ms_zero_knowledge"] = best(public_mode | zkcnn_amd.MODE_ZK)        # blinded commitments, masked rounds, proofs of dot product (public generators)
        if REF:
            extras["prover_ms_zero_knowledge_fresh_gens_full_ipa"] = best(reference_mode | zkcnn_amd.MODE_ZK)
        extras["fs_device_rounds_phases"] = list(sess.fs_stats())
    except KeyboardInterrupt:
        pass
    except Exception as e:          # noqa: BLE001 - the headline does not depend on these
        extras["companions_error"] = str(e)
    sess.close()
    sess = None

    leave(build_out, "companions")
    stage("roofline")
    # ---- the roof that binds: this is integer modular arithmetic, and on this workload (a CIFAR-sized circuit: ~1.1 k interactive rounds, most
    # of them on small tables) the dominant class is bound by LATENCY -- dependent field products and the round hand-over -- not by HBM or by
    # the multiplier. Reported: HBM fraction of the dominant class (above), measured HBM traffic of that class, and the fraction of the
    # integer-multiplier ceiling the whole proof reaches (Fr-multiply equivalents per proof / proofs per second / measured ceiling). ----
    if roofline is not None:
        mul_ceiling = (roofline.get("streaming_launch") or {}).get("fr_mul_ceiling_G_per_s")
        bytes_of = {c: table[c]["bytes"] for c in ROOFLINE_CLASSES}
        # Fr-multiply equivalents of one proof from the algorithmic byte counts the library keeps per launch -- a LOWER bound, classes
        # without a byte count (small convolution kernels, the opening's tiny MSMs) are left out:
        #   fold rounds: 96 B per entry per round = 1.5 products (4 lerps + 2..3 products per 4-entry quad); cubic rounds 1.75
        #   gate sums: 80 B per multiplication gate = ~1.5 products (the gather operand, the scale)
        #   eq tables: ~2 products per entry of every layer table (one per claim point); commitment: one mixed addition = 11 Fp products
        #   (12-limb: 288 + 288 MACs against the 136 + 136 of an Fr product = 2.12 Fr equivalents) per non-zero scalar byte, ~0.72 of the scalars
        fr_muls = ((bytes_of["round_quad"] + bytes_of["round_fine"]) / 96.0 * 1.5 + bytes_of["round_cubic"] / 96.0 * 1.75 + bytes_of["gate_reduce"] / 80.0 * 1.5 +
                   2.0 * float(first.table_entries) + bytes_of["msm_planes"] / 32.0 * 0.72 * 11 * 2.12)
        per_gpu = K * args.steps / elapsed
        if mul_ceiling:
            roofline["alu"] = {"fr_mul_equiv_per_proof": round(fr_muls), "ceiling_G_per_s": mul_ceiling,
                               "frac": round(fr_muls * per_gpu / (mul_ceiling * 1e9), 4),
                               "frac_single_stream": round(fr_muls / max(lat_prove + lat_poly, 1e-9) / (mul_ceiling * 1e9), 4),
                               "note": "Fr-multiply equivalents per proof (lower bound from the per-launch algorithmic counts) x proofs/s per GPU over the multiplier "
                                       "ceiling measured in this run (k_bench_fr_mul: dependent Montgomery products, 8 waves per SIMD)"}
        msm_t = table["msm_planes"]
        if mul_ceiling and msm_t["ms"] > 0:
            scalars = msm_t["bytes"] / 32.0
            fp_equiv = scalars * 0.72 * 11 * 2.12          # mixed additions x 11 Fp products x 2.12 Fr equivalents (see above)
            roofline["msm"] = {"bound": "alu", "class_ms_per_proof": round(msm_t["ms"], 3), "launches": msm_t["launches"], "scalars_per_s": round(scalars / (msm_t["ms"] * 1e-3)),
                               "frac_of_multiplier_ceiling": round(fp_equiv / (msm_t["ms"] * 1e-3) / (mul_ceiling * 1e9), 3),
                               "note": "the commitment's scalar pre-pass and MSM kernels of one single-stream proof (HIP events): integer-ALU work, not HBM traffic"}
        small = roofline["frac"] < 0.15 and (not roofline.get("alu") or roofline["alu"]["frac_single_stream"] < 0.5)
        roofline["bound"] = "latency" if small else roofline["bound"]
        roofline["bound_note"] = ("per launch the dominant class reaches a fraction of the HBM roof: most of its launches are short dependent chains behind a launch and a host "
                                  "hand-over, and in the timed shape they share the GPU with six other batches (classes.round_quad: the same launches uncontended; time against "
                                  "size: profiles/r05_launch_sizes.md); the SAME kernel on a 2 x 2^24-entry launch (streaming_launch) runs at 0.6 of the HBM peak = 0.8 of a plain copy "
                                  "with its read:write mix, and at streaming_launch.frac_of_mul_ceiling of the multiplier: neither roof is reached, VALU issue is the nearer one "
                                  "(profiles/r05_round_quad_pmc.md)") if small else "see streaming_launch"

    leave(build_out, "roofline (HIP events); CPU baseline, the transcript comparison with it and the PMC traffic passes not run yet")
    stage("cpu baseline")
    # ---- CPU baseline: the oracle (port of the reference prover) on the same workload ----
    cpu = None
    parity = {}
    if not args.no_cpu_baseline and world == 1:        # the host baseline belongs to the single-GPU line (rank 0, N = 1)
        import multiprocessing as mp
        cm, cpic, cpp = WORKLOADS[args.cpu_sample or args.workload]
        job = (cm, cpic, cpp, DATA_SEED, drive)
        # ... and in the other semantics (the public-generator variant next to the reference's, or the other way round), side by side on a second core
        job_ref = (cm, cpic, cpp, DATA_SEED, other_mode)
        ctx = mp.get_context("spawn")
        with ctx.Pool(2) as pool:                          # (i) one core each, nothing else running
            (one_s, one_wall, gates, one_sum, one_poly, cpu_sha, cpu_len), ref_cpu = pool.map(_cpu_prover_worker, [job, job_ref])
        other_ms = extras.get("prover_ms_public_generators" if REF else "prover_ms_fresh_gens_full_ipa")
        cpu = {"value": round(1e3 * one_s, 1), "unit": "prover ms/image", "cores": 1, "kind": "port",
               "sample": f"{args.cpu_sample or args.workload} ({cm}), pic_cnt={cpp}, {gates} mul gates -- the full bench workload: CPU oracle prover time "
                         f"(sumcheck {1e3 * one_sum:.0f} ms + Hyrax {1e3 * one_poly:.0f} ms), same modes as the timed GPU proofs"
                         + (" (the reference's semantics: fresh generators, full argument)" if REF else ""),
               "gpu_speedup_vs_one_core": round(1e3 * one_s / max(1e3 * (lat_prove + lat_poly), 1e-9), 1),
               "other_mode_ms": round(1e3 * ref_cpu[0], 1),
               "gpu_speedup_vs_one_core_other_mode": round(1e3 * ref_cpu[0] / other_ms, 1) if other_ms else None,
               "other_mode": ("the public-generator variant (argument cut at 256)" if REF else "fresh random generators per proof (reference src/verifier.cpp:119-128), inner-product argument down to length 1")
                             + ": CPU oracle prover time over the GPU's lone proof in that mode",
               "host_cores_available": os.cpu_count()}
        if (args.cpu_sample or args.workload) == args.workload:
            parity = {"transcript_equal_to_cpu_oracle": bool(parity_sha == cpu_sha and parity_len == cpu_len), "transcript_sha256_gpu": parity_sha,
                      "transcript_sha256_cpu_oracle": cpu_sha, "transcript_bytes": parity_len,
                      "parity_proof": f"session 0: data seed {DATA_SEED}, challenge seed {PARITY_SEED:#x}, the modes of the timed proofs; full workload"}
            if not parity["transcript_equal_to_cpu_oracle"]:
                leave(build_out, "FAILED: the GPU transcript differs from the CPU oracle's on the full workload")
                raise SystemExit(f"GPU transcript differs from the CPU oracle's on the full workload: {parity}")
        n_proc = max(0, min(args.cpu_procs, (os.cpu_count() or 1)))
        try:
            import psutil
            n_proc = min(n_proc, int(psutil.virtual_memory().available / 14e9))
        except ImportError:
            pass
        if n_proc >= 2:                                    # (ii) independent single-threaded provers side by side
            t0 = time.time()
            with ctx.Pool(n_proc) as pool:
                rs = pool.map(_cpu_prover_worker, [(cm, cpic, cpp, DATA_SEED + i, drive) for i in range(n_proc)])
            wall = time.time() - t0
            per = sum(r[0] for r in rs) / n_proc
            cores = os.cpu_count() or 1
            cpu["multi_process"] = {"processes": n_proc, "prover_s_each": round(per, 2), "wall_s_incl_witness": round(wall, 1),
                                    "proofs_per_s_measured": round(n_proc / per, 3),
                                    "proofs_per_s_if_every_core_ran_one": round(cores / per, 2),
                                    "gpu_proofs_per_s_over_that": round((world * K * args.steps / elapsed) / max(cores / per, 1e-9), 2),
                                    "note": f"{n_proc} independent oracle processes at once (prover time only, contention included); the all-cores figure "
                                            f"extrapolates to {cores} processes and needs ~{cores * 10} GB of host memory -- an upper bound for the host"}

    # ---- last, because it is the stage most likely to go wrong (counter collection on this platform): HBM traffic of the dominant class ----
    if roofline is not None and world == 1 and not args.no_pmc and dominant in CLASS_KERNELS:
        extras["only_optional_stages_left"] = True        # (for the supervising process: a failure from here on costs roofline.traffic, not the run)
        leave(build_out, "CPU baseline; PMC traffic passes not run yet (roofline.traffic is null)")
        extras.pop("only_optional_stages_left")
        stage("pmc passes")
        pmc = measure_pmc_traffic(args.workload, CLASS_KERNELS[dominant])
        if pmc:
            roofline["traffic"] = round(pmc["bytes_per_launch"], 1)
            roofline["traffic_source"] = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over {pmc['launches']} launches of {', '.join(pmc['kernels'])} in four "
                                          f"single-stream proofs with every round a launch; counter units calibrated IN THE SAME PASS on the same kernel's 2 x 2^24-entry launch (known bytes): "
                                          f"{pmc['bytes_per_counter_unit']} bytes per unit (the guide's x2 for FETCH_SIZE is for 16-byte-per-lane streams; the calibration launch IS this kernel, so its own access width is what gets calibrated)")
            roofline["traffic_over_algorithmic"] = round(pmc["bytes_per_launch"] / max(lat_prof["bytes"] / max(lat_prof["launches"], 1), 1.0), 3)
        else:
            roofline["traffic_source"] = "rocprofv3 not available (or the counter pass failed): not measured"

    out = build_out()
    leave(lambda: out)
    if not result_file:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
