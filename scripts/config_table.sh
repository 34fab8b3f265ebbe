#!/bin/bash
# bench.py over the BASELINE.json configurations -> gpurun_out/<tag>_configs.jsonl (one JSON line per run)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-configs}
OUT=$ROOT/gpurun_out/${TAG}_configs.jsonl
mkdir -p $ROOT/gpurun_out; : > $OUT
cd $ROOT
# "<workload> <proofs in flight> <lanes per lock-step batch>": a lone proof, the round-3 shape (8 independent streams), the round-4 shape (batches of 8 lanes;
# bench.py lowers the count to what fits HBM for the single-circuit workloads)
for spec in "lenet 1 1" "lenet 8 1" "lenet 32 8" "vgg11 1 1" "vgg11 8 1" "vgg11 32 8" "vgg16 8 1" "vgg16 32 8" "vgg11_pp8 1 1" "vgg11_pp8 8 1" "vgg11_pp8 16 4" "vgg16_pp4 1 1" "vgg16_pp4 8 1" "vgg16_pp4 16 4"; do
  set -- $spec
  timeout 900 python bench.py --workload $1 --streams $2 --lanes $3 --steps 5 --warmup 2 --no-cpu-baseline --no-companions --no-pmc 2>/dev/null | tail -1 >> $OUT
done
python3 - <<PY
import json
for l in open("$OUT"):
    try: d = json.loads(l)
    except Exception: print("bad line", l[:80]); continue
    c = d["config"]
    print(f'{c["workload"][:60]:60s} in flight {c["streams_per_gpu"]:3d} lanes {c.get("lanes_per_batch", 1)} | {d["value"]:8.2f} proofs/s | latency {d["prover_ms_per_image"]:8.1f} ms (sumcheck {d["prover_ms_sumcheck"]:.1f} + commit {d["prover_ms_commit"]:.1f}) | layers {c["layers"]} rounds {c["rounds"]} input 2^{(c["input_size"]-1).bit_length()} mul {c["mul_gates"]:.2e} | proof {d["proof_kb"]} KB | setup {d["setup_s"]} s | pass {d["verifier_pass"]}')
PY
