#!/bin/bash
# hardware queues x lock-step batches: does a batch stream get a hardware queue of its own, and what does a shared one cost?  -> gpurun_out/<tag>_queues.txt
# usage: scripts/exp/queues_batches.sh <tag> [semantics ...]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=${1:-q}; shift
OUT=$ROOT/gpurun_out/${TAG}_queues.txt
: > $OUT
cd $ROOT
for SEM in ${@:-reference public}; do
for spec in "4 48" "6 48" "8 48" "4 32" "8 64" "6 32" "8 32" "12 48"; do
  set -- $spec
  GPU_MAX_HW_QUEUES=$1 timeout 400 python bench.py --streams $2 --lanes 8 --steps 6 --warmup 2 --semantics $SEM --no-cpu-baseline --no-companions --no-pmc 2> /tmp/q.err | tail -1 > /tmp/q.json
  python3 - "$SEM queues $1 in flight $2" >> $OUT <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/q.json").read())
    print(f'{sys.argv[1]:36s} | {d["value"]:7.2f} proofs/s | batch walls {d.get("batch_wall_ms")} | lone {d["prover_ms_per_image"]:.1f} ms')
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/q.err").read()[-300:].replace("\n", " | "))
PY
done
done
cat $OUT
