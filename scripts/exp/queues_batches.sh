#!/bin/bash
# hardware queues x lock-step batches: does a batch stream get a hardware queue of its own, and what does a shared one cost?  -> gpurun_out/<tag>_queues.txt
# usage: [SPECS='queues streams lanes;queues streams lanes;...'] scripts/exp/queues_batches.sh <tag> [semantics ...]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=${1:-q}; shift
OUT=$ROOT/gpurun_out/${TAG}_queues.txt
: > $OUT
cd $ROOT
specs=("4 48 8" "6 48 8" "8 48 8" "4 32 8" "8 64 8" "4 64 8" "4 48 6" "8 48 6")
[ -n "$SPECS" ] && IFS=";" read -r -a specs <<< "$SPECS"
for SEM in ${@:-reference public}; do
for spec in "${specs[@]}"; do
  IFS=' ' read -r Q ST LN <<< "$spec"
  GPU_MAX_HW_QUEUES=$Q timeout 400 python bench.py --streams $ST --lanes $LN --steps 6 --warmup 2 --semantics $SEM --no-cpu-baseline --no-companions --no-pmc 2> /tmp/q.err | tail -1 > /tmp/q.json
  python3 - "$SEM queues $Q in flight $ST lanes $LN" >> $OUT <<'PY'
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
