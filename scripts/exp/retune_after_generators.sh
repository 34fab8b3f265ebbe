#!/bin/bash
# the policy switches again, now that the batches' streams are busy (the verifier's generator loop is off the host): one bench run each, one box
# -> gpurun_out/<tag>_retune.txt
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=${1:-r06}
OUT=$ROOT/gpurun_out/${TAG}_retune.txt
: > $OUT
cd $ROOT
run() {
  local name="$1"; shift
  env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-companions --no-pmc $EXTRA 2> /tmp/r.err | tail -1 > /tmp/r.json
  python3 - "$name" >> $OUT <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/r.json").read())
    print(f'{sys.argv[1]:44s} | {d["value"]:7.2f} proofs/s | batch walls {d.get("batch_wall_ms")}')
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/r.err").read()[-300:].replace("\n", " | "))
PY
}
run "default" A=1
run "lanes resident tail" ZKCNN_LANE_RESIDENT=1
EXTRA="--hybrid-tail" run "host tail at 2^6" A=1
run "digit pairs 32" ZKCNN_DIGIT_PAIRS=32
run "digit pairs 16" ZKCNN_DIGIT_PAIRS=16
run "fine log 15" ZKCNN_TEST_HOOKS=1 ZKCNN_TEST_FINE_LOG=15
run "fine log 17" ZKCNN_TEST_HOOKS=1 ZKCNN_TEST_FINE_LOG=17
run "default again" A=1
cat $OUT
