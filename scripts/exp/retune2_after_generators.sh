#!/bin/bash
ROOT=/root/repo
OUT=$ROOT/gpurun_out/r06_retune2.txt
: > $OUT
cd $ROOT
run() {
  local name="$1"; shift
  env "$@" timeout 400 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-companions --no-pmc 2> /tmp/r.err | tail -1 > /tmp/r.json
  python3 - "$name" >> $OUT <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/r.json").read())
    print(f'{sys.argv[1]:44s} | {d["value"]:7.2f} proofs/s | batch walls {d.get("batch_wall_ms")}')
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/r.err").read()[-300:].replace("\n", " | "))
PY
}
run "digit affine per 64" ZKCNN_DIGIT_AFFINE_PER=64
run "digit affine per 128" ZKCNN_DIGIT_AFFINE_PER=128
run "lone launch shapes" ZKCNN_LOAD_SHAPES=0
run "opening through window tables" ZKCNN_DIGIT_OPENING=0
run "generators on host threads" ZKCNN_HOST_GENERATORS=1
run "default" A=1
cat $OUT
