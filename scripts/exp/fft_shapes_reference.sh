#!/bin/bash
# the FFT-convolution circuits (BASELINE configs 4 / 5) in the reference's semantics, best shapes of profiles/r06_fft_shapes.txt -> gpurun_out/<tag>_fft_reference.txt
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=${1:-r06}
OUT=$ROOT/gpurun_out/${TAG}_fft_reference.txt
mkdir -p $ROOT/gpurun_out; : > $OUT
cd $ROOT
for spec in "vgg11_pp8 6 2" "vgg16_pp4 4 2" "vgg11_pp8 1 1"; do
  set -- $spec
  timeout 280 python bench.py --workload $1 --streams $2 --lanes $3 --steps 4 --warmup 2 --semantics reference --no-cpu-baseline --no-companions --no-pmc 2> /tmp/fft_shape.err | tail -1 > /tmp/fft_shape.json
  python3 - "$spec" >> $OUT <<'PY'
import json, sys
try:
    d = json.loads(open("/tmp/fft_shape.json").read())
    c = d["config"]
    pp = int(c["workload"].split("pic_cnt=")[1].split(",")[0])
    print(f'{sys.argv[1]:18s} -> in flight {c["streams_per_gpu"]} lanes {c.get("lanes_per_batch")} | {d["value"]:7.2f} proofs/s = {d["value"] * pp:6.1f} pictures/s | lone {d["prover_ms_per_image"]:6.1f} ms '
          f'(sumcheck {d["prover_ms_sumcheck"]:.1f} + commit {d["prover_ms_commit"]:.1f}) | semantics {d.get("semantics")} | pass {d["verifier_pass"]}')
except Exception as e:
    print(sys.argv[1], "FAILED", e, open("/tmp/fft_shape.err").read()[-300:].replace("\n", " | "))
PY
done
cat $OUT
