#!/bin/bash
# A/B on one box: digits per inversion of the digit table's conversion to affine under load (ZKCNN_DIGIT_AFFINE_PER), default bench shape, reference semantics
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/dig
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_batch_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/dig/tests_affine.txt
for p in 16 64 128 64 16; do
  ZKCNN_DIGIT_AFFINE_PER=$p timeout 600 python bench.py --no-cpu-baseline --no-companions --no-pmc > gpurun_out/dig/bench_a$p.json 2> gpurun_out/dig/bench_a$p.err
  python - <<PY
import json
try:
    d = json.loads(open('gpurun_out/dig/bench_a$p.json').read().strip().splitlines()[-1]); print('per $p', d['value'], d.get('batch_wall_ms'), d.get('transcript_equal_to_cpu_oracle'))
except Exception as e: print('per $p ERR', e)
PY
done
cat gpurun_out/dig/tests_affine.txt
