set -x
cd /root/repo
mkdir -p gpurun_out/dig
timeout 900 python -m pytest tests/test_batch_gpu.py tests/test_zk_gpu.py tests/test_sharing_gpu.py -x -q 2>&1 | grep -E "passed|failed|Error|error" | tail -15 > gpurun_out/dig/tests.txt
for dp in 0 1; do
  ZKCNN_DIGIT_OPENING=$dp timeout 600 python bench.py --no-cpu-baseline --no-companions --no-pmc > gpurun_out/dig/bench_d$dp.json 2> gpurun_out/dig/bench_d$dp.err
done
for p in 4 16 32; do
  ZKCNN_DIGIT_PAIRS=$p timeout 600 python bench.py --no-cpu-baseline --no-companions --no-pmc > gpurun_out/dig/bench_p$p.json 2> gpurun_out/dig/bench_p$p.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/dig/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d.get('batch_wall_ms'))
    except Exception as e: print(f, 'ERR', e)
PY
cat gpurun_out/dig/tests.txt
