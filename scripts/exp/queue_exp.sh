#!/bin/bash
cd /root/repo
run() { # name, env...
  name=$1; shift
  for i in 1 2 3 4 5; do
    v=$(env "$@" python bench.py --no-cpu-baseline --no-companions --no-pmc 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$name run $i: $v"
  done
}
run default X=1
run serial_warm ZKCNN_BENCH_SERIAL_WARM=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq8_serial GPU_MAX_HW_QUEUES=8 ZKCNN_BENCH_SERIAL_WARM=1
run hwq7 GPU_MAX_HW_QUEUES=7
