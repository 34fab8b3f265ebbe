#!/bin/bash
cd /root/repo
run() { name=$1; shift; extra=$1; shift
  for i in 1 2 3; do
    v=$(env "$@" python bench.py --no-cpu-baseline --no-companions --no-pmc $extra 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$name run $i: $v"
  done
}
run streams40 "--streams 40" X=1
run streams48 "--streams 48" X=1
run streams56_pinned "" ZKCNN_BENCH_PIN_WORKERS=1
run streams48_pinned "--streams 48" ZKCNN_BENCH_PIN_WORKERS=1
run streams32 "--streams 32" X=1
