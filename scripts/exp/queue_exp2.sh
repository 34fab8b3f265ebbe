#!/bin/bash
cd /root/repo
run() { name=$1; shift; extra=$1; shift
  for i in 1 2 3; do
    v=$(env "$@" python bench.py --no-cpu-baseline --no-companions --no-pmc $extra 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['value'])")
    echo "$name run $i: $v"
  done
}
run hwq2 "" GPU_MAX_HW_QUEUES=2
run hwq3 "" GPU_MAX_HW_QUEUES=3
run streams64 "--streams 64" X=1
run streams48 "--streams 48" X=1
