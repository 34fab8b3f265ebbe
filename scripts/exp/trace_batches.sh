#!/bin/bash
# kernel trace of the default bench shape (8 x 8 lanes, reference semantics) + scripts/exp/trace_concurrency.py on it -> gpurun_out/<tag>_concurrency.txt
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=${1:-r06}
mkdir -p $ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
D=/tmp/trace_$TAG; rm -rf $D
rocprofv3 --kernel-trace -d $D -o trace -- python $ROOT/bench.py --inner --steps ${STEPS:-4} --warmup 1 --no-cpu-baseline --no-companions --no-pmc $EXTRA > $ROOT/gpurun_out/${TAG}_trace.log 2>&1
python $ROOT/scripts/exp/trace_concurrency.py $D > $ROOT/gpurun_out/${TAG}_concurrency.txt 2>&1
cat $ROOT/gpurun_out/${TAG}_concurrency.txt
