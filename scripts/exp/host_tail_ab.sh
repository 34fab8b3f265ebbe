#!/bin/bash
# lock-step batches with the hybrid tail (ZKCNN_MODE_HOST_TAIL: a phase's last rounds on the host once its tables are small) at several hand-over sizes
# against the default; usage: scripts/exp/host_tail_ab.sh [steps]
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
STEPS=${1:-8}
BASE=3                      # DRIVE_ONLY | REUSE_GENS
HT=$((BASE | (1 << 26)))
for rep in 1 2; do
  echo "default:       $(python $ROOT/scripts/exp/batch_mode.py $BASE 8 $STEPS 4 2>&1 | grep -E "lanes:|single" | sed 's/\[mode [0-9a-fx]*\] //' | tr '\n' ' ')"
  for L in 3 4 5 6; do
    echo "host tail 2^$L: $(ZKCNN_HOST_TAIL_LOG=$L python $ROOT/scripts/exp/batch_mode.py $HT 8 $STEPS 4 2>&1 | grep -E "lanes:|single" | sed 's/\[mode [0-9a-fx]*\] //' | tr '\n' ' ')"
  done
done
