#!/bin/bash
# SQ wave-cycle breakdown of the throughput kernels of one workload (one PMC pass: SQ counters only)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WL=${1:-vgg11}; TAG=${2:-pmc_stalls}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/sq -o pmc -- \
  python $ROOT/bench.py --workload $WL --streams 1 --steps 2 --warmup 2 --no-cpu-baseline --no-companions > $OUT/sq.log 2>&1 || true
python3 - <<PY
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$OUT/sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        tot[r["Kernel_Name"].split("(")[0].replace("void ", "")[:32]][r["Counter_Name"]] += float(r["Counter_Value"])
lines = ["| kernel | wave cycles | waiting (s_waitcnt / barrier) | issue stalled | issuing | of which VALU | VALU insts per wave |", "|---|---|---|---|---|---|---|"]
for k, c in sorted(tot.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:10]:
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    lines.append(f'| {k} | {wc:.3g} | {c.get("SQ_WAIT_ANY", 0) / wc:.2f} | {c.get("SQ_WAIT_INST_ANY", 0) / wc:.2f} | {c.get("SQ_ACTIVE_INST_ANY", 0) / wc:.2f} | {c.get("SQ_ACTIVE_INST_VALU", 0) / wc:.2f} | {c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1):.0f} |')
open("$OUT/summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
