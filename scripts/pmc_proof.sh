#!/bin/bash
# HBM traffic of every kernel of one whole proof from PMC counters (FETCH_SIZE and WRITE_SIZE in separate passes, as
# MI355X_MICROARCH.md prescribes) next to the kernel durations of a plain kernel trace.  usage: scripts/pmc_proof.sh <workload> <tag>
# FETCH_SIZE is doubled (gfx950 tallies the 128-B requests of wide coalesced reads at 64 B); WRITE_SIZE is used as reported.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WL=${1:-vgg11_pp8}; TAG=${2:-pmc_proof}
OUT=$ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --inner --workload $WL --streams 1 --steps 2 --warmup 2 --no-cpu-baseline --no-companions"
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || true
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- $CMD > $OUT/$C.log 2>&1 || true
done
python3 - <<PY
import csv, glob, collections
def short(n): return n.split("(")[0].replace("void ", "")[:40]
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"]); d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        dur[k][0] += 1; dur[k][1] += d
cnt = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot = collections.defaultdict(float)
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == c: tot[short(r["Kernel_Name"])] += float(r["Counter_Value"])
    cnt[c] = tot
lines = ["# PMC HBM traffic per kernel, whole run of: $CMD", "",
         "FETCH_SIZE / WRITE_SIZE are KB counters collected in separate passes; FETCH is doubled (gfx950 correction for wide coalesced reads,",
         "MI355X_MICROARCH.md) -- an upper bound for kernels whose reads are narrow gathers. Durations from a third, counter-free pass.", "",
         "| kernel | launches | total ms | fetch GB (x2) | write GB | traffic GB/s | frac of 8 TB/s |", "|---|---|---|---|---|---|---|"]
for k, (n, us) in sorted(dur.items(), key=lambda kv: -kv[1][1])[:22]:
    f = cnt["FETCH_SIZE"].get(k, 0) * 1024 * 2 / 1e9; w = cnt["WRITE_SIZE"].get(k, 0) * 1024 / 1e9
    bw = (f + w) / (us * 1e-6) if us else 0
    lines.append(f"| {k} | {n} | {us/1e3:.2f} | {f:.2f} | {w:.2f} | {bw:.0f} | {bw/8000:.3f} |")
open("$OUT/summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
