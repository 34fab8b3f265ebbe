#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, one counter per pass (MI355X_MICROARCH.md "HBM"):
# FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x (corrected in the parser).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/pmc_r1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/$C -o pmc -- python $ROOT/scripts/bench_round_kernel.py 24 5 > $OUT/$C.log 2>&1 || true
done
python - <<PY
import csv, glob, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob("$OUT/%s/**/*counter_collection.csv" % c, recursive=True)
    tot = collections.defaultdict(list)
    for f in files:
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == c and "k_round_quad" in row.get("Kernel_Name", ""):
                tot[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    res[c] = {k: (sum(v) / len(v), len(v)) for k, v in tot.items()}
print("PMC per launch (KB):", res)
for k in res.get("FETCH_SIZE", {}):
    f = res["FETCH_SIZE"][k][0] * 1024 * 2          # gfx950: FETCH_SIZE counts 64 B per 128 B request
    w = res.get("WRITE_SIZE", {}).get(k, (0, 0))[0] * 1024
    print(f"{k}: fetch {f/1e6:.1f} MB (corrected x2) + write {w/1e6:.1f} MB = {(f+w)/1e6:.1f} MB per launch; algorithmic 96 * 2^24 = {96*2**24/1e6:.1f} MB")
PY
