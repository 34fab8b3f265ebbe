#!/bin/bash
# PMC counters of the quadratic round, before (round 4's loop: k_old) and after (the streaming form: k_new / k_round_quad2<0>), from the stand-alone
# experiment scripts/exp/round_lab (two 2^24-entry tables). One counter group per pass (--pmc with --kernel-trace only: MI355X_MICROARCH.md, PMC slots).
# usage (GPU box): scripts/pmc_round_lab.sh [tag]   -> gpurun_out/<tag>_round_pmc.md
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-r05}
OUT=$ROOT/gpurun_out/${TAG}_round_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_available.txt 2>&1 || true
i=0
while read -r GROUP; do
  [ -z "$GROUP" ] && continue
  i=$((i + 1))
  timeout 300 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/pass$i -o pmc -- $ROOT/scripts/exp/round_lab 24 5 > $OUT/pass$i.log 2>&1 || echo "pass $i ($GROUP) failed" >> $OUT/errors.txt
done <<'GROUPS'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM
GRBM_GUI_ACTIVE GRBM_COUNT
TCC_HIT_sum TCC_MISS_sum
TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
FETCH_SIZE
WRITE_SIZE
GROUPS
export OUT ROOT TAG
python3 - <<'PY'
import csv, glob, collections, os
OUT, ROOT, TAG = os.environ["OUT"], os.environ["ROOT"], os.environ["TAG"]
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(OUT + "/pass*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void ", "")
        k = k.split("(")[0]
        if not (k.startswith("k_old") or k.startswith("k_new") or k.startswith("k_round_quad2") or k.startswith("k_stream21")):
            continue
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in rows for c in rows[k]})
kernels = sorted(rows)
with open(f"{ROOT}/gpurun_out/{TAG}_round_pmc.md", "w") as o:
    o.write("# PMC counters per launch (mean over the launches of the run), scripts/pmc_round_lab.sh, two 2^24-entry tables\n\n")
    o.write("| counter | " + " | ".join(kernels) + " |\n|---|" + "---|" * len(kernels) + "\n")
    for c in names:
        o.write(f"| {c} | " + " | ".join((f"{sum(rows[k][c]) / len(rows[k][c]):.4g}" if rows[k][c] else "") for k in kernels) + " |\n")
    o.write("\nDerived (per kernel):\n\n")
    for k in kernels:
        m = {c: sum(v) / len(v) for c, v in rows[k].items() if v}
        d = []
        if "SQ_WAVE_CYCLES" in m and m.get("SQ_WAVE_CYCLES"):
            wc = m["SQ_WAVE_CYCLES"]
            for c in ("SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
                if c in m:
                    d.append(f"{c}/WAVE_CYCLES {m[c] / wc:.3f}")
        if "SQ_BUSY_CYCLES" in m and "SQ_WAVE_CYCLES" in m and m["SQ_BUSY_CYCLES"]:
            d.append(f"waves resident per busy SQ cycle {m['SQ_WAVE_CYCLES'] / m['SQ_BUSY_CYCLES']:.2f}")
        if "TCC_HIT_sum" in m and "TCC_MISS_sum" in m and m["TCC_HIT_sum"] + m["TCC_MISS_sum"]:
            d.append(f"L2 hit rate {m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum']):.3f}")
        if "TCP_TOTAL_CACHE_ACCESSES_sum" in m and "TCP_TCC_READ_REQ_sum" in m and m["TCP_TOTAL_CACHE_ACCESSES_sum"]:
            d.append(f"L1 requests passed to L2 / L1 accesses {m['TCP_TCC_READ_REQ_sum'] / m['TCP_TOTAL_CACHE_ACCESSES_sum']:.3f}")
        if "FETCH_SIZE" in m:
            d.append(f"FETCH_SIZE x2 (gfx950 correction) {2 * m['FETCH_SIZE'] * 1024 / 1e6:.0f} MB")
        if "WRITE_SIZE" in m:
            d.append(f"WRITE_SIZE {m['WRITE_SIZE'] * 1024 / 1e6:.0f} MB")
        o.write(f"* {k}: " + "; ".join(d) + "\n")
print(open(f"{ROOT}/gpurun_out/{TAG}_round_pmc.md").read())
PY
