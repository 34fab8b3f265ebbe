#!/bin/bash
# rocprofv3 kernel trace of bench.py (single stream and default streams) -> gpurun_out/<tag>_{s1,sK}.md  (copy the ones to keep into profiles/)
# usage: [WORKLOAD=vgg11_pp8] [SHAPES="1 4:4 56"] scripts/kernel_stats.sh <tag>      (a shape is <proofs in flight>[:<lanes per batch>])
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-prof}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# S = proofs in flight: 1 = a lone proof (resident round kernels), 48 = six lock-step batches of eight lanes (the default bench shape)
for S in ${SHAPES:-1 64}; do
  LANES=8; [ $S = 1 ] && LANES=1
  case $S in *:*) LANES=${S#*:}; S=${S%:*};; esac
  D=$OUT/${TAG}_s$S
  rm -rf $D
  rocprofv3 --kernel-trace --stats -d $D -o trace -- python $ROOT/bench.py --inner ${WORKLOAD:+--workload $WORKLOAD} --steps ${STEPS:-5} --warmup 2 --streams $S --lanes $LANES --no-cpu-baseline --no-companions --no-pmc > $OUT/${TAG}_s$S.log 2>&1 || true
  python - <<PY
import glob, sqlite3, json
dbs = glob.glob("$D/**/*.db", recursive=True)
out = open("$OUT/${TAG}_s$S.md", "w")
line = [l for l in open("$OUT/${TAG}_s$S.log") if l.startswith("{")]
out.write("# rocprofv3 --kernel-trace --stats -- python bench.py --inner ${WORKLOAD:+--workload $WORKLOAD} --steps ${STEPS:-5} --warmup 2 --streams $S --lanes $LANES --no-cpu-baseline --no-companions --no-pmc\n\n")
if line:
    d = json.loads(line[-1])
    out.write("bench line under the profiler: value %.2f proofs/s, prover_ms_per_image %.1f, roofline %s\n\n" % (d["value"], d["prover_ms_per_image"], json.dumps({k: d["roofline"][k] for k in ("kernel", "achieved", "frac", "avg_launch_ms")})))
for db in dbs:
    con = sqlite3.connect(db)
    try:
        rows = con.execute("select name, calls, total_duration, average, percentage from top_kernels order by total_duration desc limit 40").fetchall()
        cols = ["name", "calls", "total_ns", "avg_ns", "pct"]
    except Exception as e:
        rows = con.execute("select * from top_kernels limit 40").fetchall()
        cols = [c[1] for c in con.execute("pragma table_info(top_kernels)")]
    out.write("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols) + "\n")
    for r in rows:
        r = list(r)
        import re
        r[0] = re.sub(r"^void ", "", str(r[0]))
        r[0] = re.sub(r"call_f<&\(?(void )?", "", r[0])
        r[0] = r[0].split("(")[0][:70] if not r[0].startswith("k_run") else r[0][:70]
        out.write("| " + " | ".join(str(x) if not isinstance(x, float) else "%.1f" % x for x in r) + " |\n")
out.close()
print(open("$OUT/${TAG}_s$S.md").read())
PY
  rm -rf $D          # (the trace database is tens of MB; gpurun brings back at most 64 MiB)
done
