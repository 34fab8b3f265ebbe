#!/bin/bash
# per-kernel table of single-stream vgg11 proofs in the reference's semantics (fresh generators, full argument): gpurun_out/<tag>_fresh.md
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=${1:-prof}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
python $ROOT/scripts/exp/fresh_mode.py 4 > $OUT/${TAG}_fresh.log 2>&1
cd /tmp && export TMPDIR=/tmp
D=$OUT/${TAG}_fresh_trace
rm -rf $D
CLASSES=0 rocprofv3 --kernel-trace --stats -d $D -o trace -- python $ROOT/scripts/exp/fresh_mode.py 4 > $OUT/${TAG}_fresh_prof.log 2>&1 || true
python - <<PY
import glob, sqlite3, re
out = open("$OUT/${TAG}_fresh.md", "w")
out.write("# scripts/exp/fresh_mode.py 4: 2 + 4 + 4 single-stream vgg11 proofs (2 warm, 4 session generators + full argument, 4 FRESH generators + full argument)\n\n")
out.write(open("$OUT/${TAG}_fresh.log").read() + "\n")
for db in glob.glob("$D/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    rows = con.execute("select * from top_kernels limit 45").fetchall()
    cols = [c[1] for c in con.execute("pragma table_info(top_kernels)")]
    out.write("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols) + "\n")
    for r in rows:
        r = list(r)
        r[0] = re.sub(r"^void ", "", str(r[0]))
        r[0] = re.sub(r"call_f<&\(?(void )?", "", r[0])
        r[0] = r[0].split("(")[0][:70] if not r[0].startswith("k_run") else r[0][:70]
        out.write("| " + " | ".join(str(x) if not isinstance(x, float) else "%.1f" % x for x in r) + " |\n")
out.close()
print(open("$OUT/${TAG}_fresh.md").read())
PY
rm -rf $D
