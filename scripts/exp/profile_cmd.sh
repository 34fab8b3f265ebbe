#!/bin/bash
# rocprofv3 --kernel-trace --stats around any command: per-kernel table -> gpurun_out/<tag>_kernels.md (the trace database is deleted: gpurun brings back <= 64 MiB)
# usage: scripts/exp/profile_cmd.sh <tag> <command ...>     e.g.  scripts/exp/profile_cmd.sh r05_fresh_batches python scripts/exp/batch_mode.py 0x81 8 3 4
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
TAG=$1; shift
OUT=$ROOT/gpurun_out
mkdir -p $OUT
CMD=("$@")
for i in "${!CMD[@]}"; do [ -e "$ROOT/${CMD[$i]}" ] && CMD[$i]="$ROOT/${CMD[$i]}"; done
cd /tmp && export TMPDIR=/tmp
D=$OUT/${TAG}_trace
rm -rf $D
rocprofv3 --kernel-trace --stats -d $D -o trace -- "${CMD[@]}" > $OUT/${TAG}_kernels.log 2>&1 || true
export OUT TAG D
python3 - "$@" <<'PY'
import glob, os, re, sqlite3, sys
OUT, TAG, D = os.environ["OUT"], os.environ["TAG"], os.environ["D"]
out = open(f"{OUT}/{TAG}_kernels.md", "w")
out.write("# rocprofv3 --kernel-trace --stats -- " + " ".join(sys.argv[1:]) + "\n\n")
out.write("".join(l for l in open(f"{OUT}/{TAG}_kernels.log") if l.startswith("[")) + "\n")
for db in glob.glob(D + "/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    rows = con.execute("select * from top_kernels limit 45").fetchall()
    cols = [c[1] for c in con.execute("pragma table_info(top_kernels)")]
    out.write("| " + " | ".join(cols) + " |\n|" + "---|" * len(cols) + "\n")
    for r in rows:
        r = list(r)
        r[0] = re.sub(r"^void ", "", str(r[0]))
        r[0] = re.sub(r"call_f<&\(?(void )?", "", r[0])
        r[0] = r[0][:90]
        out.write("| " + " | ".join(str(x) if not isinstance(x, float) else "%.1f" % x for x in r) + " |\n")
out.close()
print(open(f"{OUT}/{TAG}_kernels.md").read())
PY
rm -rf $D
