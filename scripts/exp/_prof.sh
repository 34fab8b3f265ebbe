cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 800 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t -- python /root/repo/bench.py --workload ${1:-lenet} --streams 1 --steps ${2:-4} --warmup 1 --no-cpu-baseline --no-companions --no-pmc > /root/repo/gpurun_out/prof.log 2>&1
python3 - <<PY
import glob, sqlite3
for db in glob.glob("/tmp/prof/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for r in con.execute("select * from top_kernels limit 16").fetchall():
        print([str(x).split("(")[0][:50] if isinstance(x, str) else (round(x, 1) if isinstance(x, float) else x) for x in r])
PY
