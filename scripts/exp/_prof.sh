cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof
timeout 800 rocprofv3 --kernel-trace --stats -d /tmp/prof -o t -- python /root/repo/bench.py --workload ${1:-lenet} --streams 1 --steps ${2:-20} --warmup 1 --no-cpu-baseline --no-companions --no-pmc > /root/repo/gpurun_out/prof.log 2>&1
grep '^{' /root/repo/gpurun_out/prof.log | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['prover_ms_per_image'], d['prover_ms_sumcheck'], d['prover_ms_commit'])"
python3 - <<PY
import glob, sqlite3
for db in glob.glob("/tmp/prof/**/*.db", recursive=True):
    con = sqlite3.connect(db)
    for r in con.execute("select * from top_kernels limit 40").fetchall():
        print([str(x).split("(")[0][:50] if isinstance(x, str) else (round(x, 1) if isinstance(x, float) else x) for x in r])
PY
