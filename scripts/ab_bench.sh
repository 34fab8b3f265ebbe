#!/bin/bash
# A/B runs of bench.py over library builds kept under ab/<name>/ (scratch, not tracked): prints the prover split per variant.
# usage: scripts/ab_bench.sh "<name>[:ENV=VAL]" ...   e.g.  scripts/ab_bench.sh lib_old lib_new:GPU_MAX_HW_QUEUES=8 lib_new
cd "$(dirname "$0")/.."
WL=${WORKLOAD:-vgg11}
for v in "$@"; do
  name=${v%%:*}; envs=""
  [[ "$v" == *:* ]] && envs=${v#*:}
  cp ab/$name/*.so zkcnn_amd/lib/
  for rep in 1 2; do
    out=$(env $envs timeout 300 python bench.py --workload $WL --steps 10 --warmup 2 --streams ${STREAMS:-1} --no-cpu-baseline ${EXTRA:---no-companions --no-pmc} 2>/dev/null | tail -1)
    echo "$v rep$rep: $(echo "$out" | python3 -c 'import sys,json; d=json.loads(sys.stdin.read()); print("proofs/s", d["value"], "ms/img", d["prover_ms_per_image"], "sumcheck", d["prover_ms_sumcheck"], "commit", d["prover_ms_commit"], "pass", d["verifier_pass"])')"
  done
done
