#!/bin/bash
# Everything the round's summary quotes, on one box: GPU test suite, the default bench line, kernel statistics (1 / 8 streams), the other
# BASELINE configurations, the two-rank rehearsal on one GPU and the soak of the resident kernels -> gpurun_out/<tag>_*  (usage: scripts/round_end.sh [tag])
TAG=${1:-r03y}
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; grep -a "passed\|failed" gpurun_out/${TAG}_pytest.log | tail -2
timeout 900 python bench.py > gpurun_out/${TAG}_final_bench.json 2> gpurun_out/${TAG}_final_bench.err; tail -c 300 gpurun_out/${TAG}_final_bench.json
bash scripts/kernel_stats.sh ${TAG} > /dev/null 2>&1; head -4 gpurun_out/${TAG}_s1.md | tail -2
bash scripts/config_table.sh ${TAG} 2>&1 | tail -12
timeout 600 python bench.py --gpus 2 --rehearse-shared-gpu --streams 4 --no-cpu-baseline --no-companions --no-pmc > gpurun_out/${TAG}_rehearsal.json 2> gpurun_out/${TAG}_rehearsal.err; tail -c 400 gpurun_out/${TAG}_rehearsal.json
timeout 900 python scripts/soak_resident.py 12 4 2>&1 | tail -5
