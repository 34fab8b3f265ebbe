set -x
cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03z_pytest.log 2>&1; tail -3 gpurun_out/r03z_pytest.log
timeout 900 python bench.py > gpurun_out/r03z_final_bench.json 2> gpurun_out/r03z_final_bench.err; tail -c 600 gpurun_out/r03z_final_bench.json
bash scripts/kernel_stats.sh r03z > /dev/null 2>&1; head -12 gpurun_out/r03z_s1.md
