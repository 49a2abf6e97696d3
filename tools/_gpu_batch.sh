cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_wgrad_x6c.py -m gpu -q > gpurun_out/g7_tests.log 2>&1; tail -2 gpurun_out/g7_tests.log | cut -c1-300
timeout 300 python tools/step_breakdown.py gpurun_out/gl_b.json > gpurun_out/gl_b.txt 2>&1; tail -1 gpurun_out/gl_b.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/g7_bench.json 2> gpurun_out/g7_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g7_bench.json')); print(d['value'], d['ms_per_step'])"
