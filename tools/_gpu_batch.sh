cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_wgrad_x6c.py -m gpu -q > gpurun_out/g12_tests.log 2>&1; tail -2 gpurun_out/g12_tests.log | cut -c1-200
timeout 300 python tools/trace_x6c.py wg5 wg7 2>&1 | grep "workgroup\|==" | cut -c1-260 > gpurun_out/trace_wg.txt
timeout 300 python tools/step_breakdown.py gpurun_out/gl_e.json > gpurun_out/gl_e.txt 2>&1; tail -1 gpurun_out/gl_e.txt
PASE_X6C_WGRAD_MODE=4 timeout 300 python tools/step_breakdown.py gpurun_out/gl_e4.json > gpurun_out/gl_e4.txt 2>&1; tail -1 gpurun_out/gl_e4.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/g12_bench.json 2> gpurun_out/g12_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g12_bench.json')); print(d['value'], d['ms_per_step'])"
