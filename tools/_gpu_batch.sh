cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sinc_x6.py tests/test_conv_x6c.py tests/test_kernels.py tests/test_encoder.py -m gpu -q > gpurun_out/g5_tests.log 2>&1; tail -4 gpurun_out/g5_tests.log | cut -c1-300
timeout 300 python tools/step_breakdown.py gpurun_out/gl_sinc.json > gpurun_out/gl_sinc.txt 2>&1; tail -1 gpurun_out/gl_sinc.txt; grep "Cin1 x 251" gpurun_out/gl_sinc.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/g5_bench.json 2> gpurun_out/g5_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g5_bench.json')); print(d['value'], d['ms_per_step'])"
