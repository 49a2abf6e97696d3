cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python tools/trace_x6c.py blk5 blk7 dec3 2>&1 | grep "workgroup 0\|==" | cut -c1-140 > gpurun_out/trace_slim2.txt
timeout 600 python -m pytest tests/test_conv_x6c.py tests/test_bs32_launches.py -m gpu -q > gpurun_out/g15_tests.log 2>&1; tail -2 gpurun_out/g15_tests.log | cut -c1-200
timeout 300 python tools/step_breakdown.py gpurun_out/gl_g.json > gpurun_out/gl_g.txt 2>&1; tail -1 gpurun_out/gl_g.txt
PASE_LIB=$PWD/tools/_ab/libpase_oldloop.so timeout 300 python tools/step_breakdown.py gpurun_out/gl_gold.json > gpurun_out/gl_gold.txt 2>&1; tail -1 gpurun_out/gl_gold.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/g15_bench.json 2> gpurun_out/g15_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g15_bench.json')); print(d['value'], d['ms_per_step'])"
