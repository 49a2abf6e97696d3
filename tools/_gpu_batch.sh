cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_x6c.py tests/test_wgrad_x6c.py tests/test_pase_step.py tests/test_bs32_launches.py -m gpu -q > gpurun_out/g4_tests.log 2>&1; tail -5 gpurun_out/g4_tests.log | cut -c1-300
timeout 300 python tools/step_breakdown.py gpurun_out/gl_xp.json > gpurun_out/gl_xp.txt 2>&1; tail -1 gpurun_out/gl_xp.txt
PASE_X6C_XP=0 timeout 300 python tools/step_breakdown.py gpurun_out/gl_noxp.json > gpurun_out/gl_noxp.txt 2>&1; tail -1 gpurun_out/gl_noxp.txt
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/g4_bench.json 2> gpurun_out/g4_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g4_bench.json')); print(d['value'], d['ms_per_step'])"
