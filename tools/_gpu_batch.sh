cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_pase_step.py tests/test_encoder.py tests/test_kernels.py tests/test_ddp_gloo.py -m gpu -q -s > gpurun_out/g18_tests.log 2>&1; tail -3 gpurun_out/g18_tests.log | cut -c1-200; grep "reserved 16" gpurun_out/g18_tests.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/g18_bench.json 2> gpurun_out/g18_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g18_bench.json')); print(d['value'], d['ms_per_step'])"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04b -o r1 -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-h2d > gpurun_out/prof_r04b.log 2>&1
