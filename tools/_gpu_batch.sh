cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_x6c.py tests/test_wgrad_x6c.py tests/test_encoder.py tests/test_pase_step.py -m gpu -q -x > gpurun_out/g25_tests.log 2>&1; tail -2 gpurun_out/g25_tests.log | cut -c1-200
timeout 300 python tools/trace_x6c.py wg5 wg7 lps qrnn > gpurun_out/trace_dl.txt 2>&1
grep -h "==\|workgroup 0" gpurun_out/trace_dl.txt | cut -c1-330
timeout 300 python tools/step_breakdown.py gpurun_out/bd_d1.json > gpurun_out/bd_d1.txt 2>&1
PASE_LIB=tools/_ab/libpase_nodl.so timeout 300 python tools/step_breakdown.py gpurun_out/bd_d0.json > gpurun_out/bd_d0.txt 2>&1
timeout 300 python tools/step_breakdown.py gpurun_out/bd_d1b.json > gpurun_out/bd_d1b.txt 2>&1
PASE_LIB=tools/_ab/libpase_nodl.so timeout 300 python tools/step_breakdown.py gpurun_out/bd_d0b.json > gpurun_out/bd_d0b.txt 2>&1
for f in d1 d0 d1b d0b; do tail -n 1 gpurun_out/bd_$f.txt; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/g25_bench.json 2> gpurun_out/g25_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g25_bench.json')); print(d['value'], d['ms_per_step'])"
PASE_LIB=tools/_ab/libpase_nodl.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/g25_bench_nodl.json 2> gpurun_out/g25_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g25_bench_nodl.json')); print(d['value'], d['ms_per_step'])"
