cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_conv_x6c.py tests/test_wgrad_x6c.py tests/test_encoder.py tests/test_pase_step.py tests/test_bs32_launches.py -m gpu -q -x > gpurun_out/g31_tests.log 2>&1; tail -2 gpurun_out/g31_tests.log | cut -c1-200
timeout 300 python tools/trace_x6c.py dgrad21525 blk7 wglps wgqrnn blk5 > gpurun_out/trace_hw.txt 2>&1
grep -h "==\|workgroup 0" gpurun_out/trace_hw.txt | cut -c1-330
timeout 300 python tools/step_breakdown.py gpurun_out/bd_h1.json > gpurun_out/bd_h1.txt 2>&1
PASE_LIB=tools/_ab/libpase_autowait.so timeout 300 python tools/step_breakdown.py gpurun_out/bd_h0.json > gpurun_out/bd_h0.txt 2>&1
timeout 300 python tools/step_breakdown.py gpurun_out/bd_h1b.json > gpurun_out/bd_h1b.txt 2>&1
PASE_LIB=tools/_ab/libpase_autowait.so timeout 300 python tools/step_breakdown.py gpurun_out/bd_h0b.json > gpurun_out/bd_h0b.txt 2>&1
for f in h1 h0 h1b h0b; do tail -n 1 gpurun_out/bd_$f.txt; done
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/g31_bench.json 2> gpurun_out/g31_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g31_bench.json')); print(d['value'], d['ms_per_step'])"
PASE_LIB=tools/_ab/libpase_autowait.so timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/g31_bench_auto.json 2> gpurun_out/g31_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g31_bench_auto.json')); print(d['value'], d['ms_per_step'])"
