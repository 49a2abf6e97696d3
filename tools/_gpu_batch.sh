cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_conv_x6c.py tests/test_bs32_launches.py tests/test_encoder.py tests/test_pase_step.py -m gpu -q -x > gpurun_out/g33_tests.log 2>&1; tail -2 gpurun_out/g33_tests.log | cut -c1-200
timeout 300 python tools/step_breakdown.py gpurun_out/bd_j1.json > gpurun_out/bd_j1.txt 2>&1
PASE_LIB=tools/_ab/libpase_head2.so timeout 300 python tools/step_breakdown.py gpurun_out/bd_j0.json > gpurun_out/bd_j0.txt 2>&1
timeout 300 python tools/step_breakdown.py gpurun_out/bd_j1b.json > gpurun_out/bd_j1b.txt 2>&1
PASE_LIB=tools/_ab/libpase_head2.so timeout 300 python tools/step_breakdown.py gpurun_out/bd_j0b.json > gpurun_out/bd_j0b.txt 2>&1
for f in j1 j0 j1b j0b; do tail -n 1 gpurun_out/bd_$f.txt; done
