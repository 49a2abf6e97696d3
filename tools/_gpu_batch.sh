cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_wgrad_x6c.py tests/test_pase_step.py -m gpu -q -s > gpurun_out/g3_tests.log 2>&1; tail -5 gpurun_out/g3_tests.log | cut -c1-300
timeout 300 python tools/step_breakdown.py gpurun_out/gl_zp2.json > gpurun_out/gl_zp2.txt 2>&1; tail -1 gpurun_out/gl_zp2.txt
