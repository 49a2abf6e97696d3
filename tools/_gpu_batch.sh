cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_bench_config.py tests/test_ddp_gloo.py -m gpu -q -s > gpurun_out/g30_tests.log 2>&1; tail -3 gpurun_out/g30_tests.log | cut -c1-200; grep "reserved 16\|OUT OF TOL" gpurun_out/g30_tests.log | head; grep -A4 "worst relative" gpurun_out/g30_tests.log | head -40
