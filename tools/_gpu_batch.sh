cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ddp_gloo.py -m gpu -q -s > gpurun_out/g14_tests.log 2>&1; tail -4 gpurun_out/g14_tests.log | cut -c1-300; grep "reserved 16" gpurun_out/g14_tests.log
bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1; tail -3 gpurun_out/profile_round_r04.log
