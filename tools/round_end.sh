# Round-end evidence on the GPU box (through gpurun):  bash tools/round_end.sh [tests|prof|bench ...]   (default: all)
cd ${GRAFT_REPO_ROOT:-/root/repo}
WHAT=${@:-tests prof bench}
for w in $WHAT; do
case $w in
tests)
  timeout 1400 python -m pytest tests -m gpu -q > gpurun_out/gpu_suite_r05.log 2>&1; tail -4 gpurun_out/gpu_suite_r05.log | cut -c1-200 ;;
prof)
  bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1; tail -3 gpurun_out/profile_round_r05.log ;;
bench)
  python bench.py > gpurun_out/bench_r05.json 2> gpurun_out/bench_r05.err; tail -c 400 gpurun_out/bench_r05.json
  python tools/step_breakdown.py gpurun_out/gemm_launches_r05.json > gpurun_out/gemm_launches_r05.txt 2>&1
  PASE_X6=0 python tools/step_breakdown.py gpurun_out/gemm_launches_fp32_r05.json > gpurun_out/gemm_launches_fp32_r05.txt 2>&1
  PASE_X6C_FORCE=1 PASE_X6C_WGRAD_FLAT=1 python tools/step_breakdown.py gpurun_out/gemm_launches_x6c_forced_r05.json > gpurun_out/gemm_launches_x6c_forced_r05.txt 2>&1
  tail -n 1 gpurun_out/gemm_launches_r05.txt gpurun_out/gemm_launches_fp32_r05.txt gpurun_out/gemm_launches_x6c_forced_r05.txt
  python bench.py --producer --no-cpu-baseline > gpurun_out/bench_producer_r05.json 2> gpurun_out/bench_producer_r05.err; tail -c 300 gpurun_out/bench_producer_r05.json
  python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/bench_2rank_smoke_r05.json 2> gpurun_out/bench_2rank_smoke_r05.err; tail -c 1200 gpurun_out/bench_2rank_smoke_r05.json ;;
esac
done
