# Round-end evidence on the GPU box (through gpurun):  bash tools/round_end.sh [tests|prof|bench|extra ...]   (default: all)
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=${TAG:-r06}
mkdir -p gpurun_out
WHAT=${@:-tests prof bench extra}
for w in $WHAT; do
case $w in
tests)
  timeout 1500 python -m pytest tests -m gpu -q > gpurun_out/gpu_suite_${TAG}.log 2>&1; tail -4 gpurun_out/gpu_suite_${TAG}.log | cut -c1-200 ;;
prof)
  bash tools/profile_round.sh ${TAG} > gpurun_out/profile_round_${TAG}.log 2>&1; tail -3 gpurun_out/profile_round_${TAG}.log ;;
bench)
  python bench.py > gpurun_out/bench_${TAG}.json 2> gpurun_out/bench_${TAG}.err; tail -c 400 gpurun_out/bench_${TAG}.json
  python tools/step_breakdown.py gpurun_out/gemm_launches_${TAG}.json > gpurun_out/gemm_launches_${TAG}.txt 2>&1
  PASE_X6=0 python tools/step_breakdown.py gpurun_out/gemm_launches_fp32_${TAG}.json > gpurun_out/gemm_launches_fp32_${TAG}.txt 2>&1
  PASE_X6C_FORCE=1 PASE_X6C_WGRAD_FLAT=1 python tools/step_breakdown.py gpurun_out/gemm_launches_x6c_forced_${TAG}.json > gpurun_out/gemm_launches_x6c_forced_${TAG}.txt 2>&1
  tail -n 1 gpurun_out/gemm_launches_${TAG}.txt gpurun_out/gemm_launches_fp32_${TAG}.txt gpurun_out/gemm_launches_x6c_forced_${TAG}.txt
  python bench.py --producer --no-cpu-baseline --no-torch-gpu-baseline > gpurun_out/bench_producer_${TAG}.json 2> gpurun_out/bench_producer_${TAG}.err; tail -c 300 gpurun_out/bench_producer_${TAG}.json
  python bench.py --gpus 2 --steps 4 --warmup 2 --no-cpu-baseline --no-torch-gpu-baseline > gpurun_out/bench_2rank_smoke_${TAG}.json 2> gpurun_out/bench_2rank_smoke_${TAG}.err; tail -c 600 gpurun_out/bench_2rank_smoke_${TAG}.json ;;
extra)
  # thermal steadiness: 200 timed steps; the CPU baseline once at the benchmark's own batch size
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-torch-gpu-baseline --no-h2d --no-capped-leg > gpurun_out/bench_${TAG}_200steps.json 2> gpurun_out/bench_${TAG}_200steps.err; python -c "
import json; d=json.load(open('gpurun_out/bench_${TAG}_200steps.json')); print('200 steps: ms/step', d['ms_per_step'], 'value', d['value'])"
  timeout 900 python bench.py --steps 5 --warmup 2 --no-h2d --no-capped-leg --no-torch-gpu-baseline --cpu-baseline-bs32 > gpurun_out/bench_${TAG}_cpu_bs32.json 2> gpurun_out/bench_${TAG}_cpu_bs32.err; python -c "
import json; d=json.load(open('gpurun_out/bench_${TAG}_cpu_bs32.json')); print('cpu bs32:', d['cpu_baseline'])"
  # BASELINE.json configs[4]'s model on one GPU; eight ranks sharing the GPU (functional smoke of the N = 8 path); host enqueue cost
  timeout 600 python bench.py --variant emb256 --no-h2d --no-capped-leg > gpurun_out/bench_emb256_${TAG}.json 2> gpurun_out/bench_emb256_${TAG}.err; tail -c 300 gpurun_out/bench_emb256_${TAG}.json
  timeout 600 python bench.py --gpus 8 --steps 4 --warmup 2 --no-h2d > gpurun_out/bench_8rank_smoke_${TAG}.json 2> gpurun_out/bench_8rank_smoke_${TAG}.err; tail -c 300 gpurun_out/bench_8rank_smoke_${TAG}.json
  timeout 300 python tools/host_time.py --json gpurun_out/host_time_${TAG}.json 2> gpurun_out/host_time_${TAG}.err | tail -1 ;;
esac
done
