cd /tmp && export TMPDIR=/tmp
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for i in 1 2 3; do
for sgr in 0 4 8; do
PASE_X6C_STAGGER=$sgr timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d > gpurun_out/g27_bench_s$sgr.json 2> gpurun_out/g27_bench.err; python -c "
import json; d=json.load(open('gpurun_out/g27_bench_s$sgr.json')); print('stagger $sgr', d['value'], d['ms_per_step'])"
done; done
