#!/bin/bash
# Collect this round's rocprofv3 evidence on the GPU box (run through gpurun):  bash tools/profile_round.sh r02
#   1. kernel trace + stats of the bench command (per-kernel time),
#   2. PMC passes in their OWN runs (--kernel-trace only beside --pmc): FETCH_SIZE, WRITE_SIZE, SQ counters,
#   3. tools/summarize_profiles.py -> profiles/summary_<tag>.json (carries the kernel-source digest).
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out
B1="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-torch-gpu-baseline --no-h2d --no-capped-leg"
B2="python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-torch-gpu-baseline --no-h2d --no-capped-leg"
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o r1 -- $B1 > gpurun_out/prof_$TAG.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch_$TAG -o r -- $B2 > gpurun_out/pmc_fetch_$TAG.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write_$TAG -o r -- $B2 > gpurun_out/pmc_write_$TAG.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d gpurun_out/pmc_sq_$TAG -o r -- $B2 > gpurun_out/pmc_sq_$TAG.log 2>&1
ls gpurun_out/prof_$TAG gpurun_out/pmc_fetch_$TAG gpurun_out/pmc_write_$TAG gpurun_out/pmc_sq_$TAG
python tools/summarize_profiles.py $TAG > gpurun_out/summary_$TAG.txt 2>&1
cp profiles/summary_$TAG.json gpurun_out/ 2>/dev/null
cp gpurun_out/prof_$TAG/r1_kernel_stats.csv gpurun_out/bench_${TAG}_kernel_stats.csv 2>/dev/null
tail -5 gpurun_out/summary_$TAG.txt
